// CPU check of vft_iterate_add (veryfasttree_amd/csrc/vft_iterate_add.h) against the step-by-step loop it replaces.
// Usage: iterate_add_check <cases> <maxcount> <seed>; prints "mismatches 0" when every case agrees.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include "../../veryfasttree_amd/csrc/vft_iterate_add.h"

static uint64_t rngState;
static uint64_t rnd() {
    rngState ^= rngState << 13;
    rngState ^= rngState >> 7;
    rngState ^= rngState << 17;
    return rngState;
}
static double uni() { return (double) (rnd() >> 11) / 9007199254740992.0; }

template <typename REAL>
static long run(long cases, uint64_t maxCount, const char *name) {
    long bad = 0;
    for (long t = 0; t < cases; t++) {
        double c;
        uint64_t count;
        const int kind = (int) (rnd() % 5);
        if (kind == 0) {   // the real thing: 1 / nActive, count <= nActive
            const uint64_t n = 2 + rnd() % maxCount;
            c = 1.0 / (double) n;
            count = rnd() % (n + 1);
        } else if (kind == 1) {   // random constant, random count
            c = ldexp(0.5 + 0.5 * uni(), -(int) (rnd() % 30));
            count = rnd() % (maxCount + 1);
        } else if (kind == 2) {   // constants with few mantissa bits: ties in many binades
            c = ldexp((double) (1 + rnd() % 64), -(int) (10 + rnd() % 30));
            count = rnd() % (maxCount + 1);
        } else if (kind == 3) {   // exactly half an ulp of some binade of REAL above the constant
            const int e = -(int) (rnd() % 20);
            const int p = sizeof(REAL) == 4 ? 24 : 53;
            c = ldexp((double) (2 * (rnd() % 1000) + 1), e - p - 1) + ldexp(1.0, e - 12);
            count = rnd() % (maxCount + 1);
        } else {   // large counts of a power of two
            c = ldexp(1.0, -(int) (rnd() % 24));
            count = rnd() % (maxCount + 1);
        }
        const REAL a = vft_iterate_add<REAL>(c, count), b = vft_iterate_add_ref<REAL>(c, count);
        if (memcmp(&a, &b, sizeof(REAL)) != 0) {
            if (bad < 5) fprintf(stderr, "%s: c = %a count = %llu: %a vs %a\n", name, c, (unsigned long long) count, (double) a, (double) b);
            bad++;
        }
    }
    return bad;
}

int main(int argc, char **argv) {
    const long cases = argc > 1 ? atol(argv[1]) : 2000;
    const uint64_t maxCount = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1000000;
    rngState = argc > 3 ? strtoull(argv[3], nullptr, 10) * 2654435761u + 88172645463325252ull : 88172645463325252ull;
    const long bad = run<float>(cases, maxCount, "float") + run<double>(cases, maxCount, "double");
    printf("mismatches %ld\n", bad);
    return bad != 0;
}
