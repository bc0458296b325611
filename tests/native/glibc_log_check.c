/* TEST INFRASTRUCTURE: vft_glibc_log (veryfasttree_amd/csrc/vft_glibc_log.h) against this machine's libm log() on n
   pseudo-random positive normal doubles drawn from the ranges the likelihood code produces (running products kept in
   [1e-4, 1e4], arguments near 1, and the full exponent range).  Prints the number of mismatching bit patterns. */
#include <stdio.h>
#include <stdlib.h>
#include "vft_glibc_log.h"

static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint64_t next(void) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double unit(void) { return (double) (next() >> 11) * (1.0 / 9007199254740992.0); }

int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 10000000;
    long bad = 0;
    for (long t = 0; t < n; t++) {
        double x;
        switch (t & 3) {
            case 0: x = exp(-9.3 + 18.6 * unit()); break;            /* [1e-4, 1e4], log-uniform */
            case 1: x = 0.93 + 0.14 * unit(); break;                   /* around 1: both paths and their borders */
            case 2: { uint64_t b = (next() & 0x7fffffffffffffffull); if ((b >> 52) == 0 || (b >> 52) == 0x7ff) b = 0x3ff0000000000000ull | (b & 0xfffffffffffffull); memcpy(&x, &b, 8); break; }
            default: x = unit(); if (x < 1e-300) x = 0.5; break;        /* (0, 1) uniform */
        }
        const double a = vft_glibc_log(x), b = log(x);
        if (memcmp(&a, &b, 8) != 0) {
            if (bad < 5) printf("MISMATCH x=%a mine=%a libm=%a\n", x, a, b);
            bad++;
        }
    }
    printf("%ld mismatches in %ld\n", bad, n);
    /* exp: the arguments of the P(t) tables (eigenvalue * rate * length <= 0, down to the underflow range), tiny
       arguments, and a band of positive ones */
    long ebad = 0;
    for (long t = 0; t < n; t++) {
        double x;
        switch (t & 3) {
            case 0: x = -30.0 * unit(); break;
            case 1: x = -760.0 * unit(); break;
            case 2: x = -exp(-45.0 * unit()); break;                  /* -1 ... -3e-20 */
            default: x = 720.0 * unit() - 10.0; break;
        }
        const double a = vft_glibc_exp(x), b = exp(x);
        if (memcmp(&a, &b, 8) != 0) {
            if (ebad < 5) printf("EXP MISMATCH x=%a mine=%a libm=%a\n", x, a, b);
            ebad++;
        }
    }
    printf("%ld exp mismatches in %ld\n", ebad, n);
    return bad != 0 || ebad != 0;
}
