// outProfile's weight chain over the LEAVES of an alignment (NJ.tcc:738-745): every active leaf that has a character at the
// column adds the same constant,
//         weight = (numeric_t)((double) weight + inweight),          inweight = 1.0 / nActive,
// up to a million times per column, one rounding per step.  The result of `count` such steps is computed here without
// taking them one by one: while the running value stays inside one binade [2^e, 2^(e+1)) every step adds the same
// multiple of that binade's ulp (the double sum is acc + c rounded to 2^(e-52), a fixed amount because acc is a multiple
// of the far larger numeric_t ulp; its rounding to numeric_t then moves acc by a fixed number of ulps, ties included once
// two consecutive steps have shown the same increment - after a tie the value is even and stays even).  So: take single
// steps until two consecutive increments agree and all three values share an exponent, then jump to the last value of
// the binade in one exact multiplication, and continue.  About three real steps per binade, ~25 binades.
// vft_iterate_add_ref is the loop itself; tests/native/iterate_add_check.c compares the two.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define VFT_IA_HD __host__ __device__ __forceinline__
#else
#define VFT_IA_HD static inline
#endif

template <typename REAL>
VFT_IA_HD REAL vft_iterate_add_ref(double c, uint64_t count) {
    REAL acc = 0;
    for (uint64_t t = 0; t < count; t++) acc = (REAL) ((double) acc + c);
    return acc;
}

template <typename REAL>
VFT_IA_HD int vft_ia_exponent(REAL x) {
    int e;
    (void) frexp((double) x, &e);
    return e;
}

template <typename REAL>
VFT_IA_HD REAL vft_iterate_add(double c, uint64_t count) {
    REAL acc = 0;
    uint64_t rem = count;
    while (rem > 0) {
        const REAL n1 = (REAL) ((double) acc + c);
        if (n1 == acc) break;   // the constant no longer moves the value: it never will
        rem--;
        if (rem < 2 || acc == 0) {
            acc = n1;
            continue;
        }
        const REAL n2 = (REAL) ((double) n1 + c);
        const double d1 = (double) n1 - (double) acc, d2 = (double) n2 - (double) n1;   // exact (neighbouring values)
        const int e = vft_ia_exponent<REAL>(acc);
        if (d1 != d2 || d1 <= 0 || vft_ia_exponent<REAL>(n2) != e) {
            acc = n1;
            continue;
        }
        // acc, n1, n2 lie in [2^(e-1), 2^e) (frexp's convention) and both steps added d1: so does every further step
        // whose result stays below 2^e.  Jump over k of them (conservatively one short of the last).
        const double top = ldexp(1.0, e);
        double q = floor((top - (double) n1) / d1);
        if (q >= 2.0) {
            q -= 2.0;   // rounding of the division + "strictly below the top"
            uint64_t k = q >= 1.8e19 ? rem : (uint64_t) q;
            if (k > rem) k = rem;
            acc = (REAL) ((double) n1 + (double) k * d1);   // exact: a multiple of the ulp below 2^e
            rem -= k;
        } else {
            acc = n1;
        }
    }
    return acc;
}
