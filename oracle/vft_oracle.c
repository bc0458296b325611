/* TEST INFRASTRUCTURE — see vft_oracle.h.  Compiled with -ffp-contract=off: no FMA contraction, like the
   reference build pinned in oracle/Makefile. */
#include "vft_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define REAL float
#define FN(x) CAT(x, _f32)
#include "vft_oracle_impl.h"
#include "vft_oracle_sort.h"
#undef REAL
#undef FN

#define REAL double
#define FN(x) CAT(x, _f64)
#include "vft_oracle_impl.h"
#include "vft_oracle_sort.h"
#undef REAL
#undef FN

int64_t vfto_profile_hash(const void *w, const unsigned char *c, const void *f, int64_t nPos, int nCodes,
                          int realBytes) {
    uint64_t h = 0xCBF29CE484222325ull;
    const unsigned char *p = (const unsigned char *) w;
    for (int64_t i = 0; i < nPos * realBytes; i++) h = (h ^ p[i]) * 0x100000001B3ull;
    for (int64_t i = 0; i < nPos; i++) h = (h ^ c[i]) * 0x100000001B3ull;
    /* vectors count only where the reference holds one (c == NOCODE && w > 0), zeros elsewhere */
    const unsigned char *fb = (const unsigned char *) f;
    for (int64_t i = 0; i < nPos; i++) {
        int has;
        if (realBytes == 4) has = ((const float *) w)[i] > 0 && c[i] == VFTO_NOCODE;
        else has = ((const double *) w)[i] > 0 && c[i] == VFTO_NOCODE;
        for (int64_t b = 0; b < (int64_t) nCodes * realBytes; b++) {
            unsigned char v = has ? fb[i * nCodes * realBytes + b] : 0;
            h = (h ^ v) * 0x100000001B3ull;
        }
    }
    return (int64_t) h;
}


/* ---- Knuth's ran_array as the reference uses it (vft_knuth.h) */
#include "vft_knuth.h"
/* out[0..n) = the reference's knuth_rand() stream from its default state (seed 314159) */
void vfto_knuth_stream(double *out, int64_t n) {
    vft_knuth g;
    vft_knuth_start(&g, 314159L);
    for (int64_t i = 0; i < n; i++) out[i] = vft_knuth_rand(&g);
}
/* Knuth's own self-test: ran_start(310952), `rounds` refills of `len` values -> first value of the last refill */
long vfto_knuth_selftest(int rounds, int len) {
    static long a[2009];
    vft_knuth g;
    vft_knuth_start(&g, 310952L);
    for (int m = 0; m <= rounds; m++) vft_knuth_array(&g, a, len);
    return a[0];
}
