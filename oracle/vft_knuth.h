/* TEST INFRASTRUCTURE (oracle) and, through veryfasttree_amd/host/KnuthRng.h, the host driver's column resampler.
 *
 * D. E. Knuth's lagged-Fibonacci generator "ran_array" (The Art of Computer Programming, vol. 2, 3rd ed., section 3.6;
 * the 2002 revision published as rng.c): x[n] = (x[n-100] - x[n-37]) mod 2^30, seeded by ran_start, consumed through a
 * 1009-element buffer of which only the first 100 values are used per refill.  The reference carries the same
 * routine (src/Knuth.cpp) and draws its bootstrap columns from it as 2^-30 * ran_arr_next() (Knuth.cpp:109-111);
 * its pipeline never calls ran_start, so the stream always starts from the routine's own default seed 314159
 * (Knuth.cpp:98-100) whatever -seed says.  Restated here from the published algorithm; pinned by Knuth's own check
 * value (ran_start(310952), 2009 refills of 1009 -> 995235265) and by 5000 values of the reference's stream
 * (tests/golden/wb_knuth.npz).  Header-only, plain C. */
#ifndef VFT_KNUTH_H
#define VFT_KNUTH_H

#define VFT_KN_KK 100
#define VFT_KN_LL 37
#define VFT_KN_MM (1L << 30)
#define VFT_KN_QUALITY 1009
#define VFT_KN_TT 70
#define vft_kn_diff(x, y) (((x) - (y)) & (VFT_KN_MM - 1))

typedef struct {
    long x[VFT_KN_KK];            /* generator state */
    long buf[VFT_KN_QUALITY];     /* refill buffer; buf[KK] = -1 marks the end of the usable part */
    int next;                     /* index of the next value in buf, or -1 before the first draw */
} vft_knuth;

static void vft_knuth_array(vft_knuth *g, long *aa, int n) {
    int i, j;
    for (j = 0; j < VFT_KN_KK; j++) aa[j] = g->x[j];
    for (; j < n; j++) aa[j] = vft_kn_diff(aa[j - VFT_KN_KK], aa[j - VFT_KN_LL]);
    for (i = 0; i < VFT_KN_LL; i++, j++) g->x[i] = vft_kn_diff(aa[j - VFT_KN_KK], aa[j - VFT_KN_LL]);
    for (; i < VFT_KN_KK; i++, j++) g->x[i] = vft_kn_diff(aa[j - VFT_KN_KK], g->x[i - VFT_KN_LL]);
}

static void vft_knuth_start(vft_knuth *g, long seed) {
    int t, j;
    long x[VFT_KN_KK + VFT_KN_KK - 1];
    long ss = (seed + 2) & (VFT_KN_MM - 2);
    for (j = 0; j < VFT_KN_KK; j++) {   /* bootstrap the buffer */
        x[j] = ss;
        ss <<= 1;
        if (ss >= VFT_KN_MM) ss -= VFT_KN_MM - 2;   /* cyclic shift of 29 bits */
    }
    x[1]++;                              /* make x[1] (and only x[1]) odd */
    for (ss = seed & (VFT_KN_MM - 1), t = VFT_KN_TT - 1; t;) {
        for (j = VFT_KN_KK - 1; j > 0; j--) {   /* "square" */
            x[j + j] = x[j];
            x[j + j - 1] = 0;
        }
        for (j = VFT_KN_KK + VFT_KN_KK - 2; j >= VFT_KN_KK; j--) {
            x[j - (VFT_KN_KK - VFT_KN_LL)] = vft_kn_diff(x[j - (VFT_KN_KK - VFT_KN_LL)], x[j]);
            x[j - VFT_KN_KK] = vft_kn_diff(x[j - VFT_KN_KK], x[j]);
        }
        if (ss & 1) {                    /* "multiply by z" */
            for (j = VFT_KN_KK; j > 0; j--) x[j] = x[j - 1];
            x[0] = x[VFT_KN_KK];         /* shift the buffer cyclically */
            x[VFT_KN_LL] = vft_kn_diff(x[VFT_KN_LL], x[VFT_KN_KK]);
        }
        if (ss) ss >>= 1;
        else t--;
    }
    for (j = 0; j < VFT_KN_LL; j++) g->x[j + VFT_KN_KK - VFT_KN_LL] = x[j];
    for (; j < VFT_KN_KK; j++) g->x[j - VFT_KN_LL] = x[j];
    for (j = 0; j < 10; j++) vft_knuth_array(g, x, VFT_KN_KK + VFT_KN_KK - 1);   /* warm things up */
    g->next = -1;
}

/* ran_arr_next: the next 30-bit value */
static long vft_knuth_next(vft_knuth *g) {
    if (g->next >= 0 && g->buf[g->next] >= 0) return g->buf[g->next++];
    vft_knuth_array(g, g->buf, VFT_KN_QUALITY);
    g->buf[VFT_KN_KK] = -1;
    g->next = 1;
    return g->buf[0];
}

/* the reference's knuth_rand(): uniform in [0, 1) */
static double vft_knuth_rand(vft_knuth *g) { return 9.31322574615479e-10 * (double) vft_knuth_next(g); }

#endif
