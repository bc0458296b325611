/*
 * TEST INFRASTRUCTURE — CPU restatement ("oracle") of VeryFastTree's profile-operation hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (veryfasttree_amd/, include/vft_hip.h) never links or calls it.
 *
 * Parity status: PINNED.  Every function here is checked bit-for-bit (integer / numeric_t outputs) or to
 * 1e-12 relative (double log-likelihoods) against golden vectors dumped from the compiled reference by
 * oracle/whitebox.cpp (tests/golden/wb_*.npz, tests/test_oracle_golden.py).
 *
 * Functions exist in two precisions: suffix _f32 (numeric_t = float, reference backend SSE128) and
 * _f64 (numeric_t = double, reference backend AVX256).  See vft_oracle_impl.h for the file:line of the
 * reference member each one follows.
 */
#ifndef VFT_ORACLE_H
#define VFT_ORACLE_H
#include <stdint.h>

#define VFTO_NOCODE 127
#define VFTO_MAXCODES 20
#define VFTO_MAXRATES 64
#define VFTO_LK_UNDERFLOW 1.0e-4            /* Constants.h:13 */
#define VFTO_LK_UNDERFLOW_INV 1.0e4         /* Constants.h:14 */
#define VFTO_LOG_LK_UNDERFLOW 9.21034037197618 /* Constants.h:15 */

#define VFTO_DECL(REAL, S)                                                                                        \
    typedef struct { /* DistanceMatrix.h:15-33, row-major nCodes x nCodes */                                     \
        const REAL *distances, *codefreq, *eigenval, *eigentot;                                                   \
    } vfto_dmat_##S;                                                                                              \
    typedef struct { /* TransitionMatrix.h:65-76; codefreq has nCodes+1 rows, the last one is NOCODE */          \
        const REAL *stat, *statinv, *eigenval, *codefreq, *eigeninv, *eigeninvT;                                  \
    } vfto_tmat_##S;                                                                                              \
    typedef struct { /* the NJ state one sweep reads (NJ.h:256-300) */                                           \
        int64_t nSeqs, maxnode, nPos;                                                                             \
        int nCodes;                                                                                               \
        const REAL *W;                                                                                            \
        const unsigned char *C;                                                                                   \
        const REAL *F;                                                                                            \
        const int64_t *parent;                                                                                    \
        const REAL *diameter, *selfweight, *selfdist;                                                             \
        double totdiam;                                                                                           \
        const REAL *out_w;                                                                                        \
        const unsigned char *out_c;                                                                               \
        const REAL *out_f, *out_cd;                                                                               \
        const vfto_dmat_##S *dm;                                                                                  \
    } vfto_state_##S;                                                                                             \
    void vfto_seqdist_##S(const unsigned char *, const unsigned char *, int64_t, int, const REAL *, REAL *,      \
                          REAL *);                                                                                \
    void vfto_profiledist_##S(const REAL *, const unsigned char *, const REAL *, const REAL *,                    \
                              const unsigned char *, const REAL *, const REAL *, int64_t, int,                    \
                              const vfto_dmat_##S *, REAL *, REAL *);                                             \
    void vfto_average_profile_##S(REAL *, unsigned char *, REAL *, const REAL *, const unsigned char *,           \
                                  const REAL *, const REAL *, const unsigned char *, const REAL *, int64_t, int,  \
                                  double, const vfto_dmat_##S *, double);                                         \
    void vfto_out_profile_##S(REAL *, unsigned char *, REAL *, REAL *, const REAL *, const unsigned char *,       \
                              const REAL *, int64_t, int64_t, int, const vfto_dmat_##S *, double);                \
    void vfto_update_out_profile_##S(REAL *, REAL *, REAL *, const unsigned char *, const REAL *,                 \
                                     const unsigned char *, const REAL *, const REAL *, const unsigned char *,    \
                                     const REAL *, const REAL *, const unsigned char *, const REAL *, int64_t,    \
                                     int64_t, int, const vfto_dmat_##S *, double);                                \
    REAL vfto_out_distance_##S(REAL, REAL, int64_t, REAL, REAL, REAL, double);                                    \
    REAL vfto_criterion_##S(REAL, REAL, int64_t, REAL, int64_t, int64_t);                                         \
    void vfto_set_dist_criterion_##S(const vfto_state_##S *, int64_t, int64_t, int64_t, int64_t, REAL *,          \
                                     int64_t *, REAL *, REAL *, REAL *);                                          \
    void vfto_set_best_hit_##S(const vfto_state_##S *, int64_t, int64_t, int64_t, REAL *, int64_t *, int64_t *,   \
                               int64_t *, REAL *, REAL *, REAL *, int64_t *);                                     \
    void vfto_sort_hits_##S(const REAL *, int64_t, int64_t *);                                                    \
    void vfto_psame_pdiff_##S(double, const REAL *, int, double *, double *);                                     \
    void vfto_exp_eigen_rates_##S(double, const REAL *, int, const REAL *, int, double, REAL *);                  \
    double vfto_pair_loglk_##S(const REAL *, const unsigned char *, const REAL *, const REAL *,                   \
                               const unsigned char *, const REAL *, int64_t, int, double, const REAL *, int,      \
                               const int64_t *, const vfto_tmat_##S *, double, double *);                         \
    void vfto_posterior_profile_##S(REAL *, unsigned char *, REAL *, const REAL *, const unsigned char *,         \
                                    const REAL *, const REAL *, const unsigned char *, const REAL *, int64_t,     \
                                    int, double, double, const REAL *, int, const int64_t *,                      \
                                    const vfto_tmat_##S *, double, double);

#ifdef __cplusplus
extern "C" {
#endif
VFTO_DECL(float, f32)
VFTO_DECL(double, f64)
/* FNV-1a over (w, c, f) of a dense profile with non-vector columns zeroed: the fixture's profile hash */
int64_t vfto_profile_hash(const void *w, const unsigned char *c, const void *f, int64_t nPos, int nCodes,
                          int realBytes);
#ifdef __cplusplus
}
#endif
#endif
