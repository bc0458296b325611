/* TEST INFRASTRUCTURE - the CPU baseline of SURVEY.md section 8d: an AVX2 + OpenMP restatement of the batched
 * one-vs-all sweep (setBestHit, NJ.tcc:3571-3646, over seqDist NJ.tcc:1601-1624 / profileDist NJ.tcc:1167-1190 with the
 * no-matrix branch of profileDistPiece NJ.tcc:919-940 and the criterion of NJ.tcc:1099-1107) for the nucleotide
 * workloads bench.py measures.  Only tests/ and bench.py's cpu_baseline leg load it (liboracle_avx2.so).
 *
 * How it uses the machine (what a well-written CPU backend for this path would do, the reference's own SSE/AVX backends
 * vectorise only the 4 values of one column):
 *   - OpenMP over targets, dynamic chunks;
 *   - leaf x leaf: 32 alignment columns per AVX2 compare (both present / differ masks, popcounts);
 *   - profile query x leaf target: the query is turned into a per-(column, code) table of exact addends once per
 *     sweep, a leaf then costs one table row per column;
 *   - profile x profile: SSE products of the two 4-vectors, double subtraction in the reference's order.
 * Every result is bit-identical to the scalar oracle (tests/test_oracle_golden.py::test_avx2_sweep_equals_scalar_oracle):
 * each (query, target) pair is still summed column by column in double; compiled with -ffp-contract=off.
 * Out-distances are taken as given (no lazy refresh): the caller passes fresh ones. */
#include <immintrin.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "vft_oracle.h"

static inline float criterion_f32(float dist, float outI, int64_t nOutI, float outJ, int64_t nOutJ, int64_t nActive) {
    double oi = outI, oj = outJ;
    if (nOutI != nActive) oi *= (double) (nActive - 1) / (double) (nOutI - 1);
    if (nOutJ != nActive) oj *= (double) (nActive - 1) / (double) (nOutJ - 1);
    return (float) ((double) dist - (oi + oj) / (double) (nActive - 2));
}

/* seqDist without a matrix: nUse / nDiff over 32 columns per step */
static inline void seq_counts(const unsigned char *a, const unsigned char *b, int64_t nPos, int64_t *nUse, int64_t *nDiff) {
    const __m256i gap = _mm256_set1_epi8((char) VFTO_NOCODE);
    int64_t use = 0, diff = 0, p = 0;
    for (; p + 32 <= nPos; p += 32) {
        const __m256i va = _mm256_loadu_si256((const __m256i *) (a + p)), vb = _mm256_loadu_si256((const __m256i *) (b + p));
        const unsigned ga = (unsigned) _mm256_movemask_epi8(_mm256_cmpeq_epi8(va, gap));
        const unsigned gb = (unsigned) _mm256_movemask_epi8(_mm256_cmpeq_epi8(vb, gap));
        const unsigned eq = (unsigned) _mm256_movemask_epi8(_mm256_cmpeq_epi8(va, vb));
        const unsigned both = ~(ga | gb);
        use += __builtin_popcount(both);
        diff += __builtin_popcount(both & ~eq);
    }
    for (; p < nPos; p++)
        if (a[p] != VFTO_NOCODE && b[p] != VFTO_NOCODE) {
            use++;
            diff += a[p] != b[p];
        }
    *nUse = use;
    *nDiff = diff;
}

/* One sweep of `node` against every node below maxnode; inactive targets get the reference's sentinel.  nt, no matrix,
   float.  nThreads <= 0: OpenMP's default. */
void vfto_avx2_sweep_f32(const vfto_state_f32 *st, int64_t node, int64_t nActive, const float *outDist, const int64_t *nOutActive,
                         float *hit_weight, float *hit_dist, float *hit_crit, int nThreads) {
    const int64_t nPos = st->nPos, nSeqs = st->nSeqs;
    const float *wq = st->W + node * nPos, *fq = st->F + node * nPos * 4;
    const unsigned char *cq = st->C + node * nPos;
    const int qLeaf = node < nSeqs;
    /* per-(column, code) addends of a leaf target against a profile query: code c -> (wgt * (1 - fq[c]), wgt), or
       (wgt * (cq == c ? 0 : 1), wgt) when the query holds a code; nothing for gaps on either side */
    double *tab = NULL;
    if (!qLeaf) {
        tab = (double *) malloc((size_t) nPos * 8 * sizeof(double));
        for (int64_t p = 0; p < nPos; p++) {
            const int qvec = wq[p] > 0 && cq[p] == VFTO_NOCODE;
            for (int c = 0; c < 4; c++) {
                double term = 0, wgt = 0;
                if (wq[p] > 0 && (qvec || cq[p] != VFTO_NOCODE)) {
                    const float ww = wq[p] * 1.0f;
                    wgt = (double) ww;
                    const double piece = qvec ? 1.0 - (double) fq[p * 4 + c] : (cq[p] == c ? 0.0 : 1.0);
                    term = wgt * piece;
                }
                tab[p * 8 + c] = term;
                tab[p * 8 + 4 + c] = wgt;
            }
        }
    }
    if (nThreads <= 0) nThreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 256) num_threads(nThreads)
    for (int64_t j = 0; j < st->maxnode; j++) {
        if (st->parent[j] >= 0) {
            hit_weight[j] = 0;
            hit_crit[j] = hit_dist[j] = 1e20f;
            continue;
        }
        float d, w;
        if (qLeaf && j < nSeqs) {
            int64_t nUse, nDiff;
            seq_counts(cq, st->C + j * nPos, nPos, &nUse, &nDiff);
            w = (float) (double) nUse;
            d = (float) (nUse > 0 ? (double) nDiff / (double) nUse : 1.0);
        } else {
            double top = 0, denom = 0;
            const unsigned char *ct = st->C + j * nPos;
            if (!qLeaf && j < nSeqs) {
                for (int64_t p = 0; p < nPos; p++) {
                    const unsigned c = ct[p];
                    if (c < 4) {
                        denom += tab[p * 8 + 4 + c];
                        top += tab[p * 8 + c];
                    }
                }
            } else {
                const float *wt = st->W + j * nPos, *ft = st->F + j * nPos * 4;
                for (int64_t p = 0; p < nPos; p++) {
                    if (!(wq[p] > 0 && wt[p] > 0)) continue;
                    const float ww = wq[p] * wt[p];
                    const double wgt = (double) ww;
                    denom += wgt;
                    const int c1 = cq[p], c2 = ct[p];
                    double piece;
                    if (c1 != VFTO_NOCODE) {
                        if (c2 != VFTO_NOCODE) piece = c1 == c2 ? 0.0 : 1.0;
                        else piece = 1.0 - (double) ft[p * 4 + c1];
                    } else if (c2 != VFTO_NOCODE) {
                        piece = 1.0 - (double) fq[p * 4 + c2];
                    } else {
                        float pr[4];
                        _mm_storeu_ps(pr, _mm_mul_ps(_mm_loadu_ps(fq + p * 4), _mm_loadu_ps(ft + p * 4)));
                        piece = 1.0;
                        piece -= (double) pr[0];
                        piece -= (double) pr[1];
                        piece -= (double) pr[2];
                        piece -= (double) pr[3];
                    }
                    top += wgt * piece;
                }
            }
            w = (float) (denom > 0 ? denom : 0.01);
            d = (float) (denom > 0 ? top / denom : 1.0);
            const float dd = st->diameter[node] + st->diameter[j];
            d = d - dd;
        }
        hit_dist[j] = d;
        hit_weight[j] = w;
        hit_crit[j] = criterion_f32(d, outDist[node], nOutActive[node], outDist[j], nOutActive[j], nActive);
    }
    free(tab);
}

int vfto_avx2_max_threads(void) { return omp_get_max_threads(); }
