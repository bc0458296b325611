#define _GNU_SOURCE   /* sched_setaffinity: the timed leg pins its threads */
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
#include <sched.h>
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

/* the per-(column, code) addends of a leaf target against a profile query: code c -> (wgt * (1 - fq[c]), wgt), or
   (wgt * (cq == c ? 0 : 1), wgt) when the query holds a code; nothing for gaps on either side.  NULL for a leaf query. */
static double *query_table(const vfto_state_f32 *st, int64_t node) {
    const int64_t nPos = st->nPos;
    if (node < st->nSeqs) return NULL;
    const float *wq = st->W + node * nPos, *fq = st->F + node * nPos * 4;
    const unsigned char *cq = st->C + node * nPos;
    double *tab = (double *) malloc((size_t) nPos * 8 * sizeof(double));
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
    return tab;
}

/* one target of a sweep: seqDist / profileDist + the criterion; inactive targets get the reference's sentinel */
static inline void sweep_target(const vfto_state_f32 *st, const double *tab, int64_t node, int64_t j, int64_t nActive,
                                const float *outDist, const int64_t *nOutActive, float *hit_weight, float *hit_dist, float *hit_crit) {
    const int64_t nPos = st->nPos, nSeqs = st->nSeqs;
    const float *wq = st->W + node * nPos, *fq = st->F + node * nPos * 4;
    const unsigned char *cq = st->C + node * nPos;
    const int qLeaf = node < nSeqs;
    if (st->parent[j] >= 0) {
        hit_weight[j] = 0;
        hit_crit[j] = hit_dist[j] = 1e20f;
        return;
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

/* One sweep of `node` against every node below maxnode; inactive targets get the reference's sentinel.  nt, no matrix,
   float.  nThreads <= 0: OpenMP's default. */
void vfto_avx2_sweep_f32(const vfto_state_f32 *st, int64_t node, int64_t nActive, const float *outDist, const int64_t *nOutActive,
                         float *hit_weight, float *hit_dist, float *hit_crit, int nThreads) {
    double *tab = query_table(st, node);
    if (nThreads <= 0) nThreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 256) num_threads(nThreads)
    for (int64_t j = 0; j < st->maxnode; j++)
        sweep_target(st, tab, node, j, nActive, outDist, nOutActive, hit_weight, hit_dist, hit_crit);
    free(tab);
}

/* The timed leg of bench.py's cpu_baseline: the same sweeps over a copy of the state that the OpenMP team itself
   allocates and FIRST-TOUCHES - targets are dealt to the threads in blocks of VFTO_BENCH_BLOCK (schedule(static, block)),
   the same thread initialises and later sweeps a block, so every thread streams memory of its own NUMA node (a numpy
   array is first-touched by one thread: 128 cores then share one node's bandwidth).  Sweeps the queries round-robin
   until `budget` seconds have passed (at least once each); returns the seconds spent inside the sweeps and the number
   of sweeps in *nDone.  The last sweep's results come back in hit_* (maxnode floats each) for the bit-identity check. */
#define VFTO_BENCH_BLOCK 64
double vfto_avx2_sweep_bench_f32(const vfto_state_f32 *st, const int64_t *queries, int64_t nQueries, int64_t nActive,
                                 const float *outDist, const int64_t *nOutActive, int nThreads, double budget,
                                 float *hit_weight, float *hit_dist, float *hit_crit, int64_t *nDone) {
    const int64_t n = st->maxnode, nPos = st->nPos;
    if (nThreads <= 0) nThreads = omp_get_max_threads();
    vfto_state_f32 loc = *st;
    float *W = (float *) malloc((size_t) n * nPos * sizeof(float)), *F = (float *) malloc((size_t) n * nPos * 4 * sizeof(float));
    unsigned char *Cc = (unsigned char *) malloc((size_t) n * nPos);
    int64_t *par = (int64_t *) malloc((size_t) n * sizeof(int64_t)), *nOut = (int64_t *) malloc((size_t) n * sizeof(int64_t));
    float *diam = (float *) malloc((size_t) n * sizeof(float)), *od = (float *) malloc((size_t) n * sizeof(float));
    float *hw = (float *) malloc((size_t) n * sizeof(float)), *hd = (float *) malloc((size_t) n * sizeof(float)),
          *hc = (float *) malloc((size_t) n * sizeof(float));
    /* The threads of the first-touch and of the timed regions are pinned, evenly spread over the CPUs this process may use (and
       released afterwards): unpinned, a 64-thread team ran its first 8 sweeps at 41x the one-thread rate and the next 5 000 at
       9x - the scheduler moves threads away from the memory they first touched (no numactl on the box). */
    cpu_set_t allowed;
    int cpus[4096], nCpu = 0;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0)
        for (int c = 0; c < CPU_SETSIZE && nCpu < 4096; c++)
            if (CPU_ISSET(c, &allowed)) cpus[nCpu++] = c;
#define VFTO_PIN()                                                                                                 \
    cpu_set_t before, mine;                                                                                        \
    const int pinned = nCpu >= omp_get_num_threads() && sched_getaffinity(0, sizeof(before), &before) == 0;        \
    if (pinned) {                                                                                                  \
        CPU_ZERO(&mine);                                                                                           \
        CPU_SET(cpus[(int64_t) omp_get_thread_num() * nCpu / omp_get_num_threads()], &mine);                       \
        sched_setaffinity(0, sizeof(mine), &mine);                                                                 \
    }
#define VFTO_UNPIN() if (pinned) sched_setaffinity(0, sizeof(before), &before)
#pragma omp parallel num_threads(nThreads)
    {
    VFTO_PIN()
#pragma omp for schedule(static, VFTO_BENCH_BLOCK)
    for (int64_t j = 0; j < n; j++) {
        memcpy(W + j * nPos, st->W + j * nPos, (size_t) nPos * sizeof(float));
        memcpy(F + j * nPos * 4, st->F + j * nPos * 4, (size_t) nPos * 4 * sizeof(float));
        memcpy(Cc + j * nPos, st->C + j * nPos, (size_t) nPos);
        par[j] = st->parent[j];
        diam[j] = st->diameter[j];
        od[j] = outDist[j];
        nOut[j] = nOutActive[j];
        hw[j] = hd[j] = hc[j] = 0;
    }
    VFTO_UNPIN();
    }
    loc.W = W;
    loc.F = F;
    loc.C = Cc;
    loc.parent = par;
    loc.diameter = diam;
    /* the query tables of the distinct seeds, built once (the GPU step stages its queries on the device as well) */
    double **tabs = (double **) malloc((size_t) nQueries * sizeof(double *));
    for (int64_t q = 0; q < nQueries; q++) tabs[q] = query_table(&loc, queries[q]);
    /* one sweep per seed to calibrate, then all repetitions inside ONE parallel region: a region per sweep measured the team's
       wake-up (6-9 ms per sweep with 64-128 threads against 0.4 ms of work per thread), not the sweep */
    int64_t done = 0;
    double spent = 0;
    {
        const double t0 = omp_get_wtime();
        for (int64_t q = 0; q < nQueries; q++) {
#pragma omp parallel for schedule(static, VFTO_BENCH_BLOCK) num_threads(nThreads)
            for (int64_t j = 0; j < n; j++) sweep_target(&loc, tabs[q], queries[q], j, nActive, od, nOut, hw, hd, hc);
        }
        const double t1 = omp_get_wtime() - t0;
        (void) t1;
    }
    int64_t reps = nQueries;
    for (int round = 0; round < 2; round++) {   /* a short batch sizes the long one */
        const double t0 = omp_get_wtime();
#pragma omp parallel num_threads(nThreads)
        {
            VFTO_PIN()
            for (int64_t r = 0; r < reps; r++) {
                const int64_t q = r % nQueries;
#pragma omp for schedule(static, VFTO_BENCH_BLOCK)
                for (int64_t j = 0; j < n; j++) sweep_target(&loc, tabs[q], queries[q], j, nActive, od, nOut, hw, hd, hc);
            }
            VFTO_UNPIN();
        }
        const double dt = omp_get_wtime() - t0;
        if (round == 1 || dt >= budget) {
            spent = dt;
            done = reps;
            break;
        }
        const double per = dt / (double) reps;
        int64_t want = (int64_t) (budget / (per > 1e-9 ? per : 1e-9)) + 1;
        reps = ((want + nQueries - 1) / nQueries) * nQueries;   /* whole passes over the seeds: the last sweep is the last seed's */
    }
    for (int64_t q = 0; q < nQueries; q++) free(tabs[q]);
    free(tabs);
    if (hit_weight) memcpy(hit_weight, hw, (size_t) n * sizeof(float));
    if (hit_dist) memcpy(hit_dist, hd, (size_t) n * sizeof(float));
    if (hit_crit) memcpy(hit_crit, hc, (size_t) n * sizeof(float));
    *nDone = done;
    free(W); free(F); free(Cc); free(par); free(nOut); free(diam); free(od); free(hw); free(hd); free(hc);
    return spent;
}

int vfto_avx2_max_threads(void) { return omp_get_max_threads(); }
