// The two line-search kernels for alignments that do not fit the register-resident instances of vft_kernels_ml.h (more than 2 048
// columns; proteins without a matrix model beyond 2 048 as well).  k_ml_node_lengths / k_ml_quartet keep a thread's columns of the
// profiles a search works on in registers - CPT columns per thread, a template parameter, 8 at most before the arrays go to scratch and
// the staging of the ordered total outgrows the LDS.  Here the same searches run with the computed profiles (the posteriors a search is
// made against) in a per-workgroup WORKSPACE in global memory and the tree's own profiles read from the arena at every evaluation: a
// thread still owns the columns p = thread, thread + WG, ... - nothing crosses threads - and every loop over "my CPT columns" becomes a
// loop over "my columns of the alignment".  Slower per evaluation (every column is loaded again: L2 traffic instead of registers), no
// limit but the NJ phase's own (vft_api.hip: 10 240 columns).
//
// Same numbers as the register kernels wherever both run (tests/test_gpu_ml_long.py forces this path on the fixtures of the short one):
//   * Jukes-Cantor: the per-thread running products over the columns thread, thread + WG, ... in that order, the partial logs summed over
//     the wavefront by the same shuffles and over the wavefronts in order - the register kernels' sum with CPT = ceil(nPos / WG);
//   * matrix models: the reference's ordered total (NJ.tcc:1345-1447) by the plain chain (vft_lk_chain, as k_pair_loglk walks it:
//     VFT_ML_STAGE columns staged, one lane multiplies them through, the running product carried from stage to stage) - the sequence of
//     roundings vft_lk_total_staged reproduces with scans;
//   * posteriors and column likelihoods are the same device functions on the same operands.
#pragma once
#include "vft_kernels_ml.h"

// a profile-shaped scratch row of one workgroup: weights, codes, vectors (read only under a vector column, as everywhere)
template <typename REAL, int NC>
struct MlLongRow {
    REAL *w;
    uint8_t *c;
    REAL *f;
    __device__ __forceinline__ void load(int64_t p, Col<REAL, NC> &o) const {
        o.w = w[p];
        o.code = (int) c[p];
        o.vec = o.w > (REAL) 0 && o.code == VFT_NOCODE_;
        if (o.vec) {
#pragma unroll
            for (int k = 0; k < NC; k++) o.f[k] = f[p * NC + k];
        }
    }
    __device__ __forceinline__ void store(int64_t p, const Col<REAL, NC> &o) const {
        w[p] = o.w;
        c[p] = (uint8_t) o.code;
        if (o.w > (REAL) 0 && o.code == VFT_NOCODE_) {
#pragma unroll
            for (int k = 0; k < NC; k++) f[p * NC + k] = o.f[k];
        }
    }
};
// bytes of one row / of a workgroup's workspace (nRows rows + nPos doubles of per-site likelihoods), 16-byte pieces
__host__ __device__ inline size_t vft_ml_long_row_bytes(int64_t nPos, int nCodes, size_t rs) {
    return (((size_t) nPos * rs + 15) & ~(size_t) 15) + (((size_t) nPos + 15) & ~(size_t) 15) + (((size_t) nPos * nCodes * rs + 15) & ~(size_t) 15);
}
__host__ __device__ inline size_t vft_ml_long_ws_bytes(int64_t nPos, int nCodes, size_t rs, int nRows) {
    return (size_t) nRows * vft_ml_long_row_bytes(nPos, nCodes, rs) + (((size_t) nPos * 8 + 15) & ~(size_t) 15);
}
template <typename REAL, int NC>
__device__ __forceinline__ MlLongRow<REAL, NC> vft_ml_long_row(char *base, int64_t nPos, int r) {
    char *b = base + (size_t) r * vft_ml_long_row_bytes(nPos, NC, sizeof(REAL));
    MlLongRow<REAL, NC> R;
    R.w = (REAL *) b;
    R.c = (uint8_t *) (b + (((size_t) nPos * sizeof(REAL) + 15) & ~(size_t) 15));
    R.f = (REAL *) ((char *) R.c + (((size_t) nPos + 15) & ~(size_t) 15));
    return R;
}
// one operand of a posterior or a pair likelihood: a node of the tree (arena) or a workspace row - uniform over the workgroup
template <typename REAL, int NC>
struct MlLongSrc {
    int64_t node;   // >= 0: the arena's profile; < 0: `row`
    MlLongRow<REAL, NC> row;
    __device__ __forceinline__ void load(const Arena<REAL> &A, int64_t p, Col<REAL, NC> &o) const {
        if (node >= 0) vft_load_col_ml<REAL, NC>(A, node, p, o);
        else row.load(p, o);
    }
};
template <typename REAL, int NC>
__device__ __forceinline__ MlLongSrc<REAL, NC> vft_ml_src(int64_t node) {
    MlLongSrc<REAL, NC> s;
    s.node = node;
    s.row.w = nullptr;
    s.row.c = nullptr;
    s.row.f = nullptr;
    return s;
}
template <typename REAL, int NC>
__device__ __forceinline__ MlLongSrc<REAL, NC> vft_ml_src(const MlLongRow<REAL, NC> &row) {
    MlLongSrc<REAL, NC> s;
    s.node = -1;
    s.row = row;
    return s;
}

// pairLogLk(X, Y) over the workgroup from the tables of one branch length (PS / PD under Jukes-Cantor, e under a matrix model): result
// to every thread.  site != nullptr: the per-site likelihoods are multiplied in.  The caller has synchronised after writing the tables;
// on return every thread has finished reading them.
template <typename REAL, int NC, int WG>
__device__ __forceinline__ double vft_ml_long_total(const Arena<REAL> &A, const MlLongSrc<REAL, NC> &X, const MlLongSrc<REAL, NC> &Y, bool jc,
                                                    const double *PS, const double *PD, const REAL *e, double *site, double *stage,
                                                    double *red, double *totS) {
    const int64_t nPos = A.d.nPos;
    if (!jc || (NC == 4 && A.jcExact)) {   // the reference's ordered total, VFT_ML_STAGE columns at a time (k_pair_loglk's walk)
        double lk = 1.0, loglk = 0.0;
        for (int64_t p0 = 0; p0 < nPos; p0 += VFT_ML_STAGE) {
            const int64_t cnt = nPos - p0 < VFT_ML_STAGE ? nPos - p0 : VFT_ML_STAGE;
            for (int64_t q = threadIdx.x; q < cnt; q += WG) {
                const int64_t p = p0 + q;
                Col<REAL, NC> c1, c2;
                X.load(A, p, c1);
                Y.load(A, p, c2);
                const int r = A.ratecat[p];
                double lkAB;
                const bool has = vft_pair_lk_col<REAL, NC>(A, c1, c2, jc, PS[r], PD[r], e + r * NC, lkAB);
                if (has && site) site[p] *= lkAB;
                stage[q] = has ? lkAB : VFT_LK_SKIP;
            }
            __syncthreads();
            if (threadIdx.x == 0) vft_lk_chain(stage, cnt, jc, lk, loglk);
            __syncthreads();
        }
        if (threadIdx.x == 0) *totS = vft_lk_finish(lk, loglk);
        __syncthreads();
        const double t = *totS;
        __syncthreads();   // (the next call's thread 0 writes totS again)
        return t;
    }
    double lk = 1.0, loglk = 0.0;
    for (int64_t p = threadIdx.x; p < nPos; p += WG) {
        Col<REAL, NC> c1, c2;
        X.load(A, p, c1);
        Y.load(A, p, c2);
        const int r = A.ratecat[p];
        double lkAB;
        if (vft_pair_lk_col<REAL, NC>(A, c1, c2, jc, PS[r], PD[r], e + r * NC, lkAB)) {
            vft_lk_accumulate(lkAB, jc, lk, loglk);
            if (site) site[p] *= lkAB;
        }
    }
    double part = loglk + log(lk);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    double tot = 0;
#pragma unroll
    for (int w = 0; w < WG / 64; w++) tot += red[w];
    __syncthreads();   // (red is free again)
    return tot;
}

// k_ml_node_lengths (vft_kernels_ml.h) with the posterior of a round in workspace row 0; ws: one workspace of
// vft_ml_long_ws_bytes(nPos, NC, sizeof(REAL), 1) bytes per workgroup, wsStride apart
template <typename REAL, int NC>
__global__ __launch_bounds__((MlOptWG<NC>::value)) void k_ml_node_lengths_long(Arena<REAL> A, const int64_t *ids, const int64_t *lenIdx,
                                                                               const int64_t *recN, REAL *blen, double minLen, double minRel,
                                                                               double ftol, double atol, unsigned int *evalCount, char *ws,
                                                                               size_t wsStride) {
    constexpr int WG = MlOptWG<NC>::value;
    typedef Col<REAL, NC> ColT;
    __shared__ REAL ee1[VFT_MAXRATES * NC], ee2[VFT_MAXRATES * NC];
    __shared__ double pS1[VFT_MAXRATES], pD1[VFT_MAXRATES], pS2[VFT_MAXRATES], pD2[VFT_MAXRATES];
    __shared__ double red[WG / 64];
    __shared__ double stage[VFT_ML_STAGE];
    __shared__ double totS;
    const int64_t k = blockIdx.x;
    const bool jc = A.tmStat == nullptr;
    const int64_t nPos = A.d.nPos;
    const MlLongRow<REAL, NC> rowB = vft_ml_long_row<REAL, NC>(ws + (size_t) k * wsStride, nPos, 0);
    auto tables2 = [&](double l1, double l2) {
        if (l1 < minLen) l1 = minLen;   // NJ.tcc:2150-2155
        if (l2 < minLen) l2 = minLen;
        if (jc) {
            for (int r = threadIdx.x; r < A.nRates; r += WG) {
                vft_psame_pdiff(l1, (double) A.rates[r], pS1[r], pD1[r], A.jcExact != 0);
                vft_psame_pdiff(l2, (double) A.rates[r], pS2[r], pD2[r], A.jcExact != 0);
            }
        } else {
            vft_exp_eigen_rates<REAL, NC>(A, l1, minRel, ee1);
            vft_exp_eigen_rates<REAL, NC>(A, l2, minRel, ee2);
        }
    };
    auto post = [&](const ColT &c1, const ColT &c2, int r, ColT &o) __attribute__((always_inline)) {
        vft_posterior_col<REAL, NC>(A, c1, c2, jc, pS1[r], pD1[r], pS2[r], pD2[r], ee1 + r * NC, ee2 + r * NC, o.w, o.code, o.f);
        o.vec = o.code == VFT_NOCODE_ && o.w > (REAL) 0;
    };
    unsigned int nEval = 0;
    for (int round = 0; round < 6; round++) {   // 2 iterations x 3 branches (NJ.tcc:5038-5059)
        const int i = round % 3, b1 = (i + 1) % 3, b2 = (i + 2) % 3;
        const int64_t nI = ids[3 * k + i], n1 = ids[3 * k + b1], n2 = ids[3 * k + b2];
        const int64_t lI = lenIdx[3 * k + i];
        __syncthreads();   // the previous round's writes to blen[] and reads of the tables are done
        tables2((double) blen[lenIdx[3 * k + b1]], (double) blen[lenIdx[3 * k + b2]]);
        __syncthreads();
        for (int64_t p = threadIdx.x; p < nPos; p += WG) {
            ColT c1, c2, o;
            vft_load_col_ml<REAL, NC>(A, n1, p, c1);
            vft_load_col_ml<REAL, NC>(A, n2, p, c2);
            post(c1, c2, A.ratecat[p], o);
            rowB.store(p, o);
        }
        __syncthreads();
        const MlLongSrc<REAL, NC> sA = vft_ml_src<REAL, NC>(nI), sB = vft_ml_src<REAL, NC>(rowB);
        auto negLogLk = [&](double x) -> double {
            if (jc) {
                for (int r = threadIdx.x; r < A.nRates; r += WG) vft_psame_pdiff(x, (double) A.rates[r], pS1[r], pD1[r], A.jcExact != 0);
            } else {
                vft_exp_eigen_rates<REAL, NC>(A, x, minRel, ee1);
            }
            __syncthreads();
            nEval++;
            return -vft_ml_long_total<REAL, NC, WG>(A, sA, sB, jc, pS1, pD1, ee1, nullptr, stage, red, &totS);
        };
        double len = (double) blen[lI];
        if (len < minLen) len = minLen;
        double fBest;
        len = vft_min_branch_length(negLogLk, minLen, len, VFT_MLOPT_MAXLEN, ftol, atol, fBest);
        if (threadIdx.x == 0) blen[lI] = (REAL) len;
    }
    if (evalCount && threadIdx.x == 0) atomicAdd(evalCount, nEval);
    const int64_t rec = recN[k];
    if (rec < 0) return;
    __syncthreads();
    tables2((double) blen[lenIdx[3 * k]], (double) blen[lenIdx[3 * k + 1]]);
    __syncthreads();
    for (int64_t p = threadIdx.x; p < nPos; p += WG) {
        ColT c1, c2, o;
        vft_load_col_ml<REAL, NC>(A, ids[3 * k], p, c1);
        vft_load_col_ml<REAL, NC>(A, ids[3 * k + 1], p, c2);
        post(c1, c2, A.ratecat[p], o);
        vft_store_col_ml<REAL, NC>(A, rec, p, o.w, o.code, o.f);
    }
    if (threadIdx.x == 0) A.mlIs[rec - A.d.nSeqs] = 1;
}

// k_ml_quartet (vft_kernels_ml.h: the modes, the flow of MLQuartetOptimize, the verdict - statement for statement) with the three
// computed profiles of a search in workspace rows: row 0 = the outer posterior of step 0 (AB as "pair 1"), row 1 = the outer posterior of
// steps 1-4 (BCD, ACD, ABD, ABC), row 2 = the inner posterior (CD in steps 0-2, AB in steps 3-4); the fourth piece is the nPos per-site
// likelihoods of mode 0.  ws: one workspace of vft_ml_long_ws_bytes(nPos, NC, sizeof(REAL), 3) bytes per workgroup (grid x * y).
template <typename REAL, int NC>
__global__ __launch_bounds__((MlOptWG<NC>::value)) void k_ml_quartet_long(Arena<REAL> A, const int64_t *ids, const int64_t *lenIdx, REAL *blen,
                                                                          double minLen, double minRel, double ftol, double atol,
                                                                          double closeLimit, int mlAccuracy, int mode, double *loglkOut,
                                                                          double *siteOut, double *lenOut, QuartetNNIResult *nniOut,
                                                                          QuartetNNIState *nniState, unsigned int *evalCount, char *ws,
                                                                          size_t wsStride) {
    constexpr int WG = MlOptWG<NC>::value;
    typedef Col<REAL, NC> ColT;
    typedef MlLongSrc<REAL, NC> Src;
    __shared__ REAL ee[4][VFT_MAXRATES * NC];
    __shared__ double pS[4][VFT_MAXRATES], pD[4][VFT_MAXRATES];
    __shared__ double red[WG / 64];
    __shared__ double stage[VFT_ML_STAGE];
    __shared__ double totS;
    const int64_t k = blockIdx.x;
    const bool jc = A.tmStat == nullptr;
    const int64_t nPos = A.d.nPos;
    char *wsBase = ws + ((size_t) blockIdx.y * gridDim.x + blockIdx.x) * wsStride;
    const MlLongRow<REAL, NC> row0 = vft_ml_long_row<REAL, NC>(wsBase, nPos, 0), row1 = vft_ml_long_row<REAL, NC>(wsBase, nPos, 1),
                              row2 = vft_ml_long_row<REAL, NC>(wsBase, nPos, 2);
    double *site = (double *) (wsBase + 3 * vft_ml_long_row_bytes(nPos, NC, sizeof(REAL)));
    unsigned int nEval = 0;
    // P(t) tables of one branch length into slot s (callers synchronise); posteriorProfile clamps its lengths
    auto table = [&](int s, double len, bool clamp) {
        if (clamp && len < minLen) len = minLen;
        if (jc) {
            for (int r = threadIdx.x; r < A.nRates; r += WG) vft_psame_pdiff(len, (double) A.rates[r], pS[s][r], pD[s][r], A.jcExact != 0);
        } else {
            vft_exp_eigen_rates<REAL, NC>(A, len, minRel, ee[s]);
        }
    };
    // out[p] = posteriorProfile(U, V) of every column with the tables of slots s1 / s2
    auto postRow = [&](const Src &U, const Src &V, int s1, int s2, const MlLongRow<REAL, NC> &out) __attribute__((always_inline)) {
        for (int64_t p = threadIdx.x; p < nPos; p += WG) {
            ColT c1, c2, o;
            U.load(A, p, c1);
            V.load(A, p, c2);
            const int r = A.ratecat[p];
            vft_posterior_col<REAL, NC>(A, c1, c2, jc, pS[s1][r], pD[s1][r], pS[s2][r], pD[s2][r], ee[s1] + r * NC, ee[s2] + r * NC, o.w, o.code, o.f);
            o.vec = o.code == VFT_NOCODE_ && o.w > (REAL) 0;
            out.store(p, o);
        }
    };
    // pairLogLk(X, Y, len) over the workgroup (table slot 0); useSite: multiply the per-site likelihoods in
    auto pairTotal = [&](const Src &X, const Src &Y, double len, bool useSite, bool lead = true) __attribute__((always_inline)) -> double {
        if (lead) __syncthreads();
        table(0, len, false);
        __syncthreads();
        nEval++;
        return vft_ml_long_total<REAL, NC, WG>(A, X, Y, jc, pS[0], pD[0], ee[0], useSite ? site : nullptr, stage, red, &totS);
    };
    auto storeSite = [&](int topo) __attribute__((always_inline)) {
        for (int64_t p = threadIdx.x; p < nPos; p += WG)   // SHSupport takes the logs, NJ.tcc:1134-1137 (glibc's log where the totals are the reference's)
            siteOut[(k * 3 + topo) * nPos + p] = ((!jc || (NC == 4 && A.jcExact)) && site[p] > 0.0) ? vft_glibc_log(site[p]) : log(site[p]);
    };
    auto resetSite = [&]() __attribute__((always_inline)) {
        for (int64_t p = threadIdx.x; p < nPos; p += WG) site[p] = 1.0;
    };
    const int64_t nA = ids[4 * k], nB = ids[4 * k + 1], nC = ids[4 * k + 2], nD = ids[4 * k + 3];
    double base[5];
#pragma unroll
    for (int t = 0; t < 5; t++) base[t] = (double) blen[lenIdx[5 * k + t]];
    double crit[3] = {0, 0, 0};
    const Src sRow0 = vft_ml_src<REAL, NC>(row0), sRow1 = vft_ml_src<REAL, NC>(row1), sRow2 = vft_ml_src<REAL, NC>(row2);

    if (mode == 0) {
        // ---- AB|CD with the lengths as they are: pairLogLk(A,B) + pairLogLk(C,D) + pairLogLk(AB,CD)
        resetSite();
        double tot = 0;
#pragma unroll 1
        for (int half = 0; half < 2; half++) {   // (A, B) -> row 2 = AB, then (C, D) -> row 1 = CD
            const Src X = vft_ml_src<REAL, NC>(half ? nC : nA), Y = vft_ml_src<REAL, NC>(half ? nD : nB);
            tot += pairTotal(X, Y, base[2 * half] + base[2 * half + 1], true);
            __syncthreads();
            table(1, base[2 * half], true);
            table(2, base[2 * half + 1], true);
            __syncthreads();
            postRow(X, Y, 1, 2, half == 0 ? row2 : row1);
        }
        tot += pairTotal(sRow2, sRow1, base[4], true);
        crit[0] = tot;
        storeSite(0);
    }

    // ---- MLQuartetOptimize jobs; pairing t: (a, b | c, d) = (A, B | C, D), (A, C | B, D), (A, D | C, B)
    double len[3][5] = {{base[0], base[1], base[2], base[3], base[4]},
                        {base[0], base[2], base[1], base[3], base[4]},
                        {base[0], base[3], base[2], base[1], base[4]}};
    const int nRounds = mlAccuracy < 2 ? 2 : mlAccuracy;
    int phase = 0, round = 0;
    bool consider1 = true, consider2 = true, star = false;
    if (mode == 2) {   // one pairing of one round: blockIdx.y
        const QuartetNNIState &st = nniState[k];
        const int t = (int) blockIdx.y;
        if (st.done || (t == 1 && !st.consider1) || (t == 2 && !st.consider2)) return;
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const double v = st.len[t][j];
#pragma unroll
            for (int tt = 0; tt < 3; tt++)
                if (tt == t) len[tt][j] = v;
        }
    }
    for (int guard = 0; guard < 64; guard++) {
        int t;
        if (mode == 2) {
            if (guard > 0) break;
            t = (int) blockIdx.y;
        } else if (mode == 0) {
            if (guard == 0) t = 1;
            else if (guard == 1) t = 2;
            else if (guard == 2) {
                t = crit[1] > crit[2] ? 1 : 2;
                if (!(mlAccuracy > 1 || (t == 1 ? crit[1] : crit[2]) > crit[0] - closeLimit)) break;
            } else break;
        } else {
            if (phase == 3) {   // end of a round (NJ.tcc:4961-4983)
                if (mlAccuracy < 2) {
                    if (crit[1] < crit[0] - closeLimit || (len[1][4] <= 2.0 * minLen && crit[1] < crit[0])) consider1 = false;
                    if (crit[2] < crit[0] - closeLimit || (len[2][4] <= 2.0 * minLen && crit[2] < crit[0])) consider2 = false;
                    if (!consider1 && !consider2) break;
                    if (crit[1] > crit[0] + closeLimit && crit[1] > crit[2] + closeLimit) break;
                    if (crit[2] > crit[0] + closeLimit && crit[2] > crit[1] + closeLimit) break;
                }
                if (++round == nRounds) break;
                phase = 0;
            }
            t = phase++;
            if ((t == 1 && !consider1) || (t == 2 && !consider2)) continue;
        }
        const int64_t qa = nA, qb = t == 0 ? nB : t == 1 ? nC : nD, qc = t == 1 ? nB : nC, qd = t == 2 ? nB : nD;
        double L[5];   // len[t] (a copy: runtime indices would send the arrays to scratch memory; written back below)
#pragma unroll
        for (int j = 0; j < 5; j++) {
            L[j] = t == 0 ? len[0][j] : t == 1 ? len[1][j] : len[2][j];
            if (L[j] < minLen) L[j] = minLen;
        }
#define VFT_STORE_L()                                   \
    do {                                                \
        _Pragma("unroll") for (int j = 0; j < 5; j++) { \
            if (t == 0) len[0][j] = L[j];               \
            else if (t == 1) len[1][j] = L[j];          \
            else len[2][j] = L[j];                      \
        }                                               \
    } while (0)
        double negll = 0;
        bool starHere = false;
        for (int step = 0; step < 5; step++) {
            // pair1 -> X, pair2 -> Y for this step; branch optimised: I, A, B, C, D = slots 4, 0, 1, 2, 3
            __syncthreads();
            if (step <= 2) {
                table(1, L[2], true);   // CD = posterior(C, D)
                table(2, L[3], true);
            } else {
                table(1, L[0], true);   // AB = posterior(A, B)
                table(2, L[1], true);
            }
            if (step == 0) {
                table(3, L[0], true);
                table(0, L[1], true);
            } else {
                // outer posterior: (B, CD: lB, lI) (A, CD: lA, lI) (AB, D: lI, lD) (AB, C: lI, lC)
                table(3, step == 1 ? L[1] : step == 2 ? L[0] : L[4], true);
                table(0, step <= 2 ? L[4] : step == 3 ? L[3] : L[2], true);
            }
            __syncthreads();
            // the inner posterior: CD (steps 0-2) or AB (steps 3, 4) -> row 2
            postRow(vft_ml_src<REAL, NC>(step <= 2 ? qc : qa), vft_ml_src<REAL, NC>(step <= 2 ? qd : qb), 1, 2, row2);
            // the outer posterior (a thread reads row 2 where it wrote it): (A, B) -> pair 1 = AB in row 0, pair 2 = CD | (B, CD) resp.
            // (A, CD) -> pair 2 = BCD / ACD in row 1, pair 1 = A / B | (AB, D) resp. (AB, C) -> pair 2 = ABD / ABC in row 1, pair 1 = C / D
            Src X, Y;
            if (step == 0) {
                postRow(vft_ml_src<REAL, NC>(qa), vft_ml_src<REAL, NC>(qb), 3, 0, row0);
                X = sRow0;
                Y = sRow2;
            } else {
                if (step <= 2) postRow(vft_ml_src<REAL, NC>(step == 1 ? qb : qa), sRow2, 3, 0, row1);
                else postRow(sRow2, vft_ml_src<REAL, NC>(step == 3 ? qd : qc), 3, 0, row1);
                X = vft_ml_src<REAL, NC>(step == 1 ? qa : step == 2 ? qb : step == 3 ? qc : qd);
                Y = sRow1;
            }
            __syncthreads();   // the posteriors above are done with the tables
            auto negLogLk = [&](double x) -> double { return -pairTotal(X, Y, x, false, false); };
            const int slot = step == 0 ? 4 : step - 1;
            {
                const double found = vft_min_branch_length(negLogLk, minLen, vft_sel_get<double, 5>(L, slot), VFT_MLOPT_MAXLEN, ftol, atol, negll);
                vft_sel_set<double, 5>(L, slot, found);
            }
            if (step == 0 && mode != 0 && t == 0) {
                // star topology test (NJ.tcc:1691-1700): is the internal branch worth more than closeLogLkLimit?
                const double loglkStar = -negLogLk(minLen);
                if (loglkStar < -negll - closeLimit) {
                    starHere = true;
                    break;
                }
            }
        }
        if (starHere) {
            // -negloglk + pairLogLk(A, B, lA + lB) + pairLogLk(C, D, lC + lD)
            double tot = -negll + pairTotal(vft_ml_src<REAL, NC>(qa), vft_ml_src<REAL, NC>(qb), L[0] + L[1], false);
            tot += pairTotal(vft_ml_src<REAL, NC>(qc), vft_ml_src<REAL, NC>(qd), L[2] + L[3], false);
            crit[0] = tot;
            crit[1] = crit[2] = -1e20;
            star = true;
            VFT_STORE_L();
            break;
        }
        // total: pairLogLk(ABC, D) (= the last search's optimum) + pairLogLk(AB, C, lI + lC) + pairLogLk(A, B, lA + lB);
        // D = the node qd, ABC = row 1, AB = row 2 here
        const bool sitep = mode == 0;
        if (sitep) {
            resetSite();
            pairTotal(sRow1, vft_ml_src<REAL, NC>(qd), L[3], true);
        }
        double tot = -negll;
        tot += pairTotal(sRow2, vft_ml_src<REAL, NC>(qc), L[4] + L[2], sitep);
        tot += pairTotal(vft_ml_src<REAL, NC>(qa), vft_ml_src<REAL, NC>(qb), L[0] + L[1], sitep);
        vft_sel_set<double, 3>(crit, t, tot);
        if (sitep) storeSite(t);
        VFT_STORE_L();
    }
#undef VFT_STORE_L
    if (threadIdx.x != 0) return;
    if (evalCount) atomicAdd(evalCount, nEval);
    if (mode == 2) {
        QuartetNNIState &st = nniState[k];
        const int t = (int) blockIdx.y;
        st.crit[t] = vft_sel_get<double, 3>(crit, t);
#pragma unroll
        for (int j = 0; j < 5; j++) st.len[t][j] = t == 0 ? len[0][j] : t == 1 ? len[1][j] : len[2][j];
        if (t == 0) st.star = star ? 1 : 0;
        return;
    }
    if (mode == 0) {
#pragma unroll
        for (int t = 0; t < 3; t++) loglkOut[3 * k + t] = crit[t];
        if (lenOut) {
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int j = 0; j < 5; j++) lenOut[(2 * k + t) * 5 + j] = len[1 + t][j];
        }
        return;
    }
    // MLQuartetNNI's verdict (NJ.tcc:4989-5003) and DoNNI's branch-length update (NJ.tcc:5889-5915)
    int choice = 0;
    if (!star) {
        if (crit[1] > crit[0] && crit[1] > crit[2]) choice = 1;
        else if (crit[2] > crit[0] && crit[2] > crit[1]) choice = 2;
    }
    const int64_t *li = lenIdx + 5 * k;
    if (star) {
        blen[li[4]] = (REAL) len[0][4];
    } else {
        double L[5];
#pragma unroll
        for (int j = 0; j < 5; j++) L[j] = choice == 0 ? len[0][j] : choice == 1 ? len[1][j] : len[2][j];
        blen[li[0]] = (REAL) L[0];                                   // A
        blen[li[1]] = (REAL) (choice == 0 ? L[1] : choice == 1 ? L[2] : L[3]);   // B
        blen[li[2]] = (REAL) (choice == 0 ? L[2] : choice == 1 ? L[1] : L[2]);   // C
        blen[li[3]] = (REAL) (choice == 0 ? L[3] : choice == 1 ? L[3] : L[1]);   // D
        blen[li[4]] = (REAL) L[4];
    }
    QuartetNNIResult r;
    r.criteria[0] = crit[0];
    r.criteria[1] = crit[1];
    r.criteria[2] = crit[2];
    r.choice = choice;
    r.star = star ? 1 : 0;
    nniOut[k] = r;
}

#define VFT_ML_LONG_INSTANCES(PFX)                                                                                                      \
    PFX template __global__ void k_ml_node_lengths_long<float, 4>(Arena<float>, const int64_t *, const int64_t *, const int64_t *, float *, double, \
                                                                  double, double, double, unsigned int *, char *, size_t);             \
    PFX template __global__ void k_ml_node_lengths_long<float, 20>(Arena<float>, const int64_t *, const int64_t *, const int64_t *, float *,       \
                                                                   double, double, double, double, unsigned int *, char *, size_t);     \
    PFX template __global__ void k_ml_node_lengths_long<double, 4>(Arena<double>, const int64_t *, const int64_t *, const int64_t *, double *,     \
                                                                   double, double, double, double, unsigned int *, char *, size_t);     \
    PFX template __global__ void k_ml_node_lengths_long<double, 20>(Arena<double>, const int64_t *, const int64_t *, const int64_t *, double *,    \
                                                                    double, double, double, double, unsigned int *, char *, size_t);    \
    PFX template __global__ void k_ml_quartet_long<float, 4>(Arena<float>, const int64_t *, const int64_t *, float *, double, double, double,      \
                                                             double, double, int, int, double *, double *, double *, QuartetNNIResult *,          \
                                                             QuartetNNIState *, unsigned int *, char *, size_t);                                   \
    PFX template __global__ void k_ml_quartet_long<float, 20>(Arena<float>, const int64_t *, const int64_t *, float *, double, double, double,     \
                                                              double, double, int, int, double *, double *, double *, QuartetNNIResult *,         \
                                                              QuartetNNIState *, unsigned int *, char *, size_t);                                  \
    PFX template __global__ void k_ml_quartet_long<double, 4>(Arena<double>, const int64_t *, const int64_t *, double *, double, double, double,   \
                                                              double, double, int, int, double *, double *, double *, QuartetNNIResult *,         \
                                                              QuartetNNIState *, unsigned int *, char *, size_t);                                  \
    PFX template __global__ void k_ml_quartet_long<double, 20>(Arena<double>, const int64_t *, const int64_t *, double *, double, double, double,  \
                                                               double, double, int, int, double *, double *, double *, QuartetNNIResult *,        \
                                                               QuartetNNIState *, unsigned int *, char *, size_t);
