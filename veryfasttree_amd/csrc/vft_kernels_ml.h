// ML-phase kernels: P(t) tables, pairLogLk and posteriorProfile, batched over independent node pairs
// (one tree level of recomputeMLProfiles / treeLogLk is one launch).
//
// P(t) is never materialised in HBM: each workgroup builds the tables its pair needs — pSame/pDiff per rate
// category for Jukes-Cantor (NJ.tcc:2005-2018), exp(eigenvalue * rate * length) per (category, code) for matrix
// models (NJ.tcc:2020-2038) — in LDS, next to the 4x4 / 20x20 codeFreq / eigeninv matrices.
#pragma once
#include "vft_device.h"
#include "vft_kernels_profile.h"
#include "vft_glibc_log.h"
#include <type_traits>

#define VFT_ML_WG 128
#define VFT_ML_STAGE 2048   /* columns staged in LDS per pass of the ordered likelihood total */
#define VFT_MAXRATES 64
#define VFT_LK_UNDERFLOW 1.0e-4              /* Constants.h:13 */
#define VFT_LK_UNDERFLOW_INV 1.0e4           /* Constants.h:14 */
#define VFT_LOG_LK_UNDERFLOW 9.21034037197618 /* Constants.h:15 */

// tools-only build (-DVFT_ML_TIMING): thread 0 of workgroup 0 of k_ml_quartet accumulates the clock ticks (100 MHz) of its phases
#ifdef VFT_ML_TIMING
static __device__ unsigned long long vftMlTicks[16];
#define VFT_ML_TICK(k)                                                                  \
    do {                                                                                \
        if (threadIdx.x == 0) {                                                         \
            const unsigned long long now_ = wall_clock64();                             \
            atomicAdd(&vftMlTicks[k], now_ - mlTick_);                                  \
            mlTick_ = now_;                                                             \
        }                                                                               \
    } while (0)
#else
#define VFT_ML_TICK(k) do { } while (0)
#endif

// expEigenRates (NJ.tcc:2020-2038, NDEBUG branch with fastexp level 0) into LDS: out[r*NC + j]
template <typename REAL, int NC>
__device__ __forceinline__ void vft_exp_eigen_rates(const Arena<REAL> &A, double length, double minRel, REAL *out) {
    for (int t = threadIdx.x; t < A.nRates * NC; t += blockDim.x) {
        const int r = t / NC, j = t % NC;
        double relLen = length * (double) A.rates[r];
        if (relLen < minRel) relLen = minRel;
        const REAL rl = (REAL) relLen;           // vector_multiply_by takes numeric_t
        const REAL x = A.tmEigenval[j] * rl;
        out[t] = (REAL) vft_glibc_exp((double) x);   // fastexp level 0 = libm's exp, bit for bit (vft_glibc_log.h)
    }
}

// exact: libm's exp bit for bit (vft_glibc_log.h), as the matrix models' tables take it - Arena::jcExact
__device__ __forceinline__ void vft_psame_pdiff(double length, double rate, double &pSame, double &pDiff, bool exact = false) {
    const double x = (-4.0 / 3.0) * fabs(length * rate);
    pSame = 0.25 + 0.75 * (exact ? vft_glibc_exp(x) : exp(x));
    pDiff = (1.0 - pSame) / 3.0;
}

// frequency vector of a column in eigen-space for matrix models: the stored vector, or codeFreq[code] (row NC is
// the gap row), mixed with the gap vector when 0 < w < 1 (NJ.tcc:1284-1300, 2283-2303)
template <typename REAL, int NC>
__device__ __forceinline__ void vft_model_freq(const Arena<REAL> &A, const Col<REAL, NC> &c, bool mixAlways, REAL *f) {
    const double w = (double) c.w;
    if (c.vec) {
#pragma unroll
        for (int k = 0; k < NC; k++) f[k] = c.f[k];
        if (!mixAlways) return;   // posteriorProfile mixes only code columns (NJ.tcc:2283-2292)
    } else {
        const int row = c.code == VFT_NOCODE_ ? NC : c.code;
#pragma unroll
        for (int k = 0; k < NC; k++) f[k] = A.tmCodeFreq[row * NC + k];
    }
    if (w > 0.0 && w < 1.0) {
#pragma unroll
        for (int k = 0; k < NC; k++) f[k] = (REAL) (w * (double) f[k] + (1.0 - w) * (double) A.tmCodeFreq[NC * NC + k]);
    }
}

// One column of pairLogLk (NJ.tcc:1202-1266 Jukes-Cantor, :1267-1439 matrix models): the likelihood of the column,
// or false for gap against gap under a matrix model (likelihood 1, NJ.tcc:1277-1281).  pSame/pDiff resp. ee (the
// exp(eigenvalue * rate * length) row) are those of the column's rate category.
template <typename REAL, int NC>
__device__ __forceinline__ bool vft_pair_lk_col(const Arena<REAL> &A, const Col<REAL, NC> &c1, const Col<REAL, NC> &c2,
                                                bool jc, double pSame, double pDiff, const REAL *ee, double &lkAB) {
    lkAB = 0;
    if (jc) {
        const double wA = (double) c1.w, wB = (double) c2.w;
        if (!c1.vec && !c2.vec) {
            if (c1.code == VFT_NOCODE_) lkAB = 0.25;
            else if (c2.code == VFT_NOCODE_) lkAB = 0.25;
            else if (c1.code == c2.code) lkAB = pSame * wA * wB + 0.25 * (1 - wA * wB);
            else lkAB = pDiff * wA * wB + 0.25 * (1 - wA * wB);
        } else if (!c1.vec) {
            if (c1.code == VFT_NOCODE_) lkAB = 0.25;
            else lkAB = wA * (pDiff + (double) vft_pick<REAL, NC>(c2.f, c1.code) * (pSame - pDiff)) + (1.0 - wA) * 0.25;
        } else if (!c2.vec) {
            if (c2.code == VFT_NOCODE_) lkAB = 0.25;
            else lkAB = wB * (pDiff + (double) vft_pick<REAL, NC>(c1.f, c2.code) * (pSame - pDiff)) + (1.0 - wB) * 0.25;
        } else {
#pragma unroll
            for (int j = 0; j < (NC < 4 ? NC : 4); j++) {
                const REAL om = (REAL) 1 - c1.f[j];   // int - numeric_t is numeric_t, NJ.tcc:1253
                lkAB += (double) c2.f[j] * ((double) c1.f[j] * pSame + (double) om * pDiff);
            }
        }
        return true;
    }
    if (c1.w == 0 && c2.w == 0 && c1.code == VFT_NOCODE_ && c2.code == VFT_NOCODE_) return false;
    REAL fA[NC], fB[NC];
    vft_model_freq<REAL, NC>(A, c1, true, fA);
    vft_model_freq<REAL, NC>(A, c2, true, fB);
    if (NC == 4) {
#pragma unroll
        for (int j = 0; j < NC; j++) {
            REAL pr = ee[j] * fA[j];   // numeric_t triple product, NJ.tcc:1305
            pr = pr * fB[j];
            lkAB += (double) pr;
        }
    } else {
        REAL e[NC];
#pragma unroll
        for (int j = 0; j < NC; j++) e[j] = ee[j];
        lkAB = (double) vft_red4_mul3<REAL, NC>(e, fA, fB);   // NJ.tcc:1359
    }
    return true;
}

// the reference's running product with its underflow rescaling (NJ.tcc:1257-1262, :1314-1321)
// (lk > 0: the reference asserts lkAB > 0 and would spin forever otherwise; profiles that are not in the model's
//  eigenbasis can produce that, and a kernel must terminate)
__device__ __forceinline__ void vft_lk_accumulate(double lkAB, bool jc, double &lk, double &loglk) {
    lk *= lkAB;
    while (lk < VFT_LK_UNDERFLOW && lk > 0) {
        lk *= VFT_LK_UNDERFLOW_INV;
        loglk -= VFT_LOG_LK_UNDERFLOW;
    }
    if (!jc) {
        while (lk > VFT_LK_UNDERFLOW_INV) {
            lk *= VFT_LK_UNDERFLOW;
            loglk += VFT_LOG_LK_UNDERFLOW;
        }
    }
}

// The reference's total in the reference's order (NJ.tcc:1198-1201, :1257-1262 / :1314-1321, :1444): ONE running
// product over all columns in column order with the underflow rescaling, one final log - glibc's log, bit for bit
// (vft_glibc_log.h).  Matrix models need it: Brent's searches use the values arithmetically, in float precision the
// likelihood is rough at the 1e-10 level, and a total that differs in the last place sends a search to another point
// (the cause of the float32 -gtr topology drift of round 1); in double precision the same
// mechanism moved a few lengths in the ninth decimal and a handful of supports.  The per-thread running products of
// the fast path reorder the multiplications; they stay for Jukes-Cantor, where every column is double arithmetic, the
// function is smooth at that scale, and whole runs up to 100 000 taxa have come out byte-identical.
//   stage[0..n): the columns' likelihoods in column order, VFT_LK_SKIP for columns the reference skips (gap against gap
//   under a matrix model); written by all threads, a barrier, then this - called by ONE thread - walks them.
#define VFT_LK_SKIP (-2.0)
__device__ __forceinline__ void vft_lk_chain(const double *stage, int64_t n, bool jc, double &lk, double &loglk) {
#pragma unroll 8
    for (int64_t p = 0; p < n; p++) {
        const double v = stage[p];
        if (v == VFT_LK_SKIP) continue;
        vft_lk_accumulate(v, jc, lk, loglk);
    }
}

__device__ __forceinline__ double vft_lk_finish(double lk, double loglk) {
    return loglk + (lk > 0.0 && lk < 1.0e300 ? vft_glibc_log(lk) : log(lk));
}

// The ordered total for the line-search kernels, result to every thread.  A chain of nPos dependent multiply / compare /
// rescale steps on one lane costs ~36 cycles per column; here the rescaling decisions are taken off the chain:
//   1. every thread stages its columns' likelihoods a_i (1.0 where the reference skips a column: multiplying by 1.0 is
//      exact) and log2 a_i (vft_lk_stage);
//   2. the workgroup decides every rescaling from prefix sums of the logs.  With s_i = log2 of the product of a_0..a_i and
//      T = log2(1e4), the reference's two while loops keep log2(lk) = s_i + n_i T inside [-T, T] by the smallest change
//      of the net rescaling count: n_i = clamp(n_(i-1), lo_i, hi_i), lo_i = ceil((-T - s_i) / T), hi_i = floor((T - s_i) / T)
//      (under a matrix model a column's likelihood can exceed 1 - profiles are scaled by 1 / stat - so both loops fire).
//      Clamps compose into clamps, so all n_i come out of scans: a thread owns a run of C consecutive columns; the prefix
//      sums, the composed clamps and the event counts are each one wave scan plus one exchange through LDS (round 2 had
//      wavefront 0 walk runs of COLS / 64 columns four times: 9.6 us of a 15 us evaluation at 300 protein columns).  A
//      prefix sum closer than 1e-9 (in units of T) to a threshold, a non-positive or non-finite value, or a list overflow
//      sends the call to the plain chain;
//   3. the multiplier list - every a_i followed by one 1e4 (or 1e-4) per rescaling it triggers - is laid out in LDS and
//      ONE lane multiplies it through in order: the reference's sequence of roundings at ~7 cycles per element; a lane
//      of another wavefront applies -/+ LogLkUnderflow once per rescaling in the same order (also a chain of roundings).
struct LkOrderedShared {
    double prod, loglk;
    int irregular, nList, nEvents, pad;
    double waveSum[16];
    int waveLo[16], waveHi[16], waveEv[16];
};

// the clamp interval [l, h] of a column whose prefix log-sum is sI; odd: too close to a threshold to decide here
__device__ __forceinline__ void vft_lk_interval(double sI, int &l, int &h, bool &odd) {
    if (sI == 0.0) {   // nothing but exact ones so far: lk is exactly 1, and -T, 0, T are all inside the closed window
        l = -1;
        h = 1;
        return;
    }
    const double x = sI * (1.0 / 13.287712379549449), fl = floor(x);   // s_i / log2(1e4)
    if (x - fl < 1.0e-9 || fl + 1.0 - x < 1.0e-9 || !(fabs(x) < 1.0e6)) odd = true;
    l = -(int) fl - 1;
    h = l + 1;
}

__device__ __forceinline__ int vft_clampi(int v, int l, int h) { return v < l ? l : (v > h ? h : v); }

// step 1 for one column (every thread, for each of its columns; the total starts with a barrier)
__device__ __forceinline__ void vft_lk_stage(double *stage, double *stageLog, int p, double lkAB) {
    const double a = lkAB == VFT_LK_SKIP ? 1.0 : lkAB;
    stage[p] = a;
    stageLog[p] = log2(a);
}

// step 1 for the CPT columns p = qc + c * CW of a group of LPC lanes that all hold the columns' likelihoods (LPC = 4: lane l takes
// the columns c = l, l + 4, ... - the logarithm is most of the staging, one lane doing all of them left three idle)
template <int CPT, int LPC>
__device__ __forceinline__ void vft_lk_stage_cols(double *stage, double *stageLog, const double (&lk)[CPT], int qc, int ql, int CW, int nPos) {
    if (LPC == 1) {
#pragma unroll
        for (int c = 0; c < CPT; c++) {
            const int p = qc + c * CW;
            if (p < nPos) vft_lk_stage(stage, stageLog, p, lk[c]);
        }
        return;
    }
#pragma unroll
    for (int g = 0; g < (CPT + LPC - 1) / LPC; g++) {
        double v = VFT_LK_SKIP;
#pragma unroll
        for (int l = 0; l < LPC; l++)
            if (g * LPC + l < CPT && ql == l) v = lk[g * LPC + l];
        const int c = g * LPC + ql, p = qc + c * CW;
        if (c < CPT && p < nPos) vft_lk_stage(stage, stageLog, p, v);
    }
}

// (round 3's first form, kept verbatim for the whole-column protein instances: hipcc 7.2 fails on them - "Illegal instruction
// detected: Operand has incorrect register class" - with any variation of this function that was tried)
// steps 2 and 3 over stage[0, nPos) / stageLog[0, nPos); COLS: capacity of the staging arrays (list: COLS * 3 / 2 + 32, events: COLS / 2)
template <int WG, int COLS>
__device__ __forceinline__ double vft_lk_total_staged_v1(double *stage, double *stageLog, double *list, signed char *events,
                                                      LkOrderedShared *sh, int nPos, bool jc) {
    constexpr int LCAP = COLS + COLS / 2;       // capacity of list
    constexpr int ECAP = COLS / 2;              // capacity of events
    constexpr int C = (COLS + WG - 1) / WG;     // columns per thread
    constexpr int NW = WG / 64;
    constexpr int BIG = 1 << 28;
    static_assert(NW <= 16, "LkOrderedShared holds 16 wavefronts");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef VFT_ML_TIMING
    unsigned long long subTick_ = wall_clock64();
#define VFT_ML_SUBTICK1(k)                                                   \
    do {                                                                    \
        if (threadIdx.x == 0) {                                             \
            const unsigned long long now_ = wall_clock64();                 \
            atomicAdd(&vftMlTicks[k], now_ - subTick_);                     \
            subTick_ = now_;                                                \
        }                                                                   \
    } while (0)
#else
#define VFT_ML_SUBTICK1(k) do { } while (0)
#endif
    if (tid == 0) sh->irregular = 0;
    __syncthreads();
    VFT_ML_SUBTICK1(10);
    bool odd = false;
    const int q0 = tid * C < nPos ? tid * C : nPos, q1 = q0 + C < nPos ? q0 + C : nPos;   // this thread's run of columns [q0, q1)
    // prefix sums of the logs: the run, a wave scan, the earlier wavefronts' totals
    double lg[C], lsum = 0;
#pragma unroll
    for (int k = 0; k < C; k++) {
        lg[k] = q0 + k < q1 ? stageLog[q0 + k] : 0.0;
        if (!(fabs(lg[k]) < 1.0e4)) odd = true;   // a <= 0, inf, nan
        lsum += lg[k];
    }
    double incl = lsum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) sh->waveSum[wave] = incl;
    __syncthreads();
    VFT_ML_SUBTICK1(11);
    double excl = 0;
    for (int w = 0; w < wave; w++) excl += sh->waveSum[w];
    excl += incl - lsum;
    // every column's interval, the run as one clamp (cl, ch), an inclusive scan of the clamps (earlier runs first)
    int L[C], H[C], cl = -BIG, ch = BIG;
    {
        double run = 0;
#pragma unroll
        for (int k = 0; k < C; k++) {
            L[k] = -BIG;
            H[k] = BIG;
            if (q0 + k < q1) {
                run += lg[k];
                vft_lk_interval(excl + run, L[k], H[k], odd);
                cl = vft_clampi(cl, L[k], H[k]);
                ch = vft_clampi(ch, L[k], H[k]);
            }
        }
    }
    int il = cl, ih = ch;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int tl = __shfl_up(il, off, 64), th = __shfl_up(ih, off, 64);
        if (lane >= off) {   // (tl, th) covers the columns before those of (il, ih): push its bounds through
            const int nl = vft_clampi(tl, il, ih), nh = vft_clampi(th, il, ih);
            il = nl;
            ih = nh;
        }
    }
    if (lane == 63) {
        sh->waveLo[wave] = il;
        sh->waveHi[wave] = ih;
    }
    int pl = __shfl_up(il, 1, 64), ph = __shfl_up(ih, 1, 64);   // the lanes in front of this one (this wavefront)
    if (lane == 0) {
        pl = -BIG;
        ph = BIG;
    }
    __syncthreads();
    VFT_ML_SUBTICK1(12);
    int nIn = 0;   // the count entering this run (lk starts at 1: n = 0): through the earlier wavefronts, then the earlier lanes
    for (int w = 0; w < wave; w++) nIn = vft_clampi(nIn, sh->waveLo[w], sh->waveHi[w]);
    nIn = vft_clampi(nIn, pl, ph);
    // rescaling events of the run, exclusive scan of their counts
    int ev = 0;
    if (!odd) {
        int n = nIn;
#pragma unroll
        for (int k = 0; k < C; k++) {
            const int n2 = vft_clampi(n, L[k], H[k]);
            ev += n2 > n ? n2 - n : n - n2;
            n = n2;
        }
        if (ev > ECAP) {
            ev = 0;
            odd = true;
        }
    }
    int einc = ev;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(einc, off, 64);
        if (lane >= off) einc += t;
    }
    if (lane == 63) sh->waveEv[wave] = einc;
    if (odd) sh->irregular = 1;   // (every writer stores 1)
    __syncthreads();
    VFT_ML_SUBTICK1(13);
    int eBase = 0, eTotal = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const int v = sh->waveEv[w];
        if (w < wave) eBase += v;
        eTotal += v;
    }
    const bool irregular = sh->irregular != 0 || eTotal > ECAP || nPos + eTotal > LCAP;   // (uniform)
    if (!irregular) {
        // the multiplier list and the event list
        int e = eBase + einc - ev, n = nIn;
#pragma unroll
        for (int k = 0; k < C; k++) {
            if (q0 + k >= q1) continue;
            const int q = q0 + k, n2 = vft_clampi(n, L[k], H[k]);
            list[q + e] = stage[q];
            const int d = n2 - n, cnt = d > 0 ? d : -d;
            for (int r = 0; r < cnt; r++) {
                list[q + e + 1 + r] = d > 0 ? VFT_LK_UNDERFLOW_INV : VFT_LK_UNDERFLOW;
                events[e + r] = d > 0 ? 1 : -1;
            }
            e += cnt;
            n = n2;
        }
        if (tid < 32) list[nPos + eTotal + tid] = 1.0;   // (multiplying by one is exact: the chain runs in whole groups of sixteen)
    }
    __syncthreads();
    VFT_ML_SUBTICK1(14);
    if (irregular) {   // uniform: the plain chain, decisions and all
        if (tid == 0) {
            double lk = 1.0, loglk = 0.0;
            vft_lk_chain(stage, nPos, jc, lk, loglk);
            sh->prod = lk;
            sh->loglk = loglk;
        }
    } else if (tid == 0) {
        // (sixteen multipliers per trip, the next sixteen already on their way from LDS; the list ends with 32 ones)
        const int K = nPos + eTotal;
        double lk = 1.0;
#ifdef VFT_TEST_PLAIN_CHAIN
#pragma unroll 16
        for (int k = 0; k < K; k++) lk *= list[k];
#else
        double cur[16];
#pragma unroll
        for (int u = 0; u < 16; u++) cur[u] = list[u];
#pragma unroll 1
        for (int k0 = 0; k0 < K; k0 += 16) {
            double nxt[16];
#pragma unroll
            for (int u = 0; u < 16; u++) nxt[u] = list[k0 + 16 + u];
#pragma unroll
            for (int u = 0; u < 16; u++) lk *= cur[u];
#pragma unroll
            for (int u = 0; u < 16; u++) cur[u] = nxt[u];
        }
#endif
        sh->prod = lk;
    } else if (tid == 64) {
        double loglk = 0.0;
        for (int r = 0; r < eTotal; r++) {
            if (events[r] > 0) loglk -= VFT_LOG_LK_UNDERFLOW;
            else loglk += VFT_LOG_LK_UNDERFLOW;
        }
        sh->loglk = loglk;
    }
    __syncthreads();
    VFT_ML_SUBTICK1(15);
    return vft_lk_finish(sh->prod, sh->loglk);
}

// steps 2 and 3 over stage[0, nPos) / stageLog[0, nPos); COLS: capacity of the staging arrays (list: COLS * 3 / 2 + 64, events: COLS / 2 + 16)
template <int WG, int COLS>
__device__ __forceinline__ double vft_lk_total_staged(double *stage, double *stageLog, double *list, signed char *events,
                                                      LkOrderedShared *sh, int nPos, bool jc) {
    constexpr int LCAP = COLS + COLS / 2;       // capacity of list
    constexpr int ECAP = COLS / 2;              // capacity of events
    constexpr int C = (COLS + WG - 1) / WG;     // columns per thread
    constexpr int NW = WG / 64;
    constexpr int BIG = 1 << 28;
    static_assert(NW <= 16, "LkOrderedShared holds 16 wavefronts");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef VFT_ML_TIMING
    unsigned long long subTick_ = wall_clock64();
#define VFT_ML_SUBTICK(k)                                                   \
    do {                                                                    \
        if (threadIdx.x == 0) {                                             \
            const unsigned long long now_ = wall_clock64();                 \
            atomicAdd(&vftMlTicks[k], now_ - subTick_);                     \
            subTick_ = now_;                                                \
        }                                                                   \
    } while (0)
#else
#define VFT_ML_SUBTICK(k) do { } while (0)
#endif
    if (tid == 0) sh->irregular = 0;
    __syncthreads();
    VFT_ML_SUBTICK(10);
    bool odd = false;
    const int q0 = tid * C < nPos ? tid * C : nPos, q1 = q0 + C < nPos ? q0 + C : nPos;   // this thread's run of columns [q0, q1)
    // prefix sums of the logs: the run, a wave scan, the earlier wavefronts' totals
    double lg[C], lsum = 0;
#pragma unroll
    for (int k = 0; k < C; k++) {
        lg[k] = q0 + k < q1 ? stageLog[q0 + k] : 0.0;
        if (!(fabs(lg[k]) < 1.0e4)) odd = true;   // a <= 0, inf, nan
        lsum += lg[k];
    }
    double incl = lsum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) sh->waveSum[wave] = incl;
    __syncthreads();
    VFT_ML_SUBTICK(11);
    double excl = 0;
    for (int w = 0; w < wave; w++) excl += sh->waveSum[w];
    excl += incl - lsum;
    // every column's interval, the run as one clamp (cl, ch), an inclusive scan of the clamps (earlier runs first)
    int L[C], H[C], cl = -BIG, ch = BIG;
    {
        double run = 0;
#pragma unroll
        for (int k = 0; k < C; k++) {
            L[k] = -BIG;
            H[k] = BIG;
            if (q0 + k < q1) {
                run += lg[k];
                vft_lk_interval(excl + run, L[k], H[k], odd);
                cl = vft_clampi(cl, L[k], H[k]);
                ch = vft_clampi(ch, L[k], H[k]);
            }
        }
    }
    int il = cl, ih = ch;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int tl = __shfl_up(il, off, 64), th = __shfl_up(ih, off, 64);
        if (lane >= off) {   // (tl, th) covers the columns before those of (il, ih): push its bounds through
            const int nl = vft_clampi(tl, il, ih), nh = vft_clampi(th, il, ih);
            il = nl;
            ih = nh;
        }
    }
    if (lane == 63) {
        sh->waveLo[wave] = il;
        sh->waveHi[wave] = ih;
    }
    int pl = __shfl_up(il, 1, 64), ph = __shfl_up(ih, 1, 64);   // the lanes in front of this one (this wavefront)
    if (lane == 0) {
        pl = -BIG;
        ph = BIG;
    }
    __syncthreads();
    VFT_ML_SUBTICK(12);
    int nIn = 0;   // the count entering this run (lk starts at 1: n = 0): through the earlier wavefronts, then the earlier lanes
    for (int w = 0; w < wave; w++) nIn = vft_clampi(nIn, sh->waveLo[w], sh->waveHi[w]);
    nIn = vft_clampi(nIn, pl, ph);
    // rescaling events of the run, exclusive scan of their counts
    int ev = 0;
    if (!odd) {
        int n = nIn;
#pragma unroll
        for (int k = 0; k < C; k++) {
            const int n2 = vft_clampi(n, L[k], H[k]);
            ev += n2 > n ? n2 - n : n - n2;
            n = n2;
        }
        if (ev > ECAP) {
            ev = 0;
            odd = true;
        }
    }
    int einc = ev;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(einc, off, 64);
        if (lane >= off) einc += t;
    }
    if (lane == 63) sh->waveEv[wave] = einc;
    if (odd) sh->irregular = 1;   // (every writer stores 1)
    __syncthreads();
    VFT_ML_SUBTICK(13);
    int eBase = 0, eTotal = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const int v = sh->waveEv[w];
        if (w < wave) eBase += v;
        eTotal += v;
    }
    const bool irregular = sh->irregular != 0 || eTotal > ECAP || nPos + eTotal > LCAP;   // (uniform)
    if (!irregular) {
        // the multiplier list and the event list
        int e = eBase + einc - ev, n = nIn;
#pragma unroll
        for (int k = 0; k < C; k++) {
            if (q0 + k >= q1) continue;
            const int q = q0 + k, n2 = vft_clampi(n, L[k], H[k]);
            list[q + e] = stage[q];
            const int d = n2 - n, cnt = d > 0 ? d : -d;
            for (int r = 0; r < cnt; r++) {
                list[q + e + 1 + r] = d > 0 ? VFT_LK_UNDERFLOW_INV : VFT_LK_UNDERFLOW;
                events[e + r] = d > 0 ? 1 : -1;
            }
            e += cnt;
            n = n2;
        }
        if (tid < 64) list[nPos + eTotal + tid] = 1.0;   // (multiplying by one is exact: the chain runs in whole groups of sixteen)
        if (tid >= 64 && tid < 80) events[eTotal + tid - 64] = 0;
    }
    __syncthreads();
    VFT_ML_SUBTICK(14);
#ifdef VFT_ML_TIMING
    if (tid == 0) {
        atomicAdd(&vftMlTicks[7], irregular ? 1ull : 0ull);
        atomicAdd(&vftMlTicks[6], (unsigned long long) eTotal);
    }
#endif
    if (irregular) {   // uniform: the plain chain, decisions and all
        if (tid == 0) {
            double lk = 1.0, loglk = 0.0;
            vft_lk_chain(stage, nPos, jc, lk, loglk);
            sh->prod = lk;
            sh->loglk = loglk;
        }
    } else if (tid == 0) {
        // Sixteen multipliers per step from two register buffers in turn: the loads of one are issued before the other is
        // multiplied through, and the scheduling barriers keep it that way (left to itself the compiler loads sixteen values,
        // waits for them and multiplies - an LDS latency per sixteen 5.6-cycle multiplications).  The list ends with 64 ones.
        const int K = nPos + eTotal;
        double lk = 1.0;
        {
            double bufA[16], bufB[16];
#pragma unroll
            for (int u = 0; u < 16; u++) bufA[u] = list[u];
#pragma unroll 1
            for (int k0 = 0; k0 < K; k0 += 32) {
#pragma unroll
                for (int u = 0; u < 16; u++) bufB[u] = list[k0 + 16 + u];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 16; u++) lk *= bufA[u];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 16; u++) bufA[u] = list[k0 + 32 + u];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 16; u++) lk *= bufB[u];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        sh->prod = lk;
    } else if (tid == 64) {
        // (sixteen events per LDS read: one read per event made this the longer of the two chains - 100 events x an LDS latency)
        double loglk = 0.0;
        {
            // (no branches: an event is +1 / -1, the sixteen bytes behind the last one are 0, and -e * LogLkUnderflow is exactly
            //  -/+ LogLkUnderflow or a zero that changes nothing - as compares and branches on these uniform values the loop was
            //  three taken branches per event, the longer of the two chains by far)
#pragma unroll 1
            for (int r0 = 0; r0 < eTotal; r0 += 16) {
                signed char e16[16];
#pragma unroll
                for (int u = 0; u < 16; u++) e16[u] = events[r0 + u];
#pragma unroll
                for (int u = 0; u < 16; u++) loglk += (double) (-(int) e16[u]) * VFT_LOG_LK_UNDERFLOW;
            }
        }
        sh->loglk = loglk;
    }
    __syncthreads();
    VFT_ML_SUBTICK(15);
    return vft_lk_finish(sh->prod, sh->loglk);
}

// the same for threads that own the columns p = tid + c * WG, c < CPT
template <int WG, int CPT>
__device__ __forceinline__ double vft_lk_total_ordered(double *stage, double *stageLog, double *list, signed char *events,
                                                       LkOrderedShared *sh, const double (&lkAB)[CPT], int64_t nPos64, bool jc) {
    const int nPos = (int) nPos64;
#pragma unroll
    for (int c = 0; c < CPT; c++) {
        const int p = (int) threadIdx.x + c * WG;
        if (p < nPos) vft_lk_stage(stage, stageLog, p, lkAB[c]);
    }
    return vft_lk_total_staged<WG, CPT * WG>(stage, stageLog, list, events, sh, nPos, jc);
}

// pairLogLk (NJ.tcc:1192-1447).  One workgroup per pair, threads over columns.  Each thread keeps the
// reference's running product with underflow rescaling for its own columns; the per-thread log-products are
// then summed (wave shuffles + LDS).  Only the order of that final sum differs from the reference.
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_ML_WG) void k_pair_loglk(Arena<REAL> A, const int64_t *aN, const int64_t *bN,
                                                          const double *length, int64_t n, double minRel,
                                                          double *loglkOut, double *siteLk) {
    __shared__ REAL expeig[VFT_MAXRATES * NC];
    __shared__ double pS[VFT_MAXRATES], pD[VFT_MAXRATES];
    __shared__ double red[VFT_ML_WG / 64];
    const int64_t k = blockIdx.x;
    const int64_t a = aN[k], b = bN[k];
    const double len = length[k];
    const bool jc = A.tmStat == nullptr;
    if (jc) {
        for (int r = threadIdx.x; r < A.nRates; r += blockDim.x) vft_psame_pdiff(len, (double) A.rates[r], pS[r], pD[r], A.jcExact != 0);
    } else {
        vft_exp_eigen_rates<REAL, NC>(A, len, minRel, expeig);
    }
    __syncthreads();
    double lk = 1.0, loglk = 0.0;
    if (!jc || (NC == 4 && A.jcExact)) {
        // matrix model (and Jukes-Cantor to the last bit, Arena::jcExact): the reference's ordered total (vft_lk_chain), VFT_ML_STAGE columns at a time
        __shared__ double stage[VFT_ML_STAGE];
        for (int64_t p0 = 0; p0 < A.d.nPos; p0 += VFT_ML_STAGE) {
            const int64_t cnt = A.d.nPos - p0 < VFT_ML_STAGE ? A.d.nPos - p0 : VFT_ML_STAGE;
            for (int64_t q = threadIdx.x; q < cnt; q += blockDim.x) {
                const int64_t p = p0 + q;
                Col<REAL, NC> c1, c2;
                vft_load_col_ml<REAL, NC>(A, a, p, c1);
                vft_load_col_ml<REAL, NC>(A, b, p, c2);
                const int r = A.ratecat[p];
                double lkAB;
                const bool has = vft_pair_lk_col<REAL, NC>(A, c1, c2, jc, pS[r], pD[r], expeig + r * NC, lkAB);
                if (siteLk) siteLk[k * A.d.nPos + p] = has ? lkAB : 1.0;
                stage[q] = has ? lkAB : VFT_LK_SKIP;
            }
            __syncthreads();
            if (threadIdx.x == 0) vft_lk_chain(stage, cnt, jc, lk, loglk);
            __syncthreads();
        }
        if (threadIdx.x == 0) loglkOut[k] = vft_lk_finish(lk, loglk);
        return;
    }
    for (int64_t p = threadIdx.x; p < A.d.nPos; p += blockDim.x) {
        Col<REAL, NC> c1, c2;
        vft_load_col_ml<REAL, NC>(A, a, p, c1);
        vft_load_col_ml<REAL, NC>(A, b, p, c2);
        const int r = A.ratecat[p];
        double lkAB;
        if (!vft_pair_lk_col<REAL, NC>(A, c1, c2, jc, pS[r], pD[r], expeig + r * NC, lkAB)) {
            if (siteLk) siteLk[k * A.d.nPos + p] = 1.0;
            continue;
        }
        if (siteLk) siteLk[k * A.d.nPos + p] = lkAB;
        vft_lk_accumulate(lkAB, jc, lk, loglk);
    }
    double part = loglk + log(lk);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0;
        for (int w = 0; w < VFT_ML_WG / 64; w++) tot += red[w];
        loglkOut[k] = tot;
    }
}

// One column of posteriorProfile (NJ.tcc:2185-2260 Jukes-Cantor incl. the "simple profile" shortcuts, :2262-2434 matrix
// models, exact ML): weight, code and - when code == NOCODE and weight > 0 - the frequency vector of the parent.
// PS/PD resp. e1/e2 belong to the column's rate category and the two branch lengths.
template <typename REAL, int NC>
__device__ __forceinline__ void vft_posterior_col(const Arena<REAL> &A, const Col<REAL, NC> &c1, const Col<REAL, NC> &c2,
                                                  bool jc, double PS1, double PD1, double PS2, double PD2,
                                                  const REAL *e1, const REAL *e2, REAL &wo, int &co, REAL *f) {
    wo = (REAL) 1.0;
    co = VFT_NOCODE_;
#pragma unroll
    for (int j = 0; j < NC; j++) f[j] = 0;
    if (jc) {
        const double w1 = (double) c1.w, w2 = (double) c2.w;
        bool simple = false;
        if (!c1.vec && !c2.vec) {
            if (c1.code == VFT_NOCODE_ && c2.code == VFT_NOCODE_) {
                co = VFT_NOCODE_;
                wo = 0;
                simple = true;
            } else if (c1.code == VFT_NOCODE_) {
                co = c2.code;
                wo = (REAL) (w2 * (PS2 - PD2));
                simple = true;
            } else if (c2.code == VFT_NOCODE_) {
                co = c1.code;
                wo = (REAL) (w1 * (PS1 - PD1));
                simple = true;
            } else if (c1.code == c2.code) {
                co = c1.code;
                const double f12code = (w1 * PS1 + (1 - w1) * 0.25) * (w2 * PS2 + (1 - w2) * 0.25);
                const double f12other = (w1 * PD1 + (1 - w1) * 0.25) * (w2 * PD2 + (1 - w2) * 0.25);
                const double pcode = f12code / (f12code + 3 * f12other);
                wo = (REAL) ((pcode - 0.25) * 4.0 / 3.0);
                if ((double) wo < 1e-6) wo = (REAL) 1e-6;
                simple = true;
            }
        }
        if (!simple) {
            REAL g1[NC], g2[NC];
            if (c1.vec) {
#pragma unroll
                for (int j = 0; j < NC; j++) g1[j] = c1.f[j];
            } else {
#pragma unroll
                for (int j = 0; j < NC; j++) {
                    REAL v = (REAL) ((1 - w1) * 0.25);
                    if (j == c1.code) v = (REAL) ((double) v + w1);
                    g1[j] = v;
                }
            }
            if (c2.vec) {
#pragma unroll
                for (int j = 0; j < NC; j++) g2[j] = c2.f[j];
            } else {
#pragma unroll
                for (int j = 0; j < NC; j++) {
                    REAL v = (REAL) ((1 - w2) * 0.25);
                    if (j == c2.code) v = (REAL) ((double) v + w2);
                    g2[j] = v;
                }
            }
            co = VFT_NOCODE_;
            wo = (REAL) 1.0;
            double lkAB = 0;
#pragma unroll
            for (int j = 0; j < NC; j++) {
                f[j] = (REAL) (((double) g1[j] * PS1 + (1.0 - (double) g1[j]) * PD1) *
                               ((double) g2[j] * PS2 + (1.0 - (double) g2[j]) * PD2));
                lkAB += (double) f[j];
            }
            const double inv = 1.0 / lkAB;
#pragma unroll
            for (int j = 0; j < NC; j++) f[j] = (REAL) ((double) f[j] * inv);
        }
    } else {
        if (c1.code == VFT_NOCODE_ && c2.code == VFT_NOCODE_ && c1.w == 0 && c2.w == 0) {
            wo = 0;   // gap with gap (NJ.tcc:2267-2272)
        } else {
            REAL f1[NC], f2[NC], fM1[NC], fM2[NC], fPost[NC];
            vft_model_freq<REAL, NC>(A, c1, false, f1);
            vft_model_freq<REAL, NC>(A, c2, false, f2);
#pragma unroll
            for (int j = 0; j < NC; j++) {
                fM1[j] = f1[j] * e1[j];
                fM2[j] = f2[j] * e2[j];
            }
            if (NC == 4) {
#pragma unroll
                for (int j = 0; j < NC; j++) {
                    double out1 = 0, out2 = 0;
#pragma unroll
                    for (int q = 0; q < NC; q++) {
                        const REAL cf = A.tmCodeFreq[j * NC + q];
                        const REAL p1 = fM1[q] * cf, p2 = fM2[q] * cf;
                        out1 += (double) p1;
                        out2 += (double) p2;
                    }
                    fPost[j] = (REAL) (out1 * out2 * (double) A.tmStatInv[j]);
                }
                double tot = 0;
#pragma unroll
                for (int j = 0; j < NC; j++) tot += (double) fPost[j];
                const double inv = 1.0 / tot;
#pragma unroll
                for (int j = 0; j < NC; j++) fPost[j] = (REAL) ((double) fPost[j] * inv);
                // matrix_by_vector4(eigeninvT, fPost, fOut) in the SSE/AVX order (SSE128Operations.tcc:250-262)
#pragma unroll
                for (int cidx = 0; cidx < NC; cidx++) {
                    REAL o = 0;
#pragma unroll
                    for (int j = 0; j < NC; j++) {
                        const REAL pr = fPost[j] * A.tmEigenInvT[j * NC + cidx];
                        o = o + pr;
                    }
                    f[cidx] = o;
                }
            } else {
                for (int j = 0; j < NC; j++) {
                    REAL cf[NC];
#pragma unroll
                    for (int q = 0; q < NC; q++) cf[q] = A.tmCodeFreq[j * NC + q];
                    const REAL d1 = vft_red4_mul<REAL, NC>(fM1, cf), d2 = vft_red4_mul<REAL, NC>(fM2, cf);
                    REAL value = d1 * d2;
                    value = value * A.tmStatInv[j];
                    fPost[j] = value >= 0 ? value : (REAL) 0;
                }
                const double tot = (double) vft_red4_sum<REAL, NC>(fPost);
                const REAL invr = (REAL) (1.0 / tot);
#pragma unroll
                for (int j = 0; j < NC; j++) fPost[j] = fPost[j] * invr;
                for (int j = 0; j < NC; j++) {
                    REAL ei[NC];
#pragma unroll
                    for (int q = 0; q < NC; q++) ei[q] = A.tmEigenInv[j * NC + q];
                    f[j] = vft_red4_mul<REAL, NC>(fPost, ei);
                }
            }
        }
    }
}

// posteriorProfile (NJ.tcc:2137-2447, exact ML).  grid.y = triple index, threads over columns.
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_ML_WG) void k_posterior(Arena<REAL> A, const int64_t *outN, const int64_t *aN,
                                                         const int64_t *bN, const double *len1A, const double *len2A,
                                                         double minLen, double minRel,
                                                         REAL *stash /* null: write the node's dense ML row */) {
    __shared__ REAL ee1[VFT_MAXRATES * NC], ee2[VFT_MAXRATES * NC];
    __shared__ double pS1[VFT_MAXRATES], pD1[VFT_MAXRATES], pS2[VFT_MAXRATES], pD2[VFT_MAXRATES];
    const int64_t k = blockIdx.y;
    double len1 = len1A[k], len2 = len2A[k];
    if (len1 < minLen) len1 = minLen;
    if (len2 < minLen) len2 = minLen;
    const bool jc = A.tmStat == nullptr;
    if (jc) {
        for (int r = threadIdx.x; r < A.nRates; r += blockDim.x) {
            vft_psame_pdiff(len1, (double) A.rates[r], pS1[r], pD1[r], A.jcExact != 0);
            vft_psame_pdiff(len2, (double) A.rates[r], pS2[r], pD2[r], A.jcExact != 0);
        }
    } else {
        vft_exp_eigen_rates<REAL, NC>(A, len1, minRel, ee1);
        vft_exp_eigen_rates<REAL, NC>(A, len2, minRel, ee2);
    }
    __syncthreads();
    const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.d.nPos) return;
    Col<REAL, NC> c1, c2;
    vft_load_col_ml<REAL, NC>(A, aN[k], p, c1);
    vft_load_col_ml<REAL, NC>(A, bN[k], p, c2);
    const int r = A.ratecat[p];
    REAL wo;
    int co;
    REAL f[NC];
    vft_posterior_col<REAL, NC>(A, c1, c2, jc, pS1[r], pD1[r], pS2[r], pD2[r], ee1 + r * NC, ee2 + r * NC, wo, co, f);
    if (stash) {
        vft_stash_col<REAL, NC>(A, outN[k], p, wo, co, f, stash + (k * A.d.nPos + p) * (NC + 1));
    } else {
        vft_store_col_ml<REAL, NC>(A, outN[k], p, wo, co, f);
        if (p == 0) A.mlIs[outN[k] - A.d.nSeqs] = 1;
    }
}

// A CHAIN of n posteriorProfile calls in order where later ones may read earlier results (up-profiles down a path, a
// node and then its parent after an NNI), the branch lengths read from the device's branchlength[] when the op runs:
// as for k_average_chain a column depends on that column of its inputs only, so a thread takes its columns through the
// whole chain; the P(t) tables of every op are rebuilt per workgroup.  Rows only; flags by k_mark_rows afterwards.
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_ML_WG) void k_posterior_chain(Arena<REAL> A, const int64_t *outN, const int64_t *aN,
                                                               const int64_t *bN, const int64_t *liA, const int64_t *liB,
                                                               const uint8_t *direct, int32_t n, const REAL *blen,
                                                               double minLen, double minRel, const int32_t *chainOff) {
    __shared__ REAL ee1[VFT_MAXRATES * NC], ee2[VFT_MAXRATES * NC];
    __shared__ double pS1[VFT_MAXRATES], pD1[VFT_MAXRATES], pS2[VFT_MAXRATES], pD2[VFT_MAXRATES];
    const bool jc = A.tmStat == nullptr;
    const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = p < A.d.nPos;
    const int r = live ? A.ratecat[p] : 0;
    // several independent chains in one launch (blockIdx.y; the lanes of a subtree schedule): ops [chainOff[y], chainOff[y + 1])
    const int32_t k0 = chainOff ? chainOff[blockIdx.y] : 0, k1 = chainOff ? chainOff[blockIdx.y + 1] : n;
    for (int32_t k = k0; k < k1; k++) {
        double len1 = (double) blen[liA[k]], len2 = (double) blen[liB[k]];
        if (len1 < minLen) len1 = minLen;
        if (len2 < minLen) len2 = minLen;
        __syncthreads();
        if (jc) {
            for (int q = threadIdx.x; q < A.nRates; q += blockDim.x) {
                vft_psame_pdiff(len1, (double) A.rates[q], pS1[q], pD1[q], A.jcExact != 0);
                vft_psame_pdiff(len2, (double) A.rates[q], pS2[q], pD2[q], A.jcExact != 0);
            }
        } else {
            vft_exp_eigen_rates<REAL, NC>(A, len1, minRel, ee1);
            vft_exp_eigen_rates<REAL, NC>(A, len2, minRel, ee2);
        }
        __syncthreads();
        if (!live) continue;
        Col<REAL, NC> c1, c2;
        const uint8_t d = direct[k];
        if (d & 1) vft_load_row<REAL, NC>(A, aN[k], p, c1);
        else vft_load_col_ml<REAL, NC>(A, aN[k], p, c1);
        if (d & 2) vft_load_row<REAL, NC>(A, bN[k], p, c2);
        else vft_load_col_ml<REAL, NC>(A, bN[k], p, c2);
        REAL wo, f[NC];
        int co;
        vft_posterior_col<REAL, NC>(A, c1, c2, jc, pS1[r], pD1[r], pS2[r], pD2[r], ee1 + r * NC, ee2 + r * NC, wo, co, f);
        vft_store_col_ml<REAL, NC>(A, outN[k], p, wo, co, f);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// optimizeAllBranchLengths' inner step (NJ.tcc:5025-5060) for one split, entirely on the device: the three branches
// around an internal node are optimised in turn, twice; for branch i the other two profiles are condensed into their
// posterior (posteriorProfile, NJ.tcc:2137) and the branch length maximises pairLogLk(P_i, posterior, x)
// (MLPairOptimize NJ.tcc:1790-1803 = onedimenmin + Brent, NJ.tcc:7025-7178).  One workgroup per split; every thread owns
// CPT columns whose two profile columns stay in registers for the whole line search, so one likelihood evaluation is
// a table rebuild (pSame/pDiff or exp(eigenvalue * rate * x) per rate category, in LDS), a few dozen VALU per column
// and one workgroup reduction - no HBM traffic and no host round trip per evaluation.  Branch lengths live in a device
// array (numeric_t, like the reference's branchlength[]), so consecutive splits of a traversal are just consecutive
// launches on the stream.  Afterwards the node's own posterior from its two children and their new lengths
// (recomputeProfile, NJ.tcc:3436-3473, useML) is written to its dense ML row.
// ------------------------------------------------------------------------------------------------------------------
// 20-state columns spread over FOUR lanes (the line-search kernels for proteins).  The reference's SSE / AVX kernels keep four
// strided accumulators over the 20 states - s[l] += x[l + 4 i] - and finish with (s0 + s1) + (s2 + s3) (vft_red4_*): lane l of
// a quad owns the states l, l + 4, ..., l + 16 of its column, i.e. accumulator l, and two quad exchanges finish the sum in the
// reference's order on all four lanes.  A thread that owns whole columns needs 20 values per profile and column - 240 VGPRs
// for the six profiles of a quartet step in double precision: round 2's kernels spilled 300 - 5 365 registers to scratch.
// A quad holds 5.
template <int CTRL>
__device__ __forceinline__ float vft_quad_perm(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ double vft_quad_perm(double x) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int) (b & 0xFFFFFFFFll), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int) (b >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double(((long long) hi << 32) | (long long) (unsigned int) lo);
}
// accumulator l on lane l -> (s0 + s1) + (s2 + s3) on all four lanes (additions commute: every lane forms the same two sums)
template <typename REAL>
__device__ __forceinline__ REAL vft_quad_red4(REAL s) {
    const REAL a = s + vft_quad_perm<0xB1>(s);   // lanes 1 0 3 2
    return a + vft_quad_perm<0x4E>(a);           // lanes 2 3 0 1
}

#define VFT_QS 5   /* states per lane */

// the model's tables in LDS for the quad kernels (global loads of them were a memory round trip per use)
template <typename REAL>
struct MlQuadTables {
    REAL codeFreq[21 * 20], eigenInv[20 * 20], statInv[20], eigenval[20], rates[VFT_MAXRATES];
};
template <typename REAL>
__device__ __forceinline__ void vft_quad_tables_load(const Arena<REAL> &A, MlQuadTables<REAL> *Q) {
    for (int t = threadIdx.x; t < 21 * 20; t += blockDim.x) Q->codeFreq[t] = A.tmCodeFreq[t];
    for (int t = threadIdx.x; t < 20 * 20; t += blockDim.x) Q->eigenInv[t] = A.tmEigenInv[t];
    for (int t = threadIdx.x; t < 20; t += blockDim.x) {
        Q->statInv[t] = A.tmStatInv[t];
        Q->eigenval[t] = A.tmEigenval[t];
    }
    for (int t = threadIdx.x; t < A.nRates && t < VFT_MAXRATES; t += blockDim.x) Q->rates[t] = A.rates[t];
}

// expEigenRates from the LDS copies (vft_exp_eigen_rates)
template <typename REAL>
__device__ __forceinline__ void vft_quad_exp_eigen_rates(const MlQuadTables<REAL> *Q, int nRates, double length, double minRel, REAL *out) {
    for (int t = threadIdx.x; t < nRates * 20; t += blockDim.x) {
        const int r = t / 20, j = t % 20;
        double relLen = length * (double) Q->rates[r];
        if (relLen < minRel) relLen = minRel;
        const REAL rl = (REAL) relLen;           // vector_multiply_by takes numeric_t
        const REAL x = Q->eigenval[j] * rl;
        out[t] = (REAL) vft_glibc_exp((double) x);
    }
}

// lane l's states of one column (vft_load_col_ml)
template <typename REAL>
__device__ __forceinline__ void vft_quad_load_col_ml(const Arena<REAL> &A, int64_t node, int64_t p, int l, Col<REAL, VFT_QS> &c) {
    if (node >= A.d.nSeqs && A.mlIs != nullptr && A.mlIs[node - A.d.nSeqs]) {
        const int64_t idx = (node - A.d.nSeqs) * A.d.nPos + p;
        c.w = A.mlW[idx];
        c.code = (int) A.mlC[idx];
        c.vec = c.w > 0 && c.code == VFT_NOCODE_;
        if (c.vec) {
            const REAL *src = A.mlF + idx * 20;
#pragma unroll
            for (int i = 0; i < VFT_QS; i++) c.f[i] = src[l + 4 * i];
        }
        return;
    }
    const int lane = (int) (node & (VFT_TILE - 1));
    const int64_t tile = node >> 6;
    if (node < A.d.nSeqs) {
        const uint4 t = A.leafT[vft_leaf_idx(A.d, tile, (int) (p >> 4), lane)];
        c.code = vft_decode<20>(vft_byte(t, (int) (p & 15)));
        c.w = c.code != VFT_NOCODE_ ? (REAL) 1 : (REAL) 0;
        c.vec = false;
        return;
    }
    const int64_t pt = tile - A.d.firstProfTile;
    const int64_t mi = vft_meta_idx(A.d, pt, p);
    const ColMask m = A.colMask[mi];
    const ColOff o = A.colOff[mi];
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint4 t = A.profC[vft_c_idx(A.d, pt, (int) (p >> 4), lane)];
    c.code = (int) vft_byte(t, (int) (p & 15));
    const bool hv = (m.vec >> lane) & 1ull;
    if ((m.w >> lane) & 1ull) c.w = A.profW[vft_wstream_base(A.d, pt) + o.w + __popcll(m.w & below)];
    else c.w = vft_implicit_weight<REAL>(c.code, hv);
    c.vec = c.w > 0 && c.code == VFT_NOCODE_;
    if (c.vec) {
        const REAL *src = A.profF + vft_fstream_base(A.d, pt);
        const int nvec = __popcll(m.vec), rank = __popcll(m.vec & below);
#pragma unroll
        for (int i = 0; i < VFT_QS; i++) c.f[i] = src[vft_fidx<REAL, 20>(o.vec, nvec, rank, l + 4 * i)];
    }
}

// lane l's part of vft_store_col_ml
template <typename REAL>
__device__ __forceinline__ void vft_quad_store_col_ml(const Arena<REAL> &A, int64_t node, int64_t p, int l, REAL w, int code, const REAL *f) {
    const int64_t idx = (node - A.d.nSeqs) * A.d.nPos + p;
    if (l == 0) {
        A.mlW[idx] = w;
        A.mlC[idx] = (uint8_t) code;
    }
    if (w > 0 && code == VFT_NOCODE_) {
        REAL *dst = A.mlF + idx * 20;
#pragma unroll
        for (int i = 0; i < VFT_QS; i++) dst[l + 4 * i] = f[i];
    }
}

// lane l's states of vft_model_freq
template <typename REAL>
__device__ __forceinline__ void vft_quad_model_freq(const MlQuadTables<REAL> *Q, const Col<REAL, VFT_QS> &c, bool mixAlways, int l, REAL *f) {
    const double w = (double) c.w;
    if (c.vec) {
#pragma unroll
        for (int i = 0; i < VFT_QS; i++) f[i] = c.f[i];
        if (!mixAlways) return;   // posteriorProfile mixes only code columns (NJ.tcc:2283-2292)
    } else {
        const int row = c.code == VFT_NOCODE_ ? 20 : c.code;
#pragma unroll
        for (int i = 0; i < VFT_QS; i++) f[i] = Q->codeFreq[row * 20 + l + 4 * i];
    }
    if (w > 0.0 && w < 1.0) {
#pragma unroll
        for (int i = 0; i < VFT_QS; i++) f[i] = (REAL) (w * (double) f[i] + (1.0 - w) * (double) Q->codeFreq[20 * 20 + l + 4 * i]);
    }
}

// vft_pair_lk_col under a matrix model (NJ.tcc:1267-1439), result on all four lanes; ee: the column's rate category's row
template <typename REAL>
__device__ __forceinline__ bool vft_quad_pair_lk_col(const MlQuadTables<REAL> *Q, const Col<REAL, VFT_QS> &c1, const Col<REAL, VFT_QS> &c2,
                                                     const REAL *ee, int l, double &lkAB) {
    lkAB = 0;
    if (c1.w == 0 && c2.w == 0 && c1.code == VFT_NOCODE_ && c2.code == VFT_NOCODE_) return false;
    REAL fA[VFT_QS], fB[VFT_QS];
    vft_quad_model_freq<REAL>(Q, c1, true, l, fA);
    vft_quad_model_freq<REAL>(Q, c2, true, l, fB);
    REAL s = 0;
#pragma unroll
    for (int i = 0; i < VFT_QS; i++) {
        REAL pr = ee[l + 4 * i] * fA[i];   // vft_red4_mul3(e, fA, fB), NJ.tcc:1359
        pr = pr * fB[i];
        s = pr + s;
    }
    lkAB = (double) vft_quad_red4<REAL>(s);
    return true;
}

// vft_posterior_col under a matrix model with 20 states (NJ.tcc:2267-2447), lane l's states of the result
template <typename REAL>
__device__ __forceinline__ void vft_quad_posterior_col(const MlQuadTables<REAL> *Q, const Col<REAL, VFT_QS> &c1, const Col<REAL, VFT_QS> &c2,
                                                       const REAL *e1, const REAL *e2, int l, Col<REAL, VFT_QS> &o) {
    o.w = (REAL) 1.0;
    o.code = VFT_NOCODE_;
#pragma unroll
    for (int i = 0; i < VFT_QS; i++) o.f[i] = 0;
    if (c1.code == VFT_NOCODE_ && c2.code == VFT_NOCODE_ && c1.w == 0 && c2.w == 0) {
        o.w = 0;   // gap with gap (NJ.tcc:2267-2272)
        o.vec = false;
        return;
    }
    REAL fM1[VFT_QS], fM2[VFT_QS], fPost[VFT_QS];
    vft_quad_model_freq<REAL>(Q, c1, false, l, fM1);
    vft_quad_model_freq<REAL>(Q, c2, false, l, fM2);
#pragma unroll
    for (int i = 0; i < VFT_QS; i++) {
        fM1[i] = fM1[i] * e1[l + 4 * i];
        fM2[i] = fM2[i] * e2[l + 4 * i];
        fPost[i] = 0;
    }
    // Every lane of the quad fetches all 20 entries of both vectors (quad broadcasts) and forms the dot products of ITS five
    // states j = l, l + 4, ... whole - four strided accumulators and (s0 + s1) + (s2 + s3), vft_red4_mul as the reference has
    // it - instead of every lane adding a quarter of all 20 dot products and two exchanges per product finishing them: no
    // cross-lane step inside the sums, eight independent chains per state.  (The loops over a lane's states stay loops: unrolled,
    // one posterior was 12 KB of straight-line code per column and per place it is used.)
    REAL g1[20], g2[20];
#pragma unroll
    for (int i = 0; i < VFT_QS; i++) {
        g1[4 * i] = vft_quad_perm<0x00>(fM1[i]);
        g1[4 * i + 1] = vft_quad_perm<0x55>(fM1[i]);
        g1[4 * i + 2] = vft_quad_perm<0xAA>(fM1[i]);
        g1[4 * i + 3] = vft_quad_perm<0xFF>(fM1[i]);
        g2[4 * i] = vft_quad_perm<0x00>(fM2[i]);
        g2[4 * i + 1] = vft_quad_perm<0x55>(fM2[i]);
        g2[4 * i + 2] = vft_quad_perm<0xAA>(fM2[i]);
        g2[4 * i + 3] = vft_quad_perm<0xFF>(fM2[i]);
    }
#pragma unroll 1
    for (int jq = 0; jq < VFT_QS; jq++) {
        const int j = l + 4 * jq;
        const REAL *cf = Q->codeFreq + j * 20;
        REAL s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 20; i += 4)
#pragma unroll
            for (int a = 0; a < 4; a++) {   // vft_red4_mul(fM1, cf), vft_red4_mul(fM2, cf)
                const REAL c = cf[i + a];
                const REAL p1 = g1[i + a] * c, p2 = g2[i + a] * c;
                s1[a] = p1 + s1[a];
                s2[a] = p2 + s2[a];
            }
        const REAL lo1 = s1[0] + s1[1], hi1 = s1[2] + s1[3], lo2 = s2[0] + s2[1], hi2 = s2[2] + s2[3];
        const REAL d1 = lo1 + hi1, d2 = lo2 + hi2;
        REAL value = d1 * d2;
        value = value * Q->statInv[j];
        value = value >= 0 ? value : (REAL) 0;
#pragma unroll
        for (int q = 0; q < VFT_QS; q++) fPost[q] = q == jq ? value : fPost[q];
    }
    REAL st = 0;
#pragma unroll
    for (int i = 0; i < VFT_QS; i++) st = fPost[i] + st;   // vft_red4_sum: lane l is accumulator l
    const double tot = (double) vft_quad_red4<REAL>(st);
    const REAL invr = (REAL) (1.0 / tot);
#pragma unroll
    for (int i = 0; i < VFT_QS; i++) fPost[i] = fPost[i] * invr;
#pragma unroll
    for (int i = 0; i < VFT_QS; i++) {
        g1[4 * i] = vft_quad_perm<0x00>(fPost[i]);
        g1[4 * i + 1] = vft_quad_perm<0x55>(fPost[i]);
        g1[4 * i + 2] = vft_quad_perm<0xAA>(fPost[i]);
        g1[4 * i + 3] = vft_quad_perm<0xFF>(fPost[i]);
    }
#pragma unroll 1
    for (int jq = 0; jq < VFT_QS; jq++) {
        const int j = l + 4 * jq;
        const REAL *ei = Q->eigenInv + j * 20;
        REAL sv[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 20; i += 4)
#pragma unroll
            for (int a = 0; a < 4; a++) {   // vft_red4_mul(fPost, eigeninv row j)
                const REAL pr = g1[i + a] * ei[i + a];
                sv[a] = pr + sv[a];
            }
        const REAL lo = sv[0] + sv[1], hi = sv[2] + sv[3];
        const REAL v = lo + hi;
#pragma unroll
        for (int q = 0; q < VFT_QS; q++) o.f[q] = q == jq ? v : o.f[q];
    }
    o.vec = true;   // code == NOCODE, w == 1
}

// element idx of a small array that must stay in registers (a runtime index would send it to scratch memory)
template <typename T, int N>
__device__ __forceinline__ T vft_sel_get(const T (&a)[N], int idx) {
    T v = a[0];
#pragma unroll
    for (int i = 1; i < N; i++) v = i == idx ? a[i] : v;
    return v;
}
template <typename T, int N>
__device__ __forceinline__ void vft_sel_set(T (&a)[N], int idx, T v) {
#pragma unroll
    for (int i = 0; i < N; i++) a[i] = i == idx ? v : a[i];
}

// ---- The batch operators under a 20-state matrix model with a quad of lanes per column (round 6).  k_pair_loglk / k_posterior give a
// lane whole columns: 160-byte vectors per lane behind loads that each touch 64 cache lines, every load inside the branches of the
// column's case analysis and therefore waited for one by one (rocprofv3 PMC, LG at 4 166 pairs x 300 columns: 85 vector loads per
// wavefront, 72 % of the wave cycles waiting, VALU busy 0.07), and k_pair_loglk's ordered total as the plain chain on one lane.  Here:
// the line-search kernels' quad layout (lane l of a quad owns the states l, l + 4, ... of its column - the reference's four strided
// accumulators, vft_quad_*), 64 columns per pass of a 256-thread workgroup, a quad's loads side by side in 32-byte pieces, the model's
// tables in LDS, and the ordered total from the workgroup-wide scans (vft_lk_total_staged: the reference's sequence of roundings
// at ~7 cycles per column).  Same operations on the same operands in the same order as the whole-column kernels: same bits.
//   k_pair_loglk_quad<REAL, CPT>: one workgroup per pair, up to CPT * 64 columns (the host takes the whole-column kernel beyond).
//   k_posterior_quad<REAL>: one workgroup per node, any length; dense ML rows only (the caller's context is in row mode).
template <typename REAL, int CPT>
__global__ __launch_bounds__(256) void k_pair_loglk_quad(Arena<REAL> A, const int64_t *aN, const int64_t *bN, const double *length, int64_t n,
                                                         double minRel, double *loglkOut, double *siteLk) {
    constexpr int WG = 256, CW = WG / 4, COLS = CPT * CW;
    typedef Col<REAL, VFT_QS> ColT;
    __shared__ REAL ee[VFT_MAXRATES * 20];
    __shared__ double stage[COLS];
    __shared__ double stageLog[COLS];
    __shared__ double stageList[COLS * 3 / 2 + 64];
    __shared__ signed char stageEvents[COLS / 2 + 16];
    __shared__ LkOrderedShared ordSh;
    __shared__ MlQuadTables<REAL> quadTab;
    const int64_t k = blockIdx.x;
    const int64_t a = aN[k], b = bN[k], nPos = A.d.nPos;
    const double len = length[k];
    const int ql = (int) (threadIdx.x & 3), qc = (int) threadIdx.x >> 2;
    vft_quad_tables_load<REAL>(A, &quadTab);
    __syncthreads();
    vft_quad_exp_eigen_rates<REAL>(&quadTab, A.nRates, len, minRel, ee);
    __syncthreads();
    double col[CPT];
#pragma unroll
    for (int c = 0; c < CPT; c++) {
        const int64_t p = (int64_t) qc + (int64_t) c * CW;
        col[c] = VFT_LK_SKIP;
        if (p < nPos) {
            ColT c1, c2;
            vft_quad_load_col_ml<REAL>(A, a, p, ql, c1);
            vft_quad_load_col_ml<REAL>(A, b, p, ql, c2);
            const int r = A.ratecat[p];
            double lkAB;
            const bool has = vft_quad_pair_lk_col<REAL>(&quadTab, c1, c2, ee + r * 20, ql, lkAB);
            if (has) col[c] = lkAB;
            if (siteLk && ql == 0) siteLk[k * nPos + p] = has ? lkAB : 1.0;
        }
    }
    vft_lk_stage_cols<CPT, 4>(stage, stageLog, col, qc, ql, CW, (int) nPos);
    const double tot = vft_lk_total_staged<WG, COLS>(stage, stageLog, stageList, stageEvents, &ordSh, (int) nPos, false);
    if (threadIdx.x == 0) loglkOut[k] = tot;
}

template <typename REAL>
__global__ __launch_bounds__(256) void k_posterior_quad(Arena<REAL> A, const int64_t *outN, const int64_t *aN, const int64_t *bN,
                                                        const double *len1A, const double *len2A, double minLen, double minRel) {
    constexpr int CW = 64;
    typedef Col<REAL, VFT_QS> ColT;
    __shared__ REAL ee1[VFT_MAXRATES * 20], ee2[VFT_MAXRATES * 20];
    __shared__ MlQuadTables<REAL> quadTab;
    const int64_t k = blockIdx.x;
    const int64_t out = outN[k], a = aN[k], b = bN[k], nPos = A.d.nPos;
    double len1 = len1A[k], len2 = len2A[k];
    if (len1 < minLen) len1 = minLen;   // NJ.tcc:2150-2155
    if (len2 < minLen) len2 = minLen;
    const int ql = (int) (threadIdx.x & 3), qc = (int) threadIdx.x >> 2;
    vft_quad_tables_load<REAL>(A, &quadTab);
    __syncthreads();
    vft_quad_exp_eigen_rates<REAL>(&quadTab, A.nRates, len1, minRel, ee1);
    vft_quad_exp_eigen_rates<REAL>(&quadTab, A.nRates, len2, minRel, ee2);
    __syncthreads();
#pragma unroll 1
    for (int64_t p = qc; p < nPos; p += CW) {
        ColT c1, c2, o;
        vft_quad_load_col_ml<REAL>(A, a, p, ql, c1);
        vft_quad_load_col_ml<REAL>(A, b, p, ql, c2);
        const int r = A.ratecat[p];
        vft_quad_posterior_col<REAL>(&quadTab, c1, c2, ee1 + r * 20, ee2 + r * 20, ql, o);
        vft_quad_store_col_ml<REAL>(A, out, p, ql, o.w, o.code, o.f);
    }
    if (threadIdx.x == 0) A.mlIs[out - A.d.nSeqs] = 1;
}

#define VFT_MLOPT_WG 256
// Threads per workgroup of the two line-search kernels.  20-state alphabets: 512, so that alignments up to 512 columns
// run with ONE column per thread - a thread keeps its columns of the three / four profiles in registers, 160 VGPRs per
// column for proteins in double, and the four-columns-per-thread instance spills 2.8 KB per lane to scratch
// (k_ml_quartet<double, 20, 4>: 1.2 ms per quartet round at 300 columns).  The column -> thread mapping is free to
// change there: matrix models take their totals in column order (vft_lk_total_ordered).  Nucleotides keep 256: the
// Jukes-Cantor totals are per-thread products summed in thread order, and that order is pinned by the fixtures.
template <int NC> struct MlOptWG { static const int value = NC == 20 ? 512 : VFT_MLOPT_WG; };
// ... and with a quad of lanes per column (proteins under a matrix model, up to 512 columns): 256 threads = 64 columns per pass.
// One wavefront per SIMD may use all 512 VGPRs: the kernels' scalar state (Brent's variables, three pairings' lengths) takes ~250
// of them, five columns of three profiles ~210; at 512 threads (256 VGPRs per lane) the columns went to scratch, 6 us per evaluation.
template <int NC, bool QUAD> struct MlLineWG { static const int value = QUAD ? 256 : MlOptWG<NC>::value; };
#define VFT_MLOPT_MAXLEN 6.0

// Brent's minimiser with the reference's bracketing (onedimenmin): all threads run it in lockstep on uniform values;
// `eval` is a workgroup-collective call.  Written as ONE loop around ONE call of eval - the bracket's three points, the two
// widening loops and Brent's iterations are states of it: `eval` is the whole likelihood evaluation, inlined, and six copies of
// it per search (times the searches of a kernel) made kernels of 0.2 - 1 MB whose every evaluation ran from a cold instruction
// cache (round 2: everything in these kernels took 3 - 4 times what its instructions cost).  Same evaluations in the same
// order, same arithmetic.
template <typename EVAL>
__device__ __forceinline__ double vft_min_branch_length(EVAL &&eval, double xmin, double xguess, double xmax, double ftol,
                                                        double atol, double &fBest) {
    // initial bracket lo < mid < hi around the guess
    double lo, mid, hi;
    if (xguess == xmin) {
        lo = xmin;
        mid = 2.0 * xguess;
        hi = 10.0 * xguess;
    } else if (xguess <= 2.0 * xmin) {
        lo = xmin;
        mid = xguess;
        hi = 5.0 * xguess;
    } else {
        lo = 0.5 * xguess;
        mid = xguess;
        hi = 2.0 * xguess;
    }
    if (hi > xmax) hi = xmax;
    if (mid >= hi) mid = 0.5 * (lo + hi);
    const double golden = 0.3819660, zeps = 1.0e-10;
    double fLo = 0, fMid = 0, fHi = 0;
    // Brent: x = best point, w = second best, v = previous w; [a, b] brackets the minimum
    double a = 0, b = 0, x = mid, fx = 0, w = 0, fw = 0, v = 0, fv = 0, step = 0.0, prevStep = 0.0, u = 0;
    // state: 0, 1, 2 = f(lo), f(mid), f(hi) of the bracket; 3 / 4 = a step of the first / second widening loop; 5 = Brent
    int state = 0, it = 0;
    double xe = lo;
#pragma unroll 1
    for (;;) {
        const double f = eval(xe);
        if (state == 0) {
            fLo = f;
            xe = mid;
            state = 1;
            continue;
        }
        if (state == 1) {
            fMid = f;
            xe = hi;
            state = 2;
            continue;
        }
        if (state == 2 || state == 4) fHi = f;
        else if (state == 3) fLo = f;
        if (state <= 3 && fLo < fMid && lo > xmin) {   // widen towards the lower limit while the minimum is not inside
            lo = (lo + xmin) / 2.0;
            if (lo < 2.0 * xmin) lo = xmin;
            xe = lo;
            state = 3;
            continue;
        }
        if (state <= 4) {
            if (fHi < fMid && hi < xmax) {   // ... and towards the upper one
                hi = (hi + xmax) / 2.0;
                if (hi > xmax * 0.95) hi = xmax;
                xe = hi;
                state = 4;
                continue;
            }
            a = lo < hi ? lo : hi;
            b = lo > hi ? lo : hi;
            x = mid;
            fx = fMid;
            if (fLo < fHi) {
                w = lo; fw = fLo; v = hi; fv = fHi;
            } else {
                w = hi; fw = fHi; v = lo; fv = fLo;
            }
            state = 5;
        } else {   // f = f(u) of a Brent step
            const double fu = f;
            if (fu <= fx) {
                if (u >= x) a = x; else b = x;
                v = w; w = x; x = u;
                fv = fw; fw = fx; fx = fu;
            } else {
                if (u < x) a = u; else b = u;
                if (fu <= fw || w == x) {
                    v = w; w = u;
                    fv = fw; fw = fu;
                } else if (fu <= fv || v == x || v == w) {
                    v = u;
                    fv = fu;
                }
            }
        }
        // the next Brent step, or the end
        if (it == 100) break;
        const double xm = 0.5 * (a + b);
        const double tol1 = ftol * fabs(x), tol2 = 2.0 * (tol1 + zeps);
        if (fabs(x - xm) <= (tol2 - 0.5 * (b - a)) || fabs(a - b) < atol) break;
        bool goldenStep = true;
        if (fabs(prevStep) > tol1) {
            // parabola through x, w, v
            const double r = (x - w) * (fx - fv);
            double q = (x - v) * (fx - fw);
            double p = (x - v) * q - (x - w) * r;
            q = 2.0 * (q - r);
            if (q > 0.0) p = -p;
            q = fabs(q);
            const double before = prevStep;
            prevStep = step;
            if (!(fabs(p) >= fabs(0.5 * q * before) || p <= q * (a - x) || p >= q * (b - x))) {
                step = p / q;
                const double uu = x + step;
                if (uu - a < tol2 || b - uu < tol2) step = (xm - x) >= 0.0 ? fabs(tol1) : -fabs(tol1);
                goldenStep = false;
            }
        }
        if (goldenStep) {
            prevStep = x >= xm ? a - x : b - x;
            step = golden * prevStep;
        }
        u = fabs(step) >= tol1 ? x + step : x + (step >= 0.0 ? fabs(tol1) : -fabs(tol1));
        xe = u;
        it++;
    }
    fBest = fx;
    return x;
}

template <typename REAL, int NC, int CPT, bool QUAD>
__global__ __launch_bounds__((MlLineWG<NC, QUAD>::value)) void k_ml_node_lengths(Arena<REAL> A, const int64_t *ids, const int64_t *lenIdx,
                                                                  const int64_t *recN, REAL *blen, double minLen,
                                                                  double minRel, double ftol, double atol,
                                                                  unsigned int *evalCount) {
    static_assert(!QUAD || NC == 20, "quads of lanes hold 20-state columns");
    constexpr int WG = MlLineWG<NC, QUAD>::value;
    constexpr int LPC = QUAD ? 4 : 1, NS = NC / LPC, CW = WG / LPC, COLS = CPT * CW;   // lanes per column, states per lane, columns per pass
    typedef Col<REAL, NS> ColT;
    __shared__ REAL ee1[VFT_MAXRATES * NC], ee2[VFT_MAXRATES * NC];
    __shared__ double pS1[VFT_MAXRATES], pD1[VFT_MAXRATES], pS2[VFT_MAXRATES], pD2[VFT_MAXRATES];
    __shared__ double red[WG / 64];
    __shared__ double stage[COLS];   // ordered total of matrix models (vft_lk_total_staged)
    __shared__ double stageLog[COLS];
    constexpr bool V1 = NC == 20 && !QUAD;   // (vft_lk_total_staged_v1)
    __shared__ double stageList[COLS * 3 / 2 + (V1 ? 32 : 64)];
    __shared__ signed char stageEvents[COLS / 2 + (V1 ? 0 : 16)];
    __shared__ LkOrderedShared ordSh;
    __shared__ typename std::conditional<QUAD, MlQuadTables<REAL>, int>::type quadTab;
    const int64_t k = blockIdx.x;
    const bool jc = A.tmStat == nullptr;
    const int64_t nPos = A.d.nPos;
    const int ql = QUAD ? (int) (threadIdx.x & 3) : 0, qc = (int) threadIdx.x / LPC;
    if constexpr (QUAD) vft_quad_tables_load<REAL>(A, &quadTab);
    int rc[CPT];
#pragma unroll
    for (int c = 0; c < CPT; c++) {
        const int64_t p = (int64_t) qc + (int64_t) c * CW;
        rc[c] = p < nPos ? A.ratecat[p] : 0;
    }
    auto eigenTable = [&](double len, REAL *out) {
        if constexpr (QUAD) vft_quad_exp_eigen_rates<REAL>(&quadTab, A.nRates, len, minRel, out);
        else vft_exp_eigen_rates<REAL, NC>(A, len, minRel, out);
    };
    // tables of the two branch lengths of a posterior, into LDS (the caller synchronises)
    auto tables2 = [&](double l1, double l2) {
        if (l1 < minLen) l1 = minLen;   // NJ.tcc:2150-2155
        if (l2 < minLen) l2 = minLen;
        if (jc) {
            for (int r = threadIdx.x; r < A.nRates; r += WG) {
                vft_psame_pdiff(l1, (double) A.rates[r], pS1[r], pD1[r], A.jcExact != 0);
                vft_psame_pdiff(l2, (double) A.rates[r], pS2[r], pD2[r], A.jcExact != 0);
            }
        } else {
            eigenTable(l1, ee1);
            eigenTable(l2, ee2);
        }
    };
    auto loadCol = [&](int64_t node, int64_t p, ColT &c) {
        if constexpr (QUAD) vft_quad_load_col_ml<REAL>(A, node, p, ql, c);
        else vft_load_col_ml<REAL, NC>(A, node, p, c);
    };
    // posteriorProfile of one column from the tables in ee1 / ee2 (pS1.. under Jukes-Cantor)
    auto post = [&](const ColT &c1, const ColT &c2, int r, ColT &o) __attribute__((always_inline)) {
        if constexpr (QUAD) {
            vft_quad_posterior_col<REAL>(&quadTab, c1, c2, ee1 + r * NC, ee2 + r * NC, ql, o);
        } else {
            vft_posterior_col<REAL, NC>(A, c1, c2, jc, pS1[r], pD1[r], pS2[r], pD2[r], ee1 + r * NC, ee2 + r * NC, o.w, o.code, o.f);
            o.vec = o.code == VFT_NOCODE_ && o.w > (REAL) 0;
        }
    };
    unsigned int nEval = 0;
    for (int round = 0; round < 6; round++) {   // 2 iterations x 3 branches (NJ.tcc:5038-5059)
        const int i = round % 3, b1 = (i + 1) % 3, b2 = (i + 2) % 3;
        const int64_t nI = ids[3 * k + i], n1 = ids[3 * k + b1], n2 = ids[3 * k + b2];
        const int64_t lI = lenIdx[3 * k + i];
        __syncthreads();   // the previous round's writes to blen[] and reads of the tables are done
        tables2((double) blen[lenIdx[3 * k + b1]], (double) blen[lenIdx[3 * k + b2]]);
        __syncthreads();
        ColT pA[CPT], pB[CPT];
#pragma unroll
        for (int c = 0; c < CPT; c++) {
            const int64_t p = (int64_t) qc + (int64_t) c * CW;
            if (p < nPos) {
                ColT c1, c2;
                loadCol(n1, p, c1);
                loadCol(n2, p, c2);
                post(c1, c2, rc[c], pB[c]);
                loadCol(nI, p, pA[c]);
            }
        }
        // -pairLogLk(P_i, posterior, x), collectively.  No barrier in front: the previous evaluation ended with one after
        // every wave had read the tables, and the one below separates the first evaluation from the posterior above.
        __syncthreads();
        auto negLogLk = [&](double x) -> double {
            if (jc) {
                for (int r = threadIdx.x; r < A.nRates; r += WG) vft_psame_pdiff(x, (double) A.rates[r], pS1[r], pD1[r], A.jcExact != 0);
            } else {
                eigenTable(x, ee1);
            }
            __syncthreads();
            if (!jc || (NC == 4 && A.jcExact)) {   // matrix model (and Jukes-Cantor to the last bit, Arena::jcExact): the reference's ordered total
                double col[CPT];
#pragma unroll
                for (int c = 0; c < CPT; c++) {
                    const int64_t p = (int64_t) qc + (int64_t) c * CW;
                    col[c] = VFT_LK_SKIP;
                    if (p < nPos) {
                        const int r = rc[c];
                        double lkAB;
                        bool ok;
                        if constexpr (QUAD) ok = vft_quad_pair_lk_col<REAL>(&quadTab, pA[c], pB[c], ee1 + r * NC, ql, lkAB);
                        else ok = vft_pair_lk_col<REAL, NC>(A, pA[c], pB[c], jc, pS1[r], pD1[r], ee1 + r * NC, lkAB);
                        if (ok) col[c] = lkAB;
                    }
                }
                vft_lk_stage_cols<CPT, LPC>(stage, stageLog, col, qc, ql, CW, (int) nPos);
                nEval++;
                return -(V1 ? vft_lk_total_staged_v1<WG, COLS>(stage, stageLog, stageList, stageEvents, &ordSh, (int) nPos, jc) : vft_lk_total_staged<WG, COLS>(stage, stageLog, stageList, stageEvents, &ordSh, (int) nPos, jc));
            }
            double tot = 0;
            if constexpr (!QUAD) {
                double lk = 1.0, loglk = 0.0;
#pragma unroll
                for (int c = 0; c < CPT; c++) {
                    const int64_t p = (int64_t) threadIdx.x + (int64_t) c * WG;
                    if (p < nPos) {
                        const int r = rc[c];
                        double lkAB;
                        if (vft_pair_lk_col<REAL, NC>(A, pA[c], pB[c], jc, pS1[r], pD1[r], ee1 + r * NC, lkAB))
                            vft_lk_accumulate(lkAB, jc, lk, loglk);
                    }
                }
                double part = loglk + log(lk);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
                if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
                __syncthreads();
#pragma unroll
                for (int w = 0; w < WG / 64; w++) tot += red[w];
            }
            nEval++;
            return -tot;
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
#pragma unroll
    for (int c = 0; c < CPT; c++) {
        const int64_t p = (int64_t) qc + (int64_t) c * CW;
        if (p < nPos) {
            ColT c1, c2, o;
            loadCol(ids[3 * k], p, c1);
            loadCol(ids[3 * k + 1], p, c2);
            post(c1, c2, rc[c], o);
            if constexpr (QUAD) vft_quad_store_col_ml<REAL>(A, rec, p, ql, o.w, o.code, o.f);
            else vft_store_col_ml<REAL, NC>(A, rec, p, o.w, o.code, o.f);
        }
    }
    if (threadIdx.x == 0) A.mlIs[rec - A.d.nSeqs] = 1;
}


// ------------------------------------------------------------------------------------------------------------------
// Quartet likelihood kernel, one workgroup per internal split (nodes A, B below it, C beside it, D = the rest of the
// tree through the up-profile; setupABCD NJ.tcc:1942-1975).  Building block: MLQuartetOptimize (NJ.tcc:1650-1788) of one
// of the three pairings - internal branch, then A, B, C, D, each a Brent search (vft_min_branch_length) against the
// posterior of the other three profiles - with the same in-register line search as k_ml_node_lengths; profiles are
// reloaded from the arena (L2) at every step, only the two profiles of the current search live in registers.
//   mode 0 - testSplitsML's per-split work (NJ.tcc:6856-6925): MLQuartetLogLk (NJ.tcc:5412-5427) of AB|CD with its
//     current lengths, the two alternatives optimised, a second pass for the better one when it comes within
//     closeLogLkLimit of AB|CD (or always, -mlacc 2), and the per-site log-likelihoods of the three topologies for the
//     SH-like resampling (k_sh_support).  All splits of the (now fixed) tree in one launch.
//   mode 1 - MLQuartetNNI (NJ.tcc:4885-5004), the evaluation of one ML NNI: up to two rounds of {AB|CD with the star
//     topology test, AC|BD, AD|BC}, dropping alternatives that fall clearly behind; the winner's five branch lengths
//     are written to branchlength[] on the device the way DoNNI does it (NJ.tcc:5889-5915: every node keeps its own
//     optimised length; after a star test only the internal branch), choice and criteria go to the host.
struct QuartetNNIResult {
    double criteria[3];
    int32_t choice, star;
};

// MLQuartetNNI spread over the chip (mode 2): the three pairings of a round do not depend on each other, so a round is
// ONE launch with a workgroup per pairing; this record carries the quartet between the launches of
// init -> round -> decide -> round -> verdict.  The star test belongs to AB|CD; when it fires the other two workgroups'
// results are simply not used - what the one-thread reference would not have computed at all.
struct QuartetNNIState {
    double len[3][5];
    double crit[3];
    int32_t consider1, consider2, done, star;
};

template <typename REAL, int NC, int CPT, bool QUAD>
__global__ __launch_bounds__((MlLineWG<NC, QUAD>::value)) void k_ml_quartet(Arena<REAL> A, const int64_t *ids, const int64_t *lenIdx, REAL *blen,
                                                             double minLen, double minRel, double ftol, double atol,
                                                             double closeLimit, int mlAccuracy, int mode, double *loglkOut,
                                                             double *siteOut, double *lenOut, QuartetNNIResult *nniOut,
                                                             QuartetNNIState *nniState, unsigned int *evalCount) {
    static_assert(!QUAD || NC == 20, "quads of lanes hold 20-state columns");
    constexpr int WG = MlLineWG<NC, QUAD>::value;
    constexpr int LPC = QUAD ? 4 : 1, NS = NC / LPC, CW = WG / LPC, COLS = CPT * CW;   // lanes per column, states per lane, columns per pass
    typedef Col<REAL, NS> ColT;
    __shared__ REAL ee[4][VFT_MAXRATES * NC];
    __shared__ double pS[4][VFT_MAXRATES], pD[4][VFT_MAXRATES];
    __shared__ double red[WG / 64];
    __shared__ double stage[COLS];   // ordered total of matrix models (vft_lk_total_staged)
    __shared__ double stageLog[COLS];
    constexpr bool V1 = NC == 20 && !QUAD;   // (vft_lk_total_staged_v1)
    __shared__ double stageList[COLS * 3 / 2 + (V1 ? 32 : 64)];
    __shared__ signed char stageEvents[COLS / 2 + (V1 ? 0 : 16)];
    __shared__ LkOrderedShared ordSh;
    __shared__ typename std::conditional<QUAD, MlQuadTables<REAL>, int>::type quadTab;
    const int64_t k = blockIdx.x;
    const bool jc = A.tmStat == nullptr;
    const int64_t nPos = A.d.nPos;
    const int ql = QUAD ? (int) (threadIdx.x & 3) : 0, qc = (int) threadIdx.x / LPC;
    if constexpr (QUAD) vft_quad_tables_load<REAL>(A, &quadTab);
    int rc[CPT];   // rate category of the thread's columns; -1 beyond the alignment (such columns hold no data at all)
#pragma unroll
    for (int c = 0; c < CPT; c++) {
        const int64_t p = (int64_t) qc + (int64_t) c * CW;
        rc[c] = p < nPos ? A.ratecat[p] : -1;
    }
    unsigned int nEval = 0;
#ifdef VFT_ML_TIMING
    unsigned long long mlTick_ = wall_clock64();
#endif
    // P(t) tables of one branch length into slot s (callers synchronise); posteriorProfile clamps its lengths
    auto table = [&](int s, double len, bool clamp) {
        if (clamp && len < minLen) len = minLen;
        if (jc) {
            for (int r = threadIdx.x; r < A.nRates; r += WG) vft_psame_pdiff(len, (double) A.rates[r], pS[s][r], pD[s][r], A.jcExact != 0);
        } else {
            if constexpr (QUAD) vft_quad_exp_eigen_rates<REAL>(&quadTab, A.nRates, len, minRel, ee[s]);
            else vft_exp_eigen_rates<REAL, NC>(A, len, minRel, ee[s]);
        }
    };
    auto post = [&](const ColT &c1, const ColT &c2, int s1, int s2, int r, ColT &o) __attribute__((always_inline)) {
        if (r < 0) return;
        if constexpr (QUAD) {
            vft_quad_posterior_col<REAL>(&quadTab, c1, c2, ee[s1] + r * NC, ee[s2] + r * NC, ql, o);
        } else {
            vft_posterior_col<REAL, NC>(A, c1, c2, jc, pS[s1][r], pD[s1][r], pS[s2][r], pD[s2][r], ee[s1] + r * NC, ee[s2] + r * NC,
                                        o.w, o.code, o.f);
            o.vec = o.code == VFT_NOCODE_ && o.w > (REAL) 0;
        }
    };
    // pairLogLk(X, Y, len) over the workgroup (table slot 0); site != nullptr: multiply the per-site likelihoods in
    // (lead: barrier in front - needed unless the previous thing the workgroup did was another pairTotal, whose final
    //  barrier already came after every wave's last table read)
    auto pairTotal = [&](const ColT *X, const ColT *Y, double len, double *site, bool useSite, bool lead = true) __attribute__((always_inline)) -> double {
        if (lead) __syncthreads();
        VFT_ML_TICK(0);   // whatever came before this evaluation
        table(0, len, false);
        __syncthreads();
        VFT_ML_TICK(1);   // tables
        if (!jc || (NC == 4 && A.jcExact)) {   // matrix model (and Jukes-Cantor to the last bit, Arena::jcExact): the reference's ordered total
            double col[CPT];
#pragma unroll
            for (int c = 0; c < CPT; c++) {
                const int64_t p = (int64_t) qc + (int64_t) c * CW;
                col[c] = VFT_LK_SKIP;
                if (p < nPos) {
                    const int r = rc[c];
                    double lkAB;
                    bool ok;
                    if constexpr (QUAD) ok = vft_quad_pair_lk_col<REAL>(&quadTab, X[c], Y[c], ee[0] + r * NC, ql, lkAB);
                    else ok = vft_pair_lk_col<REAL, NC>(A, X[c], Y[c], jc, pS[0][r], pD[0][r], ee[0] + r * NC, lkAB);
                    if (ok) {
                        col[c] = lkAB;
                        if (useSite) site[c] *= lkAB;
                    }
                }
            }
            vft_lk_stage_cols<CPT, LPC>(stage, stageLog, col, qc, ql, CW, (int) nPos);
            nEval++;
            VFT_ML_TICK(2);   // column likelihoods
            const double total_ = (V1 ? vft_lk_total_staged_v1<WG, COLS>(stage, stageLog, stageList, stageEvents, &ordSh, (int) nPos, jc) : vft_lk_total_staged<WG, COLS>(stage, stageLog, stageList, stageEvents, &ordSh, (int) nPos, jc));
            VFT_ML_TICK(3);   // ordered total
            return total_;
        }
        double tot = 0;
        if constexpr (!QUAD) {
            double lk = 1.0, loglk = 0.0;
#pragma unroll
            for (int c = 0; c < CPT; c++) {
                const int64_t p = (int64_t) threadIdx.x + (int64_t) c * WG;
                if (p < nPos) {
                    const int r = rc[c];
                    double lkAB;
                    if (vft_pair_lk_col<REAL, NC>(A, X[c], Y[c], jc, pS[0][r], pD[0][r], ee[0] + r * NC, lkAB)) {
                        vft_lk_accumulate(lkAB, jc, lk, loglk);
                        if (useSite) site[c] *= lkAB;
                    }
                }
            }
            double part = loglk + log(lk);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
            __syncthreads();
#pragma unroll
            for (int w = 0; w < WG / 64; w++) tot += red[w];
        }
        nEval++;
        return tot;
    };
    auto loadCols = [&](int64_t node, ColT *dst) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < CPT; c++) {
            const int64_t p = (int64_t) qc + (int64_t) c * CW;
            if (p < nPos) {
                if constexpr (QUAD) vft_quad_load_col_ml<REAL>(A, node, p, ql, dst[c]);
                else vft_load_col_ml<REAL, NC>(A, node, p, dst[c]);
            }
        }
    };
    auto storeSite = [&](int topo, const double *site) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < CPT; c++) {
            const int64_t p = (int64_t) qc + (int64_t) c * CW;
            if (p < nPos && ql == 0)   // SHSupport takes the logs, NJ.tcc:1134-1137 (glibc's log where the totals are the reference's)
                siteOut[(k * 3 + topo) * nPos + p] = ((!jc || (NC == 4 && A.jcExact)) && site[c] > 0.0) ? vft_glibc_log(site[c]) : log(site[c]);
        }
    };
    const int64_t nA = ids[4 * k], nB = ids[4 * k + 1], nC = ids[4 * k + 2], nD = ids[4 * k + 3];
    double base[5];
#pragma unroll
    for (int t = 0; t < 5; t++) base[t] = (double) blen[lenIdx[5 * k + t]];
    double crit[3] = {0, 0, 0};
    ColT X[CPT], Y[CPT], T[CPT];
    double site[CPT];

    if (mode == 0) {
        // ---- AB|CD with the lengths as they are: pairLogLk(A,B) + pairLogLk(C,D) + pairLogLk(AB,CD)
#pragma unroll
        for (int c = 0; c < CPT; c++) site[c] = 1.0;
        double tot = 0;
#pragma unroll 1
        for (int half = 0; half < 2; half++) {   // (A, B) -> T = AB, then (C, D) -> Y = CD: one piece of code for both
            loadCols(half ? nC : nA, X);
            loadCols(half ? nD : nB, Y);
            tot += pairTotal(X, Y, base[2 * half] + base[2 * half + 1], site, true);
            __syncthreads();
            table(1, base[2 * half], true);
            table(2, base[2 * half + 1], true);
            __syncthreads();
#pragma unroll
            for (int c = 0; c < CPT; c++) {
                ColT r;
                post(X[c], Y[c], 1, 2, rc[c], r);
                if (half == 0) T[c] = r;
                else Y[c] = r;
            }
        }
        tot += pairTotal(T, Y, base[4], site, true);
        crit[0] = tot;
        storeSite(0, site);
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
            loadCols(step <= 2 ? qc : qa, X);
            loadCols(step <= 2 ? qd : qb, Y);
#pragma unroll
            for (int c = 0; c < CPT; c++) post(X[c], Y[c], 1, 2, rc[c], T[c]);   // CD (steps 0-2) or AB (steps 3, 4)
            // the outer posterior, one piece of code for the five steps: (A, B) -> pair1 = AB, pair2 = CD | (B, CD) resp. (A, CD)
            // -> pair2 = BCD / ACD, pair1 = A / B | (AB, D) resp. (AB, C) -> pair2 = ABD / ABC, pair1 = C / D
            loadCols(step == 0 ? qa : step == 1 ? qb : step == 2 ? qa : step == 3 ? qd : qc, X);
            if (step == 0) loadCols(qb, Y);
#pragma unroll
            for (int c = 0; c < CPT; c++) {
                ColT r, u = X[c], v = X[c];   // (operands by value: selected field by field, the arrays stay in registers)
                if (step >= 3) u = T[c];
                if (step == 0) v = Y[c];
                else if (step <= 2) v = T[c];
                post(u, v, 3, 0, rc[c], r);
                if (step == 0) {
                    X[c] = r;
                    Y[c] = T[c];
                } else {
                    Y[c] = r;
                }
            }
            if (step != 0) loadCols(step == 1 ? qa : step == 2 ? qb : step == 3 ? qc : qd, X);
            __syncthreads();   // the posteriors above are done with the tables
            VFT_ML_TICK(4);   // a step's tables, loads and posteriors
            auto negLogLk = [&](double x) -> double { return -pairTotal(X, Y, x, site, false, false); };
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
            loadCols(qa, X);
            loadCols(qb, Y);
            double tot = -negll + pairTotal(X, Y, L[0] + L[1], site, false);
            loadCols(qc, X);
            loadCols(qd, Y);
            tot += pairTotal(X, Y, L[2] + L[3], site, false);
            crit[0] = tot;
            crit[1] = crit[2] = -1e20;
            star = true;
            VFT_STORE_L();
            break;
        }
        // total: pairLogLk(ABC, D) (= the last search's optimum) + pairLogLk(AB, C, lI + lC) + pairLogLk(A, B, lA + lB);
        // X = D, Y = ABC, T = AB here
        const bool sitep = mode == 0;
        if (sitep) {
#pragma unroll
            for (int c = 0; c < CPT; c++) site[c] = 1.0;
            pairTotal(Y, X, L[3], site, true);
        }
        double tot = -negll;
        loadCols(qc, X);
        tot += pairTotal(T, X, L[4] + L[2], site, sitep);
        loadCols(qa, X);
        loadCols(qb, Y);
        tot += pairTotal(X, Y, L[0] + L[1], site, sitep);
        vft_sel_set<double, 3>(crit, t, tot);
        if (sitep) storeSite(t, site);
        VFT_STORE_L();
    }
#undef VFT_STORE_L
    if (threadIdx.x != 0) return;
#ifdef VFT_ML_TIMING
    VFT_ML_TICK(0);
    atomicAdd(&vftMlTicks[8], (unsigned long long) nEval);
    atomicAdd(&vftMlTicks[9], 1ull);
#endif
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

// MLQuartetNNI's bookkeeping around the parallel rounds (mode 2 of k_ml_quartet), one thread per quartet
template <typename REAL>
__global__ void k_ml_nni_init(const int64_t *lenIdx, const REAL *blen, QuartetNNIState *state, int64_t n) {
    const int64_t k = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double b[5];
    for (int t = 0; t < 5; t++) b[t] = (double) blen[lenIdx[5 * k + t]];
    QuartetNNIState &st = state[k];
    const double l[3][5] = {{b[0], b[1], b[2], b[3], b[4]}, {b[0], b[2], b[1], b[3], b[4]}, {b[0], b[3], b[2], b[1], b[4]}};
    for (int t = 0; t < 3; t++) {
        st.crit[t] = 0;
        for (int j = 0; j < 5; j++) st.len[t][j] = l[t][j];
    }
    st.consider1 = st.consider2 = 1;
    st.done = st.star = 0;
}

// end of a round (NJ.tcc:4910-4916, :4961-4983)
static __global__ void k_ml_nni_decide(QuartetNNIState *state, int64_t n, double minLen, double closeLimit, int mlAccuracy, int lastRound) {
    const int64_t k = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    QuartetNNIState &st = state[k];
    if (st.done) return;
    if (st.star) {
        st.crit[1] = st.crit[2] = -1e20;
        st.done = 1;
        return;
    }
    const double *c = st.crit;
    if (mlAccuracy < 2) {
        if (c[1] < c[0] - closeLimit || (st.len[1][4] <= 2.0 * minLen && c[1] < c[0])) st.consider1 = 0;
        if (c[2] < c[0] - closeLimit || (st.len[2][4] <= 2.0 * minLen && c[2] < c[0])) st.consider2 = 0;
        if (!st.consider1 && !st.consider2) st.done = 1;
        else if (c[1] > c[0] + closeLimit && c[1] > c[2] + closeLimit) st.done = 1;
        else if (c[2] > c[0] + closeLimit && c[2] > c[1] + closeLimit) st.done = 1;
    }
    if (lastRound) st.done = 1;
}

// the verdict (NJ.tcc:4989-5003) and DoNNI's branch-length update (NJ.tcc:5889-5915)
template <typename REAL>
__global__ void k_ml_nni_verdict(const QuartetNNIState *state, const int64_t *lenIdx, REAL *blen, QuartetNNIResult *out, int64_t n) {
    const int64_t k = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const QuartetNNIState &st = state[k];
    const double *c = st.crit;
    int choice = 0;
    if (!st.star) {
        if (c[1] > c[0] && c[1] > c[2]) choice = 1;
        else if (c[2] > c[0] && c[2] > c[1]) choice = 2;
    }
    const int64_t *li = lenIdx + 5 * k;
    if (st.star) {
        blen[li[4]] = (REAL) st.len[0][4];
    } else {
        const double *L = st.len[choice];
        blen[li[0]] = (REAL) L[0];
        blen[li[1]] = (REAL) (choice == 0 ? L[1] : choice == 1 ? L[2] : L[3]);
        blen[li[2]] = (REAL) (choice == 0 ? L[2] : choice == 1 ? L[1] : L[2]);
        blen[li[3]] = (REAL) (choice == 0 ? L[3] : choice == 1 ? L[3] : L[1]);
        blen[li[4]] = (REAL) L[4];
    }
    QuartetNNIResult r;
    r.criteria[0] = c[0];
    r.criteria[1] = c[1];
    r.criteria[2] = c[2];
    r.choice = choice;
    r.star = st.star;
    out[k] = r;
}

// SHSupport (NJ.tcc:1126-1165): the share of column resamples in which the gap between the best and the second best
// of the three topologies' resampled log-likelihoods (each centred by its own total) stays below the observed gap.
// One workgroup per split; its 3 x nPos site log-likelihoods sit in LDS, a thread per resample walks the resample's
// columns in order (the reference's sequence of additions).  colT: [nPos][nBoot] (transposed, so that the threads of
// a wavefront read consecutive entries).
// inLds = 0 (alignments whose 3 x nPos doubles do not fit the LDS, > 6 800 columns): the site values are read where they are.
static __global__ __launch_bounds__(256) void k_sh_support(const double *siteLoglk, const double *loglk, const uint16_t *colT,
                                                    int64_t nPos, int32_t nBoot, double *support, int inLds) {
    extern __shared__ double sSiteLds[];   // [3][nPos]
    __shared__ unsigned int sCount;
    const int64_t k = blockIdx.x;
    if (inLds)
        for (int64_t i = threadIdx.x; i < 3 * nPos; i += blockDim.x) sSiteLds[i] = siteLoglk[k * 3 * nPos + i];
    const double *sSite = inLds ? sSiteLds : siteLoglk + k * 3 * nPos;
    if (threadIdx.x == 0) sCount = 0;
    __syncthreads();
    const double l0 = loglk[3 * k], l1 = loglk[3 * k + 1], l2 = loglk[3 * k + 2];
    const double d1 = l0 - l1, d2 = l0 - l2;
    const double delta = d1 < d2 ? d1 : d2;
    unsigned int mine = 0;
    for (int32_t b = threadIdx.x; b < nBoot; b += blockDim.x) {
        double r0 = -l0, r1 = -l1, r2 = -l2;
        for (int64_t j = 0; j < nPos; j++) {
            const int64_t pos = colT[j * nBoot + b];
            r0 += sSite[pos];
            r1 += sSite[nPos + pos];
            r2 += sSite[2 * nPos + pos];
        }
        const double r[3] = {r0, r1, r2};
        int best = 0;
        if (r[1] > r[best]) best = 1;
        if (r[2] > r[best]) best = 2;
        const double s1 = r[best] - r[(best + 1) % 3], s2 = r[best] - r[(best + 2) % 3];
        const double rd = s1 < s2 ? s1 : s2;
        if (rd < delta) mine++;
    }
    atomicAdd(&sCount, mine);
    __syncthreads();
    if (threadIdx.x == 0) support[k] = (double) sCount / (double) nBoot;
}


// ------------------------------------------------------------------------------------------------------------------
// The two line-search kernels are by far the largest pieces of device code.  They are instantiated in their own
// translation units (vft_ml_kernels_*.hip) so that the library builds as parallel hipcc jobs; vft_api.hip only
// declares the instances (`extern template`).
#define VFT_ML_NODE_LENGTHS_INSTANCE(PFX, REAL, NC, CPT, QUAD)                                                            \
    PFX template __global__ void k_ml_node_lengths<REAL, NC, CPT, QUAD>(Arena<REAL>, const int64_t *, const int64_t *, const int64_t *, \
                                                                       REAL *, double, double, double, double, unsigned int *);
#define VFT_ML_QUARTET_INSTANCE(PFX, REAL, NC, CPT, QUAD)                                                                  \
    PFX template __global__ void k_ml_quartet<REAL, NC, CPT, QUAD>(Arena<REAL>, const int64_t *, const int64_t *, REAL *, double, double, \
                                                                  double, double, double, int, int, double *, double *, double *,  \
                                                                  QuartetNNIResult *, QuartetNNIState *, unsigned int *);
// 20 states: quads of lanes per column for alignments up to 512 columns under a matrix model (two, five or eight passes of 64
// columns), whole columns per thread (four passes of 512) beyond that and under Jukes-Cantor
#define VFT_ML_NODE_LENGTHS_INSTANCES(PFX)                      \
    VFT_ML_NODE_LENGTHS_INSTANCE(PFX, float, 4, 1, false)       \
    VFT_ML_NODE_LENGTHS_INSTANCE(PFX, float, 4, 4, false)       \
    VFT_ML_NODE_LENGTHS_INSTANCE(PFX, float, 4, 8, false)       \
    VFT_ML_NODE_LENGTHS_INSTANCE(PFX, float, 20, 2, true)       \
    VFT_ML_NODE_LENGTHS_INSTANCE(PFX, float, 20, 5, true)       \
    VFT_ML_NODE_LENGTHS_INSTANCE(PFX, float, 20, 8, true)       \
    VFT_ML_NODE_LENGTHS_INSTANCE(PFX, float, 20, 4, false)      \
    VFT_ML_NODE_LENGTHS_INSTANCE(PFX, double, 4, 1, false)      \
    VFT_ML_NODE_LENGTHS_INSTANCE(PFX, double, 4, 4, false)      \
    VFT_ML_NODE_LENGTHS_INSTANCE(PFX, double, 4, 8, false)      \
    VFT_ML_NODE_LENGTHS_INSTANCE(PFX, double, 20, 2, true)      \
    VFT_ML_NODE_LENGTHS_INSTANCE(PFX, double, 20, 5, true)       \
    VFT_ML_NODE_LENGTHS_INSTANCE(PFX, double, 20, 8, true)      \
    VFT_ML_NODE_LENGTHS_INSTANCE(PFX, double, 20, 4, false)
#define VFT_ML_QUARTET_INSTANCES_F32(PFX)                       \
    VFT_ML_QUARTET_INSTANCE(PFX, float, 4, 1, false)            \
    VFT_ML_QUARTET_INSTANCE(PFX, float, 4, 4, false)            \
    VFT_ML_QUARTET_INSTANCE(PFX, float, 4, 8, false)            \
    VFT_ML_QUARTET_INSTANCE(PFX, float, 20, 2, true)            \
    VFT_ML_QUARTET_INSTANCE(PFX, float, 20, 5, true)       \
    VFT_ML_QUARTET_INSTANCE(PFX, float, 20, 8, true)            \
    VFT_ML_QUARTET_INSTANCE(PFX, float, 20, 4, false)
#define VFT_ML_QUARTET_INSTANCES_F64(PFX)                       \
    VFT_ML_QUARTET_INSTANCE(PFX, double, 4, 1, false)           \
    VFT_ML_QUARTET_INSTANCE(PFX, double, 4, 4, false)           \
    VFT_ML_QUARTET_INSTANCE(PFX, double, 4, 8, false)           \
    VFT_ML_QUARTET_INSTANCE(PFX, double, 20, 2, true)           \
    VFT_ML_QUARTET_INSTANCE(PFX, double, 20, 5, true)       \
    VFT_ML_QUARTET_INSTANCE(PFX, double, 20, 8, true)           \
    VFT_ML_QUARTET_INSTANCE(PFX, double, 20, 4, false)
#define VFT_ML_HEAVY_INSTANCES(PFX)        \
    VFT_ML_NODE_LENGTHS_INSTANCES(PFX)     \
    VFT_ML_QUARTET_INSTANCES_F32(PFX)      \
    VFT_ML_QUARTET_INSTANCES_F64(PFX)
