// ML-phase kernels: P(t) tables, pairLogLk and posteriorProfile, batched over independent node pairs
// (one tree level of recomputeMLProfiles / treeLogLk is one launch).
//
// P(t) is never materialised in HBM: each workgroup builds the tables its pair needs — pSame/pDiff per rate
// category for Jukes-Cantor (NJ.tcc:2005-2018), exp(eigenvalue * rate * length) per (category, code) for matrix
// models (NJ.tcc:2020-2038) — in LDS, next to the 4x4 / 20x20 codeFreq / eigeninv matrices.
#pragma once
#include "vft_device.h"
#include "vft_kernels_profile.h"

#define VFT_ML_WG 128
#define VFT_MAXRATES 64
#define VFT_LK_UNDERFLOW 1.0e-4              /* Constants.h:13 */
#define VFT_LK_UNDERFLOW_INV 1.0e4           /* Constants.h:14 */
#define VFT_LOG_LK_UNDERFLOW 9.21034037197618 /* Constants.h:15 */

// expEigenRates (NJ.tcc:2020-2038, NDEBUG branch with fastexp level 0) into LDS: out[r*NC + j]
template <typename REAL, int NC>
__device__ __forceinline__ void vft_exp_eigen_rates(const Arena<REAL> &A, double length, double minRel, REAL *out) {
    for (int t = threadIdx.x; t < A.nRates * NC; t += blockDim.x) {
        const int r = t / NC, j = t % NC;
        double relLen = length * (double) A.rates[r];
        if (relLen < minRel) relLen = minRel;
        const REAL rl = (REAL) relLen;           // vector_multiply_by takes numeric_t
        const REAL x = A.tmEigenval[j] * rl;
        out[t] = (REAL) exp((double) x);
    }
}

__device__ __forceinline__ void vft_psame_pdiff(double length, double rate, double &pSame, double &pDiff) {
    pSame = 0.25 + 0.75 * exp((-4.0 / 3.0) * fabs(length * rate));
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
        for (int r = threadIdx.x; r < A.nRates; r += blockDim.x) vft_psame_pdiff(len, (double) A.rates[r], pS[r], pD[r]);
    } else {
        vft_exp_eigen_rates<REAL, NC>(A, len, minRel, expeig);
    }
    __syncthreads();
    double lk = 1.0, loglk = 0.0;
    for (int64_t p = threadIdx.x; p < A.d.nPos; p += blockDim.x) {
        Col<REAL, NC> c1, c2;
        vft_load_col<REAL, NC>(A, a, p, c1);
        vft_load_col<REAL, NC>(A, b, p, c2);
        const int r = A.ratecat[p];
        double lkAB = 0;
        bool skip = false;
        if (jc) {
            const double wA = (double) c1.w, wB = (double) c2.w;
            const double pSame = pS[r], pDiff = pD[r];
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
        } else {
            if (c1.w == 0 && c2.w == 0 && c1.code == VFT_NOCODE_ && c2.code == VFT_NOCODE_) {
                skip = true;   // gap against gap: likelihood 1 (NJ.tcc:1277-1281)
            } else {
                REAL fA[NC], fB[NC];
                vft_model_freq<REAL, NC>(A, c1, true, fA);
                vft_model_freq<REAL, NC>(A, c2, true, fB);
                const REAL *ee = expeig + r * NC;
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
            }
        }
        if (skip) {
            if (siteLk) siteLk[k * A.d.nPos + p] = 1.0;
            continue;
        }
        if (siteLk) siteLk[k * A.d.nPos + p] = lkAB;
        lk *= lkAB;
        // (lk > 0: the reference asserts lkAB > 0 and would spin forever otherwise, NJ.tcc:1257-1262; profiles that
        //  are not in the model's eigenbasis can produce that, and a kernel must terminate)
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

// posteriorProfile (NJ.tcc:2137-2447, exact ML).  grid.y = triple index, threads over columns.
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_ML_WG) void k_posterior(Arena<REAL> A, const int64_t *outN, const int64_t *aN,
                                                         const int64_t *bN, const double *len1A, const double *len2A,
                                                         double minLen, double minRel,
                                                         REAL *stash /* non-null: append path */) {
    __shared__ REAL ee1[VFT_MAXRATES * NC], ee2[VFT_MAXRATES * NC];
    __shared__ double pS1[VFT_MAXRATES], pD1[VFT_MAXRATES], pS2[VFT_MAXRATES], pD2[VFT_MAXRATES];
    const int64_t k = blockIdx.y;
    double len1 = len1A[k], len2 = len2A[k];
    if (len1 < minLen) len1 = minLen;
    if (len2 < minLen) len2 = minLen;
    const bool jc = A.tmStat == nullptr;
    if (jc) {
        for (int r = threadIdx.x; r < A.nRates; r += blockDim.x) {
            vft_psame_pdiff(len1, (double) A.rates[r], pS1[r], pD1[r]);
            vft_psame_pdiff(len2, (double) A.rates[r], pS2[r], pD2[r]);
        }
    } else {
        vft_exp_eigen_rates<REAL, NC>(A, len1, minRel, ee1);
        vft_exp_eigen_rates<REAL, NC>(A, len2, minRel, ee2);
    }
    __syncthreads();
    const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.d.nPos) return;
    Col<REAL, NC> c1, c2;
    vft_load_col<REAL, NC>(A, aN[k], p, c1);
    vft_load_col<REAL, NC>(A, bN[k], p, c2);
    const int r = A.ratecat[p];
    REAL wo = (REAL) 1.0;
    int co = VFT_NOCODE_;
    REAL f[NC];
#pragma unroll
    for (int j = 0; j < NC; j++) f[j] = 0;
    if (jc) {
        const double w1 = (double) c1.w, w2 = (double) c2.w;
        const double PS1 = pS1[r], PD1 = pD1[r], PS2 = pS2[r], PD2 = pD2[r];
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
            const REAL *e1 = ee1 + r * NC, *e2 = ee2 + r * NC;
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
    vft_stash_col<REAL, NC>(A, outN[k], p, wo, co, f, stash + (k * A.d.nPos + p) * (NC + 1));
}
