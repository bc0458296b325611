// Profile construction kernels: host<->arena transfers, averageProfile, out-profile maintenance.
// Columns of a profile are independent in all of these (NJ.tcc:2074-2124, 951-996), so the natural grid is one
// thread per alignment column; only the out-profile's accumulation over profiles is order-sensitive and is kept
// sequential per column, in list order, like the reference at one thread (NJ.tcc:738-784).
#pragma once
#include "vft_device.h"

// Write one column of one node.  The vectors of a (tile, column) row are packed by lane order, so a write that
// adds or removes a vector moves the vectors of the higher lanes by one slot.  NOT safe for two nodes of the same
// tile concurrently: callers either write one node per launch or use the append path below.
template <typename REAL, int NC>
__device__ __forceinline__ void vft_store_col(const Arena<REAL> &A, int64_t node, int64_t p, REAL w, int code,
                                              const REAL *f) {
    const int lane = (int) (node & 63);
    const int64_t pt = (node >> 6) - A.d.firstProfTile;
    A.profW[vft_w_idx(A.d, pt, p, lane)] = w;
    uint8_t *cb = (uint8_t *) (A.profC + vft_c_idx(A.d, pt, (int) (p >> 4), lane));
    cb[p & 15] = (uint8_t) code;
    const bool vec = w > 0 && code == VFT_NOCODE_;
    const int64_t mi = vft_mask_idx(A.d, pt, p);
    const unsigned long long old = A.vecMask[mi];
    const unsigned long long bit = 1ull << lane;
    const bool had = (old & bit) != 0;
    const int slot = __popcll(old & (bit - 1ull));
    const int nHigher = lane == 63 ? 0 : __popcll(old >> (lane + 1));
    if (vec && !had) {
        for (int s = slot + nHigher - 1; s >= slot; s--)
#pragma unroll
            for (int k = 0; k < NC; k++)
                A.profF[vft_f_idx<REAL>(A.d, pt, p, k, s + 1)] = A.profF[vft_f_idx<REAL>(A.d, pt, p, k, s)];
        A.vecMask[mi] = old | bit;
    } else if (!vec && had) {
        for (int s = slot + 1; s <= slot + nHigher; s++)
#pragma unroll
            for (int k = 0; k < NC; k++)
                A.profF[vft_f_idx<REAL>(A.d, pt, p, k, s - 1)] = A.profF[vft_f_idx<REAL>(A.d, pt, p, k, s)];
        A.vecMask[mi] = old & ~bit;
    }
    if (vec) {
#pragma unroll
        for (int k = 0; k < NC; k++) A.profF[vft_f_idx<REAL>(A.d, pt, p, k, slot)] = f[k];
    }
}

// Append path, phase A: for nodes that have never been written and sit above every written lane of their tile
// (the NJ join loop only ever appends: newnode = maxnode++, NJ.tcc:2904).  Any number of such nodes may be written
// in one launch: weights/codes go to their final place, the mask bit is OR-ed in, the vector is parked in `stash`
// ([batch][nPos][NC]) until k_commit_vectors knows the final slot.
template <typename REAL, int NC>
__device__ __forceinline__ void vft_store_col_append(const Arena<REAL> &A, int64_t node, int64_t p, REAL w, int code,
                                                     const REAL *f, REAL *stash) {
    const int lane = (int) (node & 63);
    const int64_t pt = (node >> 6) - A.d.firstProfTile;
    A.profW[vft_w_idx(A.d, pt, p, lane)] = w;
    uint8_t *cb = (uint8_t *) (A.profC + vft_c_idx(A.d, pt, (int) (p >> 4), lane));
    cb[p & 15] = (uint8_t) code;
    if (w > 0 && code == VFT_NOCODE_) {
        atomicOr(&A.vecMask[vft_mask_idx(A.d, pt, p)], 1ull << lane);
#pragma unroll
        for (int k = 0; k < NC; k++) stash[k] = f[k];
    }
}

// Append path, phase B: every mask bit of the batch is set, slots are final.
template <typename REAL, int NC>
__global__ void k_commit_vectors(Arena<REAL> A, const int64_t *nodes, const REAL *stash) {
    const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.d.nPos) return;
    const int64_t k = blockIdx.y;
    const int64_t node = nodes[k];
    const int lane = (int) (node & 63);
    const int64_t pt = (node >> 6) - A.d.firstProfTile;
    const unsigned long long mask = A.vecMask[vft_mask_idx(A.d, pt, p)];
    if (!((mask >> lane) & 1ull)) return;
    const int slot = __popcll(mask & ((1ull << lane) - 1ull));
    const REAL *src = stash + (k * A.d.nPos + p) * NC;
#pragma unroll
    for (int q = 0; q < NC; q++) A.profF[vft_f_idx<REAL>(A.d, pt, p, q, slot)] = src[q];
}

// staging (row-major w[nPos], c[nPos], f[nPos][NC]) -> arena
template <typename REAL, int NC>
__global__ void k_profile_scatter(Arena<REAL> A, int64_t node, const REAL *w, const uint8_t *c, const REAL *f) {
    const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.d.nPos) return;
    REAL fv[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) fv[k] = f[p * NC + k];
    vft_store_col<REAL, NC>(A, node, p, w[p], (int) c[p], fv);
}

template <typename REAL, int NC>
__global__ void k_profile_gather(Arena<REAL> A, int64_t node, REAL *w, uint8_t *c, REAL *f) {
    const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.d.nPos) return;
    Col<REAL, NC> col;
    vft_load_col<REAL, NC>(A, node, p, col);
    w[p] = col.w;
    c[p] = (uint8_t) col.code;
#pragma unroll
    for (int k = 0; k < NC; k++) f[p * NC + k] = col.vec ? col.f[k] : (REAL) 0;
}

// addToFreq (NJ.tcc:821-833)
template <typename REAL, int NC>
__device__ __forceinline__ void vft_add_to_freq(const Arena<REAL> &A, REAL *fOut, double weight, const Col<REAL, NC> &in) {
    if (in.vec) {
        const REAL wr = (REAL) weight;
#pragma unroll
        for (int k = 0; k < NC; k++) {
            const REAL pr = in.f[k] * wr;
            fOut[k] = fOut[k] + pr;
        }
    } else if (A.dmDist) {
        const REAL wr = (REAL) weight;
#pragma unroll
        for (int k = 0; k < NC; k++) {
            const REAL pr = A.dmCodeFreq[in.code * NC + k] * wr;
            fOut[k] = fOut[k] + pr;
        }
    } else {
#pragma unroll
        for (int k = 0; k < NC; k++)
            if (k == in.code) fOut[k] = (REAL) ((double) fOut[k] + weight);
    }
}

// normalizeFreq (NJ.tcc:843-871)
template <typename REAL, int NC>
__device__ __forceinline__ void vft_normalize_freq(const Arena<REAL> &A, REAL *freq, double tol) {
    double total = 0;
    if (A.dmDist) {
        REAL et[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) et[k] = A.dmEigentot[k];
        total = (double) vft_red4_mul<REAL, NC>(freq, et);
    } else {
#pragma unroll
        for (int k = 0; k < NC; k++) total += (double) freq[k];
    }
    if (total > tol) {
        const REAL inv = (REAL) (1.0 / total);
#pragma unroll
        for (int k = 0; k < NC; k++) freq[k] = freq[k] * inv;
    } else if (!A.dmDist) {
#pragma unroll
        for (int k = 0; k < NC; k++) freq[k] = (REAL) (1.0 / NC);
    } else {
#pragma unroll
        for (int k = 0; k < NC; k++) freq[k] = A.dmCodeFreq[k];
    }
}

// averageProfile (NJ.tcc:2067-2135): grid.y = join index, threads over columns
template <typename REAL, int NC>
__global__ void k_average(Arena<REAL> A, const int64_t *outN, const int64_t *aN, const int64_t *bN,
                          const double *bionj, double tol, REAL *stash /* non-null: append path */) {
    const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.d.nPos) return;
    const int64_t k = blockIdx.y;
    double bw = bionj ? bionj[k] : -1.0;
    if (bw < 0) bw = 0.5;
    Col<REAL, NC> c1, c2;
    vft_load_col<REAL, NC>(A, aN[k], p, c1);
    vft_load_col<REAL, NC>(A, bN[k], p, c2);
    const REAL wo = (REAL) (bw * (double) c1.w + (1 - bw) * (double) c2.w);
    int co = VFT_NOCODE_;
    REAL f[NC];
#pragma unroll
    for (int q = 0; q < NC; q++) f[q] = 0;
    if (wo > 0) {
        if (c1.w > 0 && c1.code != VFT_NOCODE_ && (c2.w <= 0 || c1.code == c2.code)) co = c1.code;
        else if (c1.w <= 0 && c2.w > 0 && c2.code != VFT_NOCODE_) co = c2.code;
        if (co == VFT_NOCODE_) {
            if (c1.w > 0) vft_add_to_freq<REAL, NC>(A, f, (double) c1.w * bw, c1);
            if (c2.w > 0) vft_add_to_freq<REAL, NC>(A, f, (double) c2.w * (1.0 - bw), c2);
            vft_normalize_freq<REAL, NC>(A, f, tol);
        }
    }
    if (stash) vft_store_col_append<REAL, NC>(A, outN[k], p, wo, co, f, stash + (k * A.d.nPos + p) * NC);
    else vft_store_col<REAL, NC>(A, outN[k], p, wo, co, f);
}

// setCodeDist for one column of the out-profile (NJ.tcc:873-898)
template <typename REAL, int NC>
__device__ __forceinline__ void vft_out_codedist(const Arena<REAL> &A, int64_t p, const REAL *f) {
    if (!A.dmDist || !A.outCD) return;
    REAL ev[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) ev[k] = A.dmEigenval[k];
    for (int c = 0; c < NC; c++) {
        REAL cf[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) cf[k] = A.dmCodeFreq[c * NC + k];
        A.outCD[p * NC + c] = vft_red4_mul3<REAL, NC>(f, cf, ev);
    }
}

// outProfile (NJ.tcc:729-815), one thread per column, profiles accumulated in list order
template <typename REAL, int NC>
__global__ void k_outprofile_full(Arena<REAL> A, const int64_t *ids, int64_t n, double tol) {
    const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.d.nPos) return;
    const double inweight = 1.0 / (double) n;
    REAL wo = 0;
    REAL f[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) f[k] = 0;
    for (int64_t t = 0; t < n; t++) {
        Col<REAL, NC> c;
        vft_load_col<REAL, NC>(A, ids[t], p, c);
        wo = (REAL) ((double) wo + (double) c.w * inweight);
        if (c.w > 0) vft_add_to_freq<REAL, NC>(A, f, (double) c.w, c);
    }
    if (wo <= 0) wo = (REAL) 1e-20;
    vft_normalize_freq<REAL, NC>(A, f, tol);
    A.outW[p] = wo;
#pragma unroll
    for (int k = 0; k < NC; k++) A.outF[p * NC + k] = f[k];
    vft_out_codedist<REAL, NC>(A, p, f);
}

// updateOutProfile (NJ.tcc:943-1010)
template <typename REAL, int NC>
__global__ void k_outprofile_update(Arena<REAL> A, int64_t old1, int64_t old2, int64_t newn, int64_t nActiveOld,
                                    double tol) {
    const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.d.nPos) return;
    Col<REAL, NC> c1, c2, cn;
    vft_load_col<REAL, NC>(A, old1, p, c1);
    vft_load_col<REAL, NC>(A, old2, p, c2);
    vft_load_col<REAL, NC>(A, newn, p, cn);
    REAL f[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) f[k] = A.outF[p * NC + k];
    const REAL om = A.outW[p] * (REAL) nActiveOld;   // numeric_t * int64 -> numeric_t, NJ.tcc:963
    const double originalMult = (double) om;
    const double newMult = originalMult + (double) cn.w - (double) c1.w - (double) c2.w;
    REAL wo = (REAL) (newMult / (double) (nActiveOld - 1));
    if (wo <= 0) wo = (REAL) 1e-20;
#pragma unroll
    for (int k = 0; k < NC; k++) f[k] = (REAL) ((double) f[k] * originalMult);
    if (c1.w > 0) {
        const REAL neg = -c1.w;
        vft_add_to_freq<REAL, NC>(A, f, (double) neg, c1);
    }
    if (c2.w > 0) {
        const REAL neg = -c2.w;
        vft_add_to_freq<REAL, NC>(A, f, (double) neg, c2);
    }
    if (cn.w > 0) vft_add_to_freq<REAL, NC>(A, f, (double) cn.w, cn);
    vft_normalize_freq<REAL, NC>(A, f, tol);
    A.outW[p] = wo;
#pragma unroll
    for (int k = 0; k < NC; k++) A.outF[p * NC + k] = f[k];
    vft_out_codedist<REAL, NC>(A, p, f);
}
