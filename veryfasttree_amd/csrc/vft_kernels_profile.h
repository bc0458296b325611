// Profile construction kernels: host<->arena transfers, averageProfile, out-profile maintenance.
// Columns of a profile are independent in all of these (NJ.tcc:2074-2124, 951-996), so the natural grid is one
// thread per alignment column; only the out-profile's accumulation over profiles is order-sensitive and is kept
// sequential per column, in list order, like the reference at one thread (NJ.tcc:738-784).
#pragma once
#include "vft_device.h"

#define VFT_WG_PROF 256

// Write one column of one node.  The explicit weights and the vectors of a (tile, column) row are packed by lane
// order, so a write that adds or removes one moves the entries of the higher lanes by one slot.  NOT safe for two
// nodes of the same tile concurrently: callers either write one node per launch or use the append path below.
// Moves the packed entries (NV values each, at slot s -> idx(k, s)) of the lanes above `lane` when `lane` gains or
// loses its entry; returns the slot of `lane`.
template <int NV, typename IDX>
__device__ __forceinline__ int vft_packed_update(unsigned long long *maskp, int lane, bool want, IDX idx) {
    const unsigned long long old = *maskp;
    const unsigned long long bit = 1ull << lane;
    const bool had = (old & bit) != 0;
    const int slot = __popcll(old & (bit - 1ull));
    const int nHigher = lane == 63 ? 0 : __popcll(old >> (lane + 1));
    if (want && !had) {
        for (int s = slot + nHigher - 1; s >= slot; s--)
#pragma unroll
            for (int k = 0; k < NV; k++) *idx(k, s + 1) = *idx(k, s);
        *maskp = old | bit;
    } else if (!want && had) {
        for (int s = slot + 1; s <= slot + nHigher; s++)
#pragma unroll
            for (int k = 0; k < NV; k++) *idx(k, s - 1) = *idx(k, s);
        *maskp = old & ~bit;
    }
    return slot;
}

template <typename REAL, int NC>
__device__ __forceinline__ void vft_store_col(const Arena<REAL> &A, int64_t node, int64_t p, REAL w, int code,
                                              const REAL *f) {
    const int lane = (int) (node & 63);
    const int64_t pt = (node >> 6) - A.d.firstProfTile;
    uint8_t *cb = (uint8_t *) (A.profC + vft_c_idx(A.d, pt, (int) (p >> 4), lane));
    cb[p & 15] = (uint8_t) code;
    const bool vec = w > 0 && code == VFT_NOCODE_;
    const bool explicitW = !(w == vft_implicit_weight<REAL>(code, vec));
    const int64_t mi = vft_mask_idx(A.d, pt, p);
    const int slotW = vft_packed_update<1>(&A.wMask[mi], lane, explicitW,
                                           [&](int, int s) { return &A.profW[vft_w_idx(A.d, pt, p, s)]; });
    if (explicitW) A.profW[vft_w_idx(A.d, pt, p, slotW)] = w;
    const int slot = vft_packed_update<NC>(&A.vecMask[mi], lane, vec,
                                           [&](int k, int s) { return &A.profF[vft_f_idx<REAL>(A.d, pt, p, k, s)]; });
    if (vec) {
#pragma unroll
        for (int k = 0; k < NC; k++) A.profF[vft_f_idx<REAL>(A.d, pt, p, k, slot)] = f[k];
    }
}

// Append path, phase A: for nodes that have never been written and sit above every written lane of their tile
// (the NJ join loop only ever appends: newnode = maxnode++, NJ.tcc:2904).  Any number of such nodes may be written
// in one launch: codes go to their final place, the mask bits are OR-ed in, the vector and an explicit weight are
// parked in `stash` ([batch][nPos][NC + 1]) until k_commit_vectors knows the final slots.
template <typename REAL, int NC>
__device__ __forceinline__ void vft_store_col_append(const Arena<REAL> &A, int64_t node, int64_t p, REAL w, int code,
                                                     const REAL *f, REAL *stash) {
    const int lane = (int) (node & 63);
    const int64_t pt = (node >> 6) - A.d.firstProfTile;
    uint8_t *cb = (uint8_t *) (A.profC + vft_c_idx(A.d, pt, (int) (p >> 4), lane));
    cb[p & 15] = (uint8_t) code;
    const bool vec = w > 0 && code == VFT_NOCODE_;
    if (!(w == vft_implicit_weight<REAL>(code, vec))) {
        atomicOr(&A.wMask[vft_mask_idx(A.d, pt, p)], 1ull << lane);
        stash[NC] = w;
    }
    if (vec) {
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
    const int64_t mi = vft_mask_idx(A.d, pt, p);
    const unsigned long long below = (1ull << lane) - 1ull;
    const REAL *src = stash + (k * A.d.nPos + p) * (NC + 1);
    const unsigned long long wm = A.wMask[mi];
    if ((wm >> lane) & 1ull) A.profW[vft_w_idx(A.d, pt, p, __popcll(wm & below))] = src[NC];
    const unsigned long long mask = A.vecMask[mi];
    if (!((mask >> lane) & 1ull)) return;
    const int slot = __popcll(mask & below);
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
    if (stash) vft_store_col_append<REAL, NC>(A, outN[k], p, wo, co, f, stash + (k * A.d.nPos + p) * (NC + 1));
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

// ---------------------------------------------------------------------------------------------------------------
// outProfile over ALL active nodes in ascending id order (what the join loop passes, NJ.tcc:3017-3031), nucleotide,
// no distance matrix.  The accumulation over profiles is order-sensitive (float sums, NJ.tcc:738-784), so each
// (column, chain) — chain 0 the weight, chains 1..4 the four frequencies — is one thread walking the nodes in order;
// what is parallel is the 16 columns x 5 chains of a workgroup and the cooperative, coalesced staging of every tile
// (64 nodes x 16 columns: weights, codes, vector mask, packed vectors) through LDS, prefetched one tile ahead.
// k_outprofile_full does the same through per-thread gathers and is ~8x slower; it remains for arbitrary id lists and
// for the matrix / amino-acid case.
__global__ void k_tile_active_masks(const int32_t *parent, int64_t maxnode, unsigned long long *tileMask, int64_t nTiles) {
    const int64_t t = (int64_t) blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (t >= nTiles) return;
    const int64_t v = t * 64 + (threadIdx.x & 63);
    const bool act = v < maxnode && parent[v] < 0;
    const unsigned long long m = __ballot(act);
    if ((threadIdx.x & 63) == 0) tileMask[t] = m;
}

template <typename REAL>
struct OutTileRegs {
    REAL w[4];
    uint4 codes;
    unsigned long long mask, wmask;
    REAL f[4][4];
};

template <typename REAL>
__device__ __forceinline__ void vft_outtile_load(const Arena<REAL> &A, int64_t tile, int chunk, OutTileRegs<REAL> &r) {
    const int tid = threadIdx.x;
    const int64_t p0 = (int64_t) chunk * VFT_CHUNK;
    r.mask = 0;
    // lanes below nSeqs are leaves (codes from leafT); the tile that straddles nSeqs has both kinds
    if (tid < 64 && tile * 64 + tid < A.d.nSeqs) r.codes = A.leafT[vft_leaf_idx(A.d, tile, chunk, tid)];
    if (tile * 64 + 63 < A.d.nSeqs) return;   // pure leaf tile
    const int64_t pt = tile - A.d.firstProfTile;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int idx = tid * 4 + u;           // (column, slot) of the 16 x 64 block of packed explicit weights
        r.w[u] = A.profW[vft_w_idx(A.d, pt, p0 + (idx >> 6), idx & 63)];
    }
    if (tid < 64 && tile * 64 + tid >= A.d.nSeqs) r.codes = A.profC[vft_c_idx(A.d, pt, chunk, tid)];
    const int col = tid >> 4;
    r.mask = A.vecMask[vft_mask_idx(A.d, pt, p0 + col)];
    r.wmask = A.wMask[vft_mask_idx(A.d, pt, p0 + col)];
    const int cnt = __popcll(r.mask);
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int slot = (tid & 15) + 16 * u;
        if (slot < cnt) {
#pragma unroll
            for (int k = 0; k < 4; k++) r.f[u][k] = A.profF[vft_f_idx<REAL>(A.d, pt, p0 + col, k, slot)];
        }
    }
}

template <typename REAL>
__global__ __launch_bounds__(VFT_WG_PROF) void k_outprofile_full_tiled(Arena<REAL> A, const unsigned long long *tileMask,
                                                                       int64_t nTiles, int64_t nActive, double tol) {
    __shared__ REAL sW[VFT_CHUNK][64];
    __shared__ uint4 sCodes[64];
    __shared__ unsigned long long sMask[VFT_CHUNK], sWMask[VFT_CHUNK];
    __shared__ REAL sF[VFT_CHUNK][64][4];
    __shared__ REAL sRes[VFT_CHUNK][5];
    const int tid = threadIdx.x;
    const int chunk = blockIdx.x;
    const int64_t p0 = (int64_t) chunk * VFT_CHUNK;
    const double inweight = 1.0 / (double) nActive;
    const int col = tid / 5, chain = tid % 5;   // threads 0..79 run the chains
    REAL acc = 0;
    // first non-empty tile
    int64_t t = 0;
    while (t < nTiles && tileMask[t] == 0) t++;
    OutTileRegs<REAL> regs;
    if (t < nTiles) vft_outtile_load<REAL>(A, t, chunk, regs);
    while (t < nTiles) {
        const unsigned long long active = tileMask[t];
        const bool leafTile = t * 64 + 63 < A.d.nSeqs;
        __syncthreads();   // previous tile's chains are done with the LDS buffers
        if (leafTile) {
            if (tid < 64) sCodes[tid] = regs.codes;
        } else {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int idx = tid * 4 + u;
                sW[idx >> 6][idx & 63] = regs.w[u];
            }
            if (tid < 64) sCodes[tid] = regs.codes;
            if ((tid & 15) == 0) {
                sMask[tid >> 4] = regs.mask;
                sWMask[tid >> 4] = regs.wmask;
            }
            const int cnt = __popcll(regs.mask);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int slot = (tid & 15) + 16 * u;
                if (slot < cnt) {
#pragma unroll
                    for (int k = 0; k < 4; k++) sF[tid >> 4][slot][k] = regs.f[u][k];
                }
            }
        }
        __syncthreads();
        // prefetch the next non-empty tile into registers while the chains run
        int64_t tn = t + 1;
        while (tn < nTiles && tileMask[tn] == 0) tn++;
        if (tn < nTiles) vft_outtile_load<REAL>(A, tn, chunk, regs);
        if (tid < VFT_CHUNK * 5 && p0 + col < A.d.nPos) {
            unsigned long long m = active;
            while (m) {
                const int l = __ffsll((long long) m) - 1;
                m &= m - 1;
                REAL w;
                int code;
                if (leafTile || t * 64 + l < A.d.nSeqs) {
                    const uint32_t enc = vft_byte(sCodes[l], col);
                    code = vft_decode<4>(enc);
                    w = code != VFT_NOCODE_ ? (REAL) 1 : (REAL) 0;
                } else {
                    code = (int) vft_byte(sCodes[l], col);
                    const unsigned long long below = (1ull << l) - 1ull, wm = sWMask[col];
                    w = ((wm >> l) & 1ull) ? sW[col][__popcll(wm & below)]
                                           : vft_implicit_weight<REAL>(code, (sMask[col] >> l) & 1ull);
                }
                if (chain == 0) {
                    acc = (REAL) ((double) acc + (double) w * inweight);               // NJ.tcc:741
                } else if (w > 0) {
                    const int k = chain - 1;
                    if (code != VFT_NOCODE_) {
                        if (code == k) acc = (REAL) ((double) acc + (double) w);       // addToFreq, NJ.tcc:831
                    } else {
                        const int slot = __popcll(sMask[col] & ((1ull << l) - 1ull));
                        const REAL pr = sF[col][slot][k] * w;                          // vector_add_mult, NJ.tcc:825
                        acc = acc + pr;
                    }
                }
            }
        }
        t = tn;
    }
    __syncthreads();
    if (tid < VFT_CHUNK * 5) sRes[col][chain] = acc;
    __syncthreads();
    if (tid < VFT_CHUNK && p0 + tid < A.d.nPos) {
        const int64_t p = p0 + tid;
        REAL wo = sRes[tid][0];
        if (wo <= 0) wo = (REAL) 1e-20;
        REAL f[4] = {sRes[tid][1], sRes[tid][2], sRes[tid][3], sRes[tid][4]};
        vft_normalize_freq<REAL, 4>(A, f, tol);
        A.outW[p] = wo;
#pragma unroll
        for (int k = 0; k < 4; k++) A.outF[p * 4 + k] = f[k];
    }
}
