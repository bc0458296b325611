// Profile construction kernels: host<->arena transfers, averageProfile, out-profile maintenance.
// Columns of a profile are independent in all of these (NJ.tcc:2074-2124, 951-996), so the natural grid is one
// thread per alignment column; only the out-profile's accumulation over profiles is order-sensitive and is kept
// sequential per column, in list order, like the reference at one thread (NJ.tcc:738-784).
#pragma once
#include "vft_iterate_add.h"
#include "vft_device.h"

#define VFT_WG_PROF 256

// ---------------------------------------------------------------------------------------------------------------
// Writing profiles.  A kernel that produces nodes (average, posterior, upload) writes each column's code straight
// into profC (dense, per lane) and parks (f[0..NC), w) in a row-major stash [batch][nPos][NC + 1]; k_tile_commit
// then rebuilds the packed streams of every tile that received nodes (vft_layout.h).
template <typename REAL, int NC>
__device__ __forceinline__ void vft_stash_col(const Arena<REAL> &A, int64_t node, int64_t p, REAL w, int code,
                                              const REAL *f, REAL *stashRow) {
    const int lane = (int) (node & 63);
    const int64_t pt = (node >> 6) - A.d.firstProfTile;
    uint8_t *cb = (uint8_t *) (A.profC + vft_c_idx(A.d, pt, (int) (p >> 4), lane));
    cb[p & 15] = (uint8_t) code;
#pragma unroll
    for (int k = 0; k < NC; k++) stashRow[k] = f[k];
    stashRow[NC] = w;
}

// bytes of scratch one workgroup of k_tile_commit needs (new masks, offsets and both streams of one tile)
static inline size_t vft_commit_scratch_bytes(const VftDims &d, size_t rs) {
    const size_t cap = (size_t) d.nPosPad * VFT_TILE;
    const size_t b = (size_t) d.nPosPad * (sizeof(ColMask) + sizeof(ColOff)) + cap * (size_t) (d.nCodes + 1) * rs;
    return (b + 255) & ~(size_t) 255;
}

#define VFT_COMMIT_WG 1024
// coalesced copy of n elements with several loads in flight per thread (the single workgroup that owns a tile is
// latency-bound otherwise: one load -> one store per iteration is a chain of L2 round trips)
template <typename T>
__device__ __forceinline__ void vft_block_copy(T *dst, const T *src, int64_t n) {
    const int64_t step = VFT_COMMIT_WG;
    int64_t i = threadIdx.x;
    for (; i + 7 * step < n; i += 8 * step) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = src[i + u * step];
#pragma unroll
        for (int u = 0; u < 8; u++) dst[i + u * step] = v[u];
    }
    for (; i < n; i += step) dst[i] = src[i];
}

// One workgroup per tile that received nodes.  order[segFirst[s] .. segFirst[s+1]) are the batch indices (into
// nodes[] / the stash) of the nodes of segment s, all in one tile.  The tile's new masks, prefix offsets and streams
// are assembled in scratch from (old streams of the untouched lanes) + (stash of the written lanes) and then copied
// over the old ones, so nodes may be appended or rewritten in any lane, any number per launch.
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_COMMIT_WG) void k_tile_commit(Arena<REAL> A, const int64_t *nodes, const int32_t *order,
                                                               const int32_t *segFirst, const REAL *stash, char *scratch,
                                                               size_t scratchStride) {
    __shared__ int sLaneK[VFT_TILE];
    __shared__ unsigned long long sWrite;
    __shared__ uint32_t sWaveV[VFT_COMMIT_WG / 64], sWaveW[VFT_COMMIT_WG / 64], sTot[2];
    const int tid = threadIdx.x;
    const int first = segFirst[blockIdx.x], cnt = segFirst[blockIdx.x + 1] - first;
    if (tid < VFT_TILE) sLaneK[tid] = -1;
    __syncthreads();
    if (tid < cnt) {
        const int k = order[first + tid];
        sLaneK[(int) (nodes[k] & 63)] = k;
    }
    __syncthreads();
    if (tid < VFT_TILE) {
        const unsigned long long wm = __ballot(sLaneK[tid] >= 0);
        if (tid == 0) sWrite = wm;
    }
    __syncthreads();
    const unsigned long long W = sWrite;
    const int64_t pt = (nodes[order[first]] >> 6) - A.d.firstProfTile;
    const int64_t nPos = A.d.nPos, nPosPad = A.d.nPosPad, mi0 = vft_meta_idx(A.d, pt, 0);
    const int64_t fBase = vft_fstream_base(A.d, pt), wBase = vft_wstream_base(A.d, pt);
    char *sc = scratch + (size_t) blockIdx.x * scratchStride;
    ColMask *nm = (ColMask *) sc;
    ColOff *no = (ColOff *) (nm + nPosPad);
    REAL *fS = (REAL *) (no + nPosPad);
    REAL *wS = fS + (size_t) nPosPad * VFT_TILE * NC;
    // 1. new masks; every thread owns a contiguous run of columns
    const int64_t PP = (nPosPad + VFT_COMMIT_WG - 1) / VFT_COMMIT_WG;
    const int64_t pa = (int64_t) tid * PP < nPosPad ? (int64_t) tid * PP : nPosPad, pb = pa + PP < nPosPad ? pa + PP : nPosPad;
    uint32_t cV = 0, cW = 0;
    for (int64_t p = pa; p < pb; p++) {
        const ColMask old = A.colMask[mi0 + p];
        unsigned long long nv = old.vec & ~W, nw = old.w & ~W;
        if (p < nPos) {
            unsigned long long m = W;
            while (m) {
                const int L = __ffsll((long long) m) - 1;
                m &= m - 1;
                const REAL w = stash[((int64_t) sLaneK[L] * nPos + p) * (NC + 1) + NC];
                const int code = (int) vft_byte(A.profC[vft_c_idx(A.d, pt, (int) (p >> 4), L)], (int) (p & 15));
                const bool vec = w > 0 && code == VFT_NOCODE_;
                if (vec) nv |= 1ull << L;
                if (!(w == vft_implicit_weight<REAL>(code, vec))) nw |= 1ull << L;
            }
        }
        nm[p].vec = nv;
        nm[p].w = nw;
        cV += (uint32_t) __popcll(nv);
        cW += (uint32_t) __popcll(nw);
    }
    // exclusive scan over the runs: inside the wavefront by shuffles, across the 16 wavefronts through LDS
    uint32_t iV = cV, iW = cW;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(iV, off, 64), w = __shfl_up(iW, off, 64);
        if ((tid & 63) >= off) {
            iV += v;
            iW += w;
        }
    }
    if ((tid & 63) == 63) {
        sWaveV[tid >> 6] = iV;
        sWaveW[tid >> 6] = iW;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t aV = 0, aW = 0;
        for (int t = 0; t < VFT_COMMIT_WG / 64; t++) {
            const uint32_t v = sWaveV[t], w = sWaveW[t];
            sWaveV[t] = aV;
            sWaveW[t] = aW;
            aV += v;
            aW += w;
        }
        sTot[0] = aV;
        sTot[1] = aW;
    }
    __syncthreads();
    // 2. offsets of every column (each thread walks its own run) ...
    {
        uint32_t oV = sWaveV[tid >> 6] + iV - cV, oW = sWaveW[tid >> 6] + iW - cW;
        for (int64_t p = pa; p < pb; p++) {
            no[p].vec = oV;
            no[p].w = oW;
            oV += (uint32_t) __popcll(nm[p].vec);
            oW += (uint32_t) __popcll(nm[p].w);
        }
    }
    __syncthreads();
    // ... and the streams into scratch, one wavefront per column, lane = node of the tile: every element is an
    // independent copy (no serial walk over a column's bits), consecutive lanes write consecutive stream slots
    {
        const int L = tid & 63;
        const unsigned long long below = (1ull << L) - 1ull;
        const bool mine = (W >> L) & 1ull;
        const int64_t stashRow = mine ? (int64_t) sLaneK[L] * nPos : 0;
#pragma unroll 2
        for (int64_t p = tid >> 6; p < nPos; p += VFT_COMMIT_WG / 64) {
            const ColMask m = nm[p], old = A.colMask[mi0 + p];
            const ColOff oo = A.colOff[mi0 + p], on = no[p];
            if ((m.vec >> L) & 1ull) {
                const int nOld = __popcll(old.vec), rOld = __popcll(old.vec & below);
                const int nNew = __popcll(m.vec), rNew = __popcll(m.vec & below);
                REAL v[NC];
                if (mine) {
                    const REAL *src = stash + (stashRow + p) * (NC + 1);
#pragma unroll
                    for (int k = 0; k < NC; k++) v[k] = src[k];
                } else {
#pragma unroll
                    for (int k = 0; k < NC; k++) v[k] = A.profF[fBase + vft_fidx<REAL, NC>(oo.vec, nOld, rOld, k)];
                }
#pragma unroll
                for (int k = 0; k < NC; k++) fS[vft_fidx<REAL, NC>(on.vec, nNew, rNew, k)] = v[k];
            }
            if ((m.w >> L) & 1ull)
                wS[on.w + __popcll(m.w & below)] = mine ? stash[(stashRow + p) * (NC + 1) + NC]
                                                        : A.profW[wBase + oo.w + __popcll(old.w & below)];
        }
    }
    __syncthreads();   // every read of the old streams is done, scratch is complete
    // 3. copy over the tile (coalesced, 16-byte elements where the sizes allow)
    vft_block_copy<ColMask>(A.colMask + mi0, nm, nPosPad);
    vft_block_copy<ColOff>(A.colOff + mi0, no, nPosPad);
    const int64_t nF = (int64_t) sTot[0] * NC, nW = sTot[1];
    {
        const int64_t per = 16 / (int64_t) sizeof(REAL), nF16 = nF / per;
        vft_block_copy<uint4>((uint4 *) (A.profF + fBase), (const uint4 *) fS, nF16);
        for (int64_t i = nF16 * per + tid; i < nF; i += VFT_COMMIT_WG) A.profF[fBase + i] = fS[i];
        const int64_t nW16 = nW / per;
        vft_block_copy<uint4>((uint4 *) (A.profW + wBase), (const uint4 *) wS, nW16);
        for (int64_t i = nW16 * per + tid; i < nW; i += VFT_COMMIT_WG) A.profW[wBase + i] = wS[i];
    }
}

// staging (row-major w[nPos], c[nPos], f[nPos][NC]) -> arena
template <typename REAL, int NC>
__global__ void k_profile_scatter(Arena<REAL> A, int64_t node, const REAL *w, const uint8_t *c, const REAL *f, REAL *stash) {
    const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.d.nPos) return;
    REAL fv[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) fv[k] = f[p * NC + k];
    vft_stash_col<REAL, NC>(A, node, p, w[p], (int) c[p], fv, stash + p * (NC + 1));
}

template <typename REAL, int NC>
__global__ void k_profile_gather(Arena<REAL> A, int64_t node, REAL *w, uint8_t *c, REAL *f) {
    const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.d.nPos) return;
    Col<REAL, NC> col;
    vft_load_col_ml<REAL, NC>(A, node, p, col);
    w[p] = col.w;
    c[p] = (uint8_t) col.code;
#pragma unroll
    for (int k = 0; k < NC; k++) f[p * NC + k] = col.vec ? col.f[k] : (REAL) 0;
}

// addToFreq (NJ.tcc:821-833)
template <typename REAL, int NC, typename DM>
__device__ __forceinline__ void vft_add_to_freq(const Arena<REAL> &A, REAL *fOut, double weight, const Col<REAL, NC> &in, const DM &T) {
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
            const REAL pr = T.codeFreq[in.code * NC + k] * wr;
            fOut[k] = fOut[k] + pr;
        }
    } else {
#pragma unroll
        for (int k = 0; k < NC; k++)
            if (k == in.code) fOut[k] = (REAL) ((double) fOut[k] + weight);
    }
}

// normalizeFreq (NJ.tcc:843-871)
template <typename REAL, int NC, typename DM>
__device__ __forceinline__ void vft_normalize_freq(const Arena<REAL> &A, REAL *freq, double tol, const DM &T) {
    double total = 0;
    if (A.dmDist) {
        REAL et[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) et[k] = T.eigentot[k];
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
        for (int k = 0; k < NC; k++) freq[k] = T.codeFreq[k];
    }
}

// averageProfile (NJ.tcc:2067-2135) of one column
template <typename REAL, int NC>
__device__ __forceinline__ void vft_add_to_freq(const Arena<REAL> &A, REAL *fOut, double weight, const Col<REAL, NC> &in) {
    vft_add_to_freq<REAL, NC, DmGlobal<REAL>>(A, fOut, weight, in, DmGlobal<REAL>(A));
}
template <typename REAL, int NC>
__device__ __forceinline__ void vft_normalize_freq(const Arena<REAL> &A, REAL *freq, double tol) {
    vft_normalize_freq<REAL, NC, DmGlobal<REAL>>(A, freq, tol, DmGlobal<REAL>(A));
}

template <typename REAL, int NC, typename DM>
__device__ __forceinline__ void vft_average_col(const Arena<REAL> &A, const Col<REAL, NC> &c1, const Col<REAL, NC> &c2, double bw,
                                                double tol, REAL &wo, int &co, REAL *f, const DM &T) {
    wo = (REAL) (bw * (double) c1.w + (1 - bw) * (double) c2.w);
    co = VFT_NOCODE_;
#pragma unroll
    for (int q = 0; q < NC; q++) f[q] = 0;
    if (wo > 0) {
        if (c1.w > 0 && c1.code != VFT_NOCODE_ && (c2.w <= 0 || c1.code == c2.code)) co = c1.code;
        else if (c1.w <= 0 && c2.w > 0 && c2.code != VFT_NOCODE_) co = c2.code;
        if (co == VFT_NOCODE_) {
            if (c1.w > 0) vft_add_to_freq<REAL, NC, DM>(A, f, (double) c1.w * bw, c1, T);
            if (c2.w > 0) vft_add_to_freq<REAL, NC, DM>(A, f, (double) c2.w * (1.0 - bw), c2, T);
            vft_normalize_freq<REAL, NC, DM>(A, f, tol, T);
        }
    }
}
template <typename REAL, int NC>
__device__ __forceinline__ void vft_average_col(const Arena<REAL> &A, const Col<REAL, NC> &c1, const Col<REAL, NC> &c2, double bw,
                                                double tol, REAL &wo, int &co, REAL *f) {
    vft_average_col<REAL, NC, DmGlobal<REAL>>(A, c1, c2, bw, tol, wo, co, f, DmGlobal<REAL>(A));
}

// The same for nucleotides without a distance matrix, written with selects instead of branches: the kinds of column (gap, code,
// vector) differ from lane to lane, and a latency-bound caller with one wavefront per SIMD (k_walk_server) pays for every branch
// region a wavefront walks through.  Same operations on the same operands in the same order as vft_average_col / vft_add_to_freq /
// vft_normalize_freq above; results of the paths a lane does not take are computed and dropped.
template <typename REAL>
__device__ __forceinline__ void vft_average_col_nt_select(const Col<REAL, 4> &c1, const Col<REAL, 4> &c2, double tol, REAL &wo, int &co,
                                                          REAL *f) {
    const double bw = 0.5;
    wo = (REAL) (bw * (double) c1.w + (1 - bw) * (double) c2.w);
    const bool pos = wo > 0, p1 = c1.w > 0, p2 = c2.w > 0;
    const bool k1 = p1 && c1.code != VFT_NOCODE_ && (c2.w <= 0 || c1.code == c2.code);
    const bool k2 = !k1 && c1.w <= 0 && p2 && c2.code != VFT_NOCODE_;
    co = pos ? (k1 ? c1.code : k2 ? c2.code : VFT_NOCODE_) : VFT_NOCODE_;
    const bool mix = pos && co == VFT_NOCODE_;
    const double wt1 = (double) c1.w * bw, wt2 = (double) c2.w * (1.0 - bw);
    const REAL wr1 = (REAL) wt1, wr2 = (REAL) wt2;
    REAL g[4];
    double total = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        REAL v = 0;
        {
            const REAL pr = c1.f[k] * wr1;
            const REAL asVec = v + pr;
            const REAL asCode = (REAL) ((double) v + wt1);
            v = p1 ? (c1.vec ? asVec : (k == c1.code ? asCode : v)) : v;
        }
        {
            const REAL pr = c2.f[k] * wr2;
            const REAL asVec = v + pr;
            const REAL asCode = (REAL) ((double) v + wt2);
            v = p2 ? (c2.vec ? asVec : (k == c2.code ? asCode : v)) : v;
        }
        g[k] = v;
        total += (double) v;
    }
    const REAL inv = (REAL) (1.0 / total);
    const bool scale = total > tol;
#pragma unroll
    for (int k = 0; k < 4; k++) f[k] = mix ? (scale ? g[k] * inv : (REAL) (1.0 / 4)) : (REAL) 0;
}

// averageProfile: grid.y = join index, threads over columns
template <typename REAL, int NC>
__global__ void k_average(Arena<REAL> A, const int64_t *outN, const int64_t *aN, const int64_t *bN,
                          const double *bionj, double tol, REAL *stash /* null: write the node's dense row */) {
    const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.d.nPos) return;
    const int64_t k = blockIdx.y;
    double bw = bionj ? bionj[k] : -1.0;
    if (bw < 0) bw = 0.5;
    Col<REAL, NC> c1, c2;
    vft_load_col_ml<REAL, NC>(A, aN[k], p, c1);
    vft_load_col_ml<REAL, NC>(A, bN[k], p, c2);
    REAL wo, f[NC];
    int co;
    vft_average_col<REAL, NC>(A, c1, c2, bw, tol, wo, co, f);
    if (stash) {
        vft_stash_col<REAL, NC>(A, outN[k], p, wo, co, f, stash + (k * A.d.nPos + p) * (NC + 1));
    } else {
        vft_store_col_ml<REAL, NC>(A, outN[k], p, wo, co, f);
        if (p == 0) A.mlIs[outN[k] - A.d.nSeqs] = 1;
    }
}

// A CHAIN of n unweighted averages where later ones may read earlier results (recomputeProfile of a node and then of
// its parent, up-profiles down a path): a column of an average depends on that column of its inputs only, so one thread
// takes its column through the whole chain - one launch instead of n dependent ones.  Rows only (the refinement phase).
// direct[k] bit 0 / 1: input a / b of op k is the output of an earlier op of this chain (its row flag may not be
// visible to other workgroups yet, so it is read as a row unconditionally).
template <typename REAL, int NC>
__global__ void k_average_chain(Arena<REAL> A, const int64_t *outN, const int64_t *aN, const int64_t *bN, const uint8_t *direct,
                                int32_t n, double tol, const int32_t *chainOff) {
    const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.d.nPos) return;
    // several independent chains in one launch (blockIdx.y; the lanes of a subtree schedule): ops [chainOff[y], chainOff[y + 1])
    if (chainOff) {
        const int32_t k0 = chainOff[blockIdx.y];
        n = chainOff[blockIdx.y + 1] - k0;
        if (n <= 0) return;
        outN += k0;
        aN += k0;
        bN += k0;
        direct += k0;
    }
    // Every op was three dependent memory rounds (its ids, its two columns, and - for an input that is the previous op's output -
    // a store that has to land before it can be read back): 5 us per op, 124 000 launches in a 3 000-taxon protein run.  The
    // next op's ids are fetched while the current one computes, and the previous op's column is handed over in registers.
    int64_t prevOut = -1, aNext = aN[0], bNext = bN[0], oNext = outN[0];
    uint8_t dNext = direct[0];
    Col<REAL, NC> prev;
    prev.w = 0;
    prev.code = VFT_NOCODE_;
    prev.vec = false;
#pragma unroll
    for (int q = 0; q < NC; q++) prev.f[q] = 0;
    for (int32_t k = 0; k < n; k++) {
        const int64_t a = aNext, b = bNext, o = oNext;
        const uint8_t d = dNext;
        if (k + 1 < n) {
            aNext = aN[k + 1];
            bNext = bN[k + 1];
            oNext = outN[k + 1];
            dNext = direct[k + 1];
        }
        Col<REAL, NC> c1, c2;
        if (a == prevOut) c1 = prev;
        else if (d & 1) vft_load_row<REAL, NC>(A, a, p, c1);
        else vft_load_col_ml<REAL, NC>(A, a, p, c1);
        if (b == prevOut) c2 = prev;
        else if (d & 2) vft_load_row<REAL, NC>(A, b, p, c2);
        else vft_load_col_ml<REAL, NC>(A, b, p, c2);
        REAL wo, f[NC];
        int co;
        vft_average_col<REAL, NC>(A, c1, c2, 0.5, tol, wo, co, f);
        vft_store_col_ml<REAL, NC>(A, o, p, wo, co, f);
        prevOut = o;
        prev.w = wo;
        prev.code = co;
        prev.vec = wo > 0 && co == VFT_NOCODE_;   // what vft_load_row would find
#pragma unroll
        for (int q = 0; q < NC; q++) prev.f[q] = f[q];
    }
    // (the row flags are raised by k_mark_rows afterwards: raising them here would race with workgroups that are still
    //  on an earlier op and read the same node through its flag)
}

// vft_set_profile_rows: every internal node's current profile copied from the tile streams into its plain row, once
template <typename REAL, int NC>
__global__ void k_rows_from_tiles(Arena<REAL> A, int64_t first) {
    const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.d.nPos) return;
    const int64_t node = first + blockIdx.y;
    Col<REAL, NC> c;
    vft_load_col<REAL, NC>(A, node, p, c);
    vft_store_col_ml<REAL, NC>(A, node, p, c.w, c.code, c.f);
}

__device__ __forceinline__ bool vft_same_bits(float a, float b) { return __float_as_uint(a) == __float_as_uint(b); }
__device__ __forceinline__ bool vft_same_bits(double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b); }

// are two nodes' rows the same profile, bit for bit?  flags[k] |= 1 when rows a[k], b[k] differ at some column (weight, code, or
// vector under a vector column).  The speculative SPR rounds (host/MLLengths.h) ask whether an attempt that changed nothing in the
// tree left the profiles it recomputed as they were.
template <typename REAL, int NC>
__global__ void k_rows_differ(Arena<REAL> A, const int64_t *aN, const int64_t *bN, int32_t *flags) {
    const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.d.nPos) return;
    const int64_t k = blockIdx.y;
    Col<REAL, NC> c1, c2;
    vft_load_col_ml<REAL, NC>(A, aN[k], p, c1);
    vft_load_col_ml<REAL, NC>(A, bN[k], p, c2);
    bool diff = c1.code != c2.code || c1.vec != c2.vec || !vft_same_bits(c1.w, c2.w);
    if (!diff && c1.vec) {
#pragma unroll
        for (int q = 0; q < NC; q++) diff = diff || !vft_same_bits(c1.f[q], c2.f[q]);
    }
    if (diff) flags[k] = 1;   // (every writer stores the same value; the buffer may be host-mapped memory)
}

static __global__ void k_mark_rows(uint8_t *mlIs, const int64_t *nodes, int32_t n, int64_t nSeqs) {
    const int32_t k = (int32_t) (blockIdx.x * blockDim.x + threadIdx.x);
    if (k < n) mlIs[nodes[k] - nSeqs] = 1;
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

// outProfile in PARTS (the multi-GPU shard of SURVEY 8e: "per-GPU partial sums -> all-gather -> fixed-order sum"; the reference's
// threaded outProfile is the same idea with per-thread partials over a dynamic schedule, NJ.tcc:763-783).  The active list is cut into
// nParts blocks; k_outprofile_partial adds up one block exactly as k_outprofile_full adds up the whole list (weights with the in-weight
// 1 / nTotal of the WHOLE list, frequencies by vft_add_to_freq) and leaves the raw sums - part[p * (1 + NC)] = weight, then the NC
// frequencies; k_outprofile_finish adds the parts in block order (numeric_t additions: the reference merges its partials with
// vector_add, NJ.tcc:782), then floors the weight, normalises and fills codeDist as k_outprofile_full's tail does.  The result depends
// on nParts (as the reference's depends on its thread count) and on nothing else: not on how many ranks computed the parts.
template <typename REAL, int NC>
__global__ void k_outprofile_partial(Arena<REAL> A, const int64_t *ids, int64_t n, int64_t nTotal, REAL *part) {
    const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.d.nPos) return;
    const double inweight = 1.0 / (double) nTotal;
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
    part[p * (1 + NC)] = wo;
#pragma unroll
    for (int k = 0; k < NC; k++) part[p * (1 + NC) + 1 + k] = f[k];
}

template <typename REAL, int NC>
__global__ void k_outprofile_finish(Arena<REAL> A, const REAL *parts, int32_t nParts, double tol) {
    const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.d.nPos) return;
    const int64_t stride = A.d.nPos * (1 + NC);
    REAL wo = 0;
    REAL f[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) f[k] = 0;
    for (int32_t b = 0; b < nParts; b++) {
        const REAL *src = parts + (int64_t) b * stride + p * (1 + NC);
        wo = wo + src[0];
#pragma unroll
        for (int k = 0; k < NC; k++) f[k] = f[k] + src[1 + k];
    }
    if (wo <= 0) wo = (REAL) 1e-20;
    vft_normalize_freq<REAL, NC>(A, f, tol);
    A.outW[p] = wo;
#pragma unroll
    for (int k = 0; k < NC; k++) A.outF[p * NC + k] = f[k];
    vft_out_codedist<REAL, NC>(A, p, f);
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
    vft_load_col_ml<REAL, NC>(A, old1, p, c1);   // (rows of nodes joined by vft_join_fused whose tiles are not rebuilt yet)
    vft_load_col_ml<REAL, NC>(A, old2, p, c2);
    vft_load_col_ml<REAL, NC>(A, newn, p, cn);
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
// One join of the NJ loop in ONE launch (NJ.tcc:2904-2909 + 3003-3008 + 3012-3042): the state change of k_join_nodes,
// averageProfile of the two children, profileDist(new, new) (self distance / self weight) and - unless a full recompute
// follows - updateOutProfile.  All four walk the same columns of the same two children; as separate launches they were
// five dependent kernels of 3-35 us each, most of it launch latency and the tile re-pack.
// The new profile goes to the node's plain row (vft_layout.h) AND into slot `slot` of the pending stash with its codes
// in profC: the tile streams are rebuilt lazily, for up to 64 joined nodes at once, right before something that reads
// tile streams runs (a sweep, a full out-profile; vft_api.hip: flush_pending).  Pair lists, single out-distances and
// the next joins read the row.  One workgroup; dynamic LDS: 2 * nPosPad doubles.
// the body of one join, shared by k_join_fused (arguments from the host) and the join engine's k_nj_join (arguments from the
// device-resident state, vft_kernels_njengine.h); every thread of the workgroup calls, blockDim.x = VFT_WG_PROF
template <typename REAL, int NC>
__device__ __forceinline__ void vft_join_body(const Arena<REAL> &A, int64_t i, int64_t j, int64_t newn, REAL diameter,
                                              int32_t staleStamp, int64_t nActiveOld, int32_t updateOut, double tol,
                                              REAL *stash, int64_t *pendIds, int32_t slot, double *jfLds, double *newOut = nullptr) {
    // newOut != nullptr (with updateOut): also profileDist(new node, UPDATED out-profile) - what setOutDistance(new node) needs
    // right after the join (NJ.tcc:1020) - from the columns this workgroup holds anyway: newOut[0] = top, newOut[1] = denom of
    // NJ.tcc:1168-1183 (LDS: two more arrays of nPosPad doubles behind the first two)
    __shared__ double res[4];
    double *sW = jfLds, *sT = jfLds + A.d.nPosPad;
    double *sW2 = jfLds + 2 * A.d.nPosPad, *sT2 = jfLds + 3 * A.d.nPosPad;
    const int64_t nPos = A.d.nPos;
    if (threadIdx.x == 0) {
        A.parent[i] = (int32_t) newn;
        A.parent[j] = (int32_t) newn;
        A.diameter[newn] = diameter;
        A.outDist[newn] = 0;
        A.nOutActive[newn] = staleStamp;
        pendIds[slot] = newn;
        A.mlIs[newn - A.d.nSeqs] = 1;
    }
    for (int64_t p = threadIdx.x; p < nPos; p += VFT_WG_PROF) {
        Col<REAL, NC> c1, c2, cn;
        vft_load_col_ml<REAL, NC>(A, i, p, c1);
        vft_load_col_ml<REAL, NC>(A, j, p, c2);
        vft_average_col<REAL, NC>(A, c1, c2, 0.5, tol, cn.w, cn.code, cn.f);
        cn.vec = cn.w > 0 && cn.code == VFT_NOCODE_;
        vft_store_col_ml<REAL, NC>(A, newn, p, cn.w, cn.code, cn.f);
        vft_stash_col<REAL, NC>(A, newn, p, cn.w, cn.code, cn.f, stash + ((int64_t) slot * nPos + p) * (NC + 1));
        // profileDist(new, new): the addends of this column (NJ.tcc:1175-1182), summed in column order below
        double wgt = 0.0, term = 0.0;
        if (cn.w > 0) {
            const REAL ww = cn.w * cn.w;
            wgt = (double) ww;
            term = wgt * vft_piece<REAL, NC>(A, cn, cn, nullptr);
        }
        sW[p] = wgt;
        sT[p] = term;
        if (updateOut) {   // updateOutProfile, NJ.tcc:951-996 (as k_outprofile_update)
            REAL f[NC];
#pragma unroll
            for (int k = 0; k < NC; k++) f[k] = A.outF[p * NC + k];
            const REAL om = A.outW[p] * (REAL) nActiveOld;
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
            if (newOut) {   // the addends of (new node, out-profile) at this column, as vft_pair_block would form them
                Col<REAL, NC> co;
                co.w = wo;
                co.code = VFT_NOCODE_;
                co.vec = wo > 0;
#pragma unroll
                for (int k = 0; k < NC; k++) co.f[k] = f[k];
                double wgt2 = 0.0, term2 = 0.0;
                if (cn.w > 0 && co.w > 0) {
                    const REAL ww = cn.w * co.w;
                    wgt2 = (double) ww;
                    term2 = wgt2 * vft_piece<REAL, NC>(A, cn, co, A.outCD ? A.outCD + p * NC : nullptr);
                }
                sW2[p] = wgt2;
                sT2[p] = term2;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < (newOut ? 4 : 2)) {   // thread 0: top, thread 1: denom - chains in column order (as vft_pair_block); 2, 3: the same for newOut
        const double *src = threadIdx.x == 0 ? sT : threadIdx.x == 1 ? sW : threadIdx.x == 2 ? sT2 : sW2;
        double acc = 0;
        int64_t p = 0;
        for (; p + 8 <= nPos; p += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = src[p + u];
#pragma unroll
            for (int u = 0; u < 8; u++) acc += v[u];
        }
        for (; p < nPos; p++) acc += src[p];
        res[threadIdx.x] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double top = res[0], denom = res[1];
        A.selfweight[newn] = (REAL) (denom > 0 ? denom : 0.01);
        A.selfdist[newn] = (REAL) (denom > 0 ? top / denom : 1.0);
        if (newOut) {
            newOut[0] = res[2];
            newOut[1] = res[3];
        }
    }
}

template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_WG_PROF) void k_join_fused(Arena<REAL> A, int64_t i, int64_t j, int64_t newn, REAL diameter,
                                                           int32_t staleStamp, int64_t nActiveOld, int32_t updateOut, double tol,
                                                           REAL *stash, int64_t *pendIds, int32_t slot) {
    extern __shared__ __attribute__((aligned(16))) double jfLds[];
    vft_join_body<REAL, NC>(A, i, j, newn, diameter, staleStamp, nActiveOld, updateOut, tol, stash, pendIds, slot, jfLds);
}

// ---------------------------------------------------------------------------------------------------------------
// bit l of tileMask[t]: node 64 t + l is active (the out-profile kernels below walk all active nodes in ascending order)
static __global__ void k_tile_active_masks(const int32_t *parent, int64_t maxnode, unsigned long long *tileMask, int64_t nTiles) {
    const int64_t t = (int64_t) blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (t >= nTiles) return;
    const int64_t v = t * 64 + (threadIdx.x & 63);
    const bool act = v < maxnode && parent[v] < 0;
    const unsigned long long m = __ballot(act);
    if ((threadIdx.x & 63) == 0) tileMask[t] = m;
}

// ---------------------------------------------------------------------------------------------------------------
// outProfile over all active nodes (round 1 walked 64-node tiles with ~100 instructions per chain step on 13
// workgroups: 170 ms at 750 000 active nodes).  The accumulation is order-sensitive only where the
// addends differ, so it is split:
//   * nucleotide LEAVES without a distance matrix (ids below nSeqs come first in the ascending order): a leaf adds
//     exactly 1.0 to the frequency of its code and the constant 1/nActive to the weight.  The frequency sums are counts
//     (exact in numeric_t up to 2^24, where float addition of 1.0 saturates - reproduced), gathered by k_leaf_hist with
//     wave ballots over all leaf tiles in parallel; the weight chain is vft_iterate_add (closed form per binade).
//   * everything else (internal nodes; leaves when a matrix is set or the alphabet has 20 letters) runs the chain, but a
//     chain step is one LDS read and one add: per stage the whole workgroup expands VFT_OP_TILES tiles of 64 nodes x
//     COLS columns into dense addend rows in LDS (lane = node, coalesced tile-stream loads, inactive nodes compacted away
//     with the tile's active mask), prefetching the next stage's columns into registers while the (NC + 1) x COLS chain
//     threads consume the current one.  Adding the addends of a skipped record (+0.0) would also be exact, compaction
//     just saves the steps.
// Grid: nPosPad / COLS workgroups of 64 * COLS threads.
#define VFT_OP_TILES 4
template <int NC> struct OpCols { static const int value = NC == 4 ? 8 : 2; };

// per-(column, code) counts of the active leaves; hist[p * 4 + k]; grid (waves over tiles, nChunk), block 256
static __global__ __launch_bounds__(256) void k_leaf_hist(const uint4 *leafT, VftDims d, const unsigned long long *tileMask,
                                                          int64_t nLeafTiles, unsigned int *hist) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6), nWaves = (int64_t) gridDim.x * 4;
    const int chunk = blockIdx.y;
    unsigned int cnt[64];
#pragma unroll
    for (int u = 0; u < 64; u++) cnt[u] = 0;
    for (int64_t t = wave; t < nLeafTiles; t += nWaves) {
        unsigned long long m = tileMask[t];
        if (t * 64 + 63 >= d.nSeqs) m &= (1ull << (d.nSeqs - t * 64)) - 1ull;   // the tile that straddles nSeqs
        if (m == 0) continue;
        uint4 v = make_uint4(0, 0, 0, 0);   // gaps: no bits
        if ((m >> lane) & 1ull) v = leafT[vft_leaf_idx(d, t, chunk, lane)];
        const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int b = 0; b < 16; b++) {
            const uint32_t byte = (wv[b >> 2] >> (8 * (b & 3))) & 0xFFu;   // 0x10 | 1 << code, or 0 (vft_encode)
#pragma unroll
            for (int k = 0; k < 4; k++) cnt[b * 4 + k] += (unsigned int) __popcll(__ballot((byte >> k) & 1u));
        }
    }
    unsigned int mine = 0;
#pragma unroll
    for (int u = 0; u < 64; u++) mine = lane == u ? cnt[u] : mine;
    if (mine) atomicAdd(&hist[((int64_t) chunk * 16 + (lane >> 2)) * 4 + (lane & 3)], mine);
}

// workgroup: COLS loader waves (wave c = column c of the group, lane = node of the tile) + one wave of frequency chains
// + one wave of weight chains; the two kinds of chain have different step costs (a float add against convert - double
// add - convert) and would serialise inside one wave.
template <typename REAL> struct OpTiles { static const int value = sizeof(REAL) == 4 ? VFT_OP_TILES : VFT_OP_TILES / 2; };
template <int NC> struct OpThreads { static const int value = 64 * (OpCols<NC>::value + 2); };

template <typename REAL, int NC>
__global__ __launch_bounds__(OpThreads<NC>::value) void k_outprofile_chain(Arena<REAL> A, const unsigned long long *tileMask,
                                                                          int64_t firstTile, int64_t nTiles, int64_t nActive,
                                                                          double tol, const unsigned int *hist) {
    constexpr int COLS = OpCols<NC>::value, NCH = NC + 1, TILES = OpTiles<REAL>::value, CAP = TILES * 64, STRIDE = CAP + 4;
    // two addend buffers: the loaders expand stage s + 1 while the chains consume stage s (one barrier per stage)
    __shared__ __attribute__((aligned(16))) REAL sAdd[2][COLS * NCH][STRIDE];
    __shared__ int sTotal[2];
    __shared__ REAL sRes[COLS][NCH];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t p0 = (int64_t) blockIdx.x * COLS;
    const double inweight = 1.0 / (double) nActive;
    if (wv < COLS) {
        // ------------------------------------------------------------------------------------------ loaders
        const int c = wv;
        const int64_t p = p0 + c;
        const bool colOk = p < A.d.nPos;
        int64_t tNext = firstTile;
        unsigned long long mkA[TILES], mkB[TILES];
        Col<REAL, NC> colA[TILES], colB[TILES];
        auto fetch = [&](unsigned long long (&mk)[TILES], Col<REAL, NC> (&col)[TILES]) {
#pragma unroll
            for (int s = 0; s < TILES; s++) {
                unsigned long long m = 0;
                int64_t t = tNext;
                while (t < nTiles) {
                    m = tileMask[t];
                    if (hist && t * 64 < A.d.nSeqs) m &= ~((1ull << (A.d.nSeqs - t * 64)) - 1ull);   // leaves are in the histogram
                    if (m) break;
                    t++;
                }
                tNext = t + 1;
                mk[s] = t < nTiles ? m : 0ull;
                if (((mk[s] >> lane) & 1ull) && colOk) vft_load_col<REAL, NC>(A, t * 64 + lane, p, col[s]);
            }
        };
        // expands one stage into buf, publishes its size; false when the stage is empty (the end)
        auto expand = [&](unsigned long long (&mk)[TILES], Col<REAL, NC> (&col)[TILES], int buf) -> bool {
            int total = 0;
#pragma unroll
            for (int s = 0; s < TILES; s++) total += __popcll(mk[s]);
            if (tid == 0) sTotal[buf] = total;
            if (total == 0) return false;   // uniform
            int base = 0;
#pragma unroll
            for (int s = 0; s < TILES; s++) {
                if ((mk[s] >> lane) & 1ull) {
                    const int n = base + __popcll(mk[s] & ((1ull << lane) - 1ull));
                    const Col<REAL, NC> &cc = col[s];
                    const REAL w = colOk ? cc.w : (REAL) 0;
                    sAdd[buf][c * NCH + NC][n] = w;
#pragma unroll
                    for (int k = 0; k < NC; k++) {
                        REAL a = 0;
                        if (w > 0) {
                            if (cc.vec) a = cc.f[k] * w;                                        // vector_add_mult, NJ.tcc:825
                            else if (A.dmDist) a = A.dmCodeFreq[cc.code * NC + k] * w;          // NJ.tcc:828
                            else a = k == cc.code ? w : (REAL) 0;                               // NJ.tcc:831
                        }
                        sAdd[buf][c * NCH + k][n] = a;
                    }
                }
                base += __popcll(mk[s]);
            }
            return true;
        };
        fetch(mkA, colA);
        fetch(mkB, colB);
        for (int st = 0;; st += 2) {
            bool more = expand(mkA, colA, 0);
            if (more) fetch(mkA, colA);   // the stage after the next
            __syncthreads();
            if (!more) break;
            more = expand(mkB, colB, 1);
            if (more) fetch(mkB, colB);
            __syncthreads();
            if (!more) break;
        }
    } else {
        // ------------------------------------------------------------------------------------------ chains
        const bool weightWave = wv == COLS + 1;
        const int nChains = weightWave ? COLS : COLS * NC;
        const bool mine = lane < nChains;
        const int chCol = weightWave ? lane : lane / NC, chK = weightWave ? NC : lane % NC;
        REAL acc = 0;
        if (hist && mine && p0 + chCol < A.d.nPos) {   // the leaves' contribution (see above)
            const unsigned int *h = hist + (p0 + chCol) * 4;
            if (!weightWave) {
                const unsigned int n = h[chK];
                acc = sizeof(REAL) == 4 ? (REAL) (n < (1u << 24) ? n : (1u << 24)) : (REAL) n;
            } else {
                acc = vft_iterate_add<REAL>(inweight, (uint64_t) h[0] + h[1] + h[2] + h[3]);
            }
        }
        __syncthreads();   // stage 0 is in buffer 0
        for (int buf = 0;; buf ^= 1) {
            const int total = sTotal[buf];
            if (total == 0) break;   // uniform
            if (mine) {
                const REAL *src = sAdd[buf][chCol * NCH + chK];
                int n = 0;
                if (!weightWave) {
                    // reads run one group ahead of the adds (the adds are the dependent chain, the LDS latency is not)
                    REAL a0 = 0, a1 = 0, a2 = 0, a3 = 0;
                    if (total >= 4) {
                        a0 = src[0];
                        a1 = src[1];
                        a2 = src[2];
                        a3 = src[3];
                    }
                    for (; n + 8 <= total; n += 4) {
                        const REAL b0 = src[n + 4], b1 = src[n + 5], b2 = src[n + 6], b3 = src[n + 7];
                        acc = acc + a0;
                        acc = acc + a1;
                        acc = acc + a2;
                        acc = acc + a3;
                        a0 = b0;
                        a1 = b1;
                        a2 = b2;
                        a3 = b3;
                    }
                    if (n + 4 <= total) {
                        acc = acc + a0;
                        acc = acc + a1;
                        acc = acc + a2;
                        acc = acc + a3;
                        n += 4;
                    }
                    for (; n < total; n++) acc = acc + src[n];
                } else {
                    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
                    if (total >= 4) {
                        a0 = (double) src[0] * inweight;
                        a1 = (double) src[1] * inweight;
                        a2 = (double) src[2] * inweight;
                        a3 = (double) src[3] * inweight;
                    }
                    for (; n + 8 <= total; n += 4) {
                        const double b0 = (double) src[n + 4] * inweight, b1 = (double) src[n + 5] * inweight,
                                     b2 = (double) src[n + 6] * inweight, b3 = (double) src[n + 7] * inweight;
                        acc = (REAL) ((double) acc + a0);                                       // NJ.tcc:741
                        acc = (REAL) ((double) acc + a1);
                        acc = (REAL) ((double) acc + a2);
                        acc = (REAL) ((double) acc + a3);
                        a0 = b0;
                        a1 = b1;
                        a2 = b2;
                        a3 = b3;
                    }
                    if (n + 4 <= total) {
                        acc = (REAL) ((double) acc + a0);
                        acc = (REAL) ((double) acc + a1);
                        acc = (REAL) ((double) acc + a2);
                        acc = (REAL) ((double) acc + a3);
                        n += 4;
                    }
                    for (; n < total; n++) acc = (REAL) ((double) acc + (double) src[n] * inweight);
                }
            }
            __syncthreads();   // the loaders have the next stage in the other buffer (or its size 0)
        }
        if (mine) sRes[chCol][chK] = acc;
    }
    __syncthreads();
    if (tid < COLS && p0 + tid < A.d.nPos) {
        const int64_t pp = p0 + tid;
        REAL wo = sRes[tid][NC];
        if (wo <= 0) wo = (REAL) 1e-20;
        REAL f[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) f[k] = sRes[tid][k];
        vft_normalize_freq<REAL, NC>(A, f, tol);
        A.outW[pp] = wo;
#pragma unroll
        for (int k = 0; k < NC; k++) A.outF[pp * NC + k] = f[k];
        vft_out_codedist<REAL, NC>(A, pp, f);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Local-bootstrap support of the split (A,B)|(C,D) around an internal node (splitSupport, NJ.tcc:607-702): the six
// pairwise distances of the quartet are recomputed over nBoot column resamples (the same resamples for every node)
// and the split counts as supported when it beats both alternatives.  One workgroup per node: its threads first
// compute the per-column weights and weighted distance pieces of the six pairs into LDS, then every thread walks
// whole resamples (the sums are in sample order, in double, as in the reference).  colT is the resample table
// transposed, [nPos][nBoot], so that the threads of a workgroup read it coalesced.
// The decision `support1 > 0 && support2 > 0` goes through log() (logCorrect), and the device's log and glibc's differ
// in the last bit now and then: resamples whose margins are within `eps` of zero are NOT decided here but appended to
// `flagged` (node index + the six uncorrected distances) for the host to decide with its own libm, so that the counts
// equal the reference's exactly.  counts[k] = resamples decided "supported" on the device.
#define VFT_SUPPORT_WG 256
#define VFT_SUPPORT_REC 7   // doubles per flagged record
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_SUPPORT_WG) void k_split_support(Arena<REAL> A, const int64_t *nA, const int64_t *nB,
                                                                  const int64_t *nC, const int64_t *nD, int64_t n,
                                                                  const int32_t *colT, int32_t nBoot, int32_t scoredist,
                                                                  double eps, unsigned int *counts,
                                                                  unsigned int *nFlagged, double *flagged,
                                                                  unsigned int flagCap) {
    extern __shared__ __attribute__((aligned(16))) double spLds[];   // [6][nPos] pieces, then [6][nPos] weights
    __shared__ unsigned int nSupport;
    const int64_t k = blockIdx.x;
    if (k >= n) return;
    const int64_t nPos = A.d.nPos;
    double *sP = spLds, *sW = spLds + 6 * nPos;
    if (threadIdx.x == 0) nSupport = 0;
    const int64_t q[4] = {nA[k], nB[k], nC[k], nD[k]};
    for (int64_t p = threadIdx.x; p < nPos; p += blockDim.x) {
        Col<REAL, NC> c[4];
#pragma unroll
        for (int e = 0; e < 4; e++) vft_load_col_ml<REAL, NC>(A, q[e], p, c[e]);
        int j = 0;
#pragma unroll
        for (int x = 0; x < 4; x++)
#pragma unroll
            for (int y = x + 1; y < 4; y++, j++) {   // qAB, qAC, qAD, qBC, qBD, qCD
                const REAL ww = c[x].w * c[y].w;      // numeric_t product, NJ.tcc:624-629
                const double w = (double) ww;
                sW[j * nPos + p] = w;
                sP[j * nPos + p] = w * vft_piece<REAL, NC>(A, c[x], c[y], nullptr);
            }
    }
    __syncthreads();
    for (int32_t b = threadIdx.x; b < nBoot; b += blockDim.x) {
        double totp[6] = {0, 0, 0, 0, 0, 0}, totw[6] = {0, 0, 0, 0, 0, 0};
        for (int64_t i = 0; i < nPos; i++) {
            const int32_t col = colT[i * nBoot + b];
#pragma unroll
            for (int j = 0; j < 6; j++) {
                totp[j] += sP[j * nPos + col];
                totw[j] += sW[j * nPos + col];
            }
        }
        double raw[6], d[6];
#pragma unroll
        for (int j = 0; j < 6; j++) {
            double x = totw[j] > 0.01 ? totp[j] / totw[j] : 3.0;
            raw[j] = x;
            // logCorrect, NJ.tcc:322-330
            if (scoredist) x = x < 0.99 ? -1.3 * log(1.0 - x) : 3.0;
            else x = x < 0.74 ? -0.75 * log(1.0 - x * 4.0 / 3.0) : 3.0;
            d[j] = x < 3.0 ? x : 3.0;
        }
        const double support1 = d[1] + d[4] - d[0] - d[5];   // AC + BD - AB - CD
        const double support2 = d[2] + d[3] - d[0] - d[5];   // AD + BC - AB - CD
        if (fabs(support1) < eps || fabs(support2) < eps) {
            const unsigned int slot = atomicAdd(nFlagged, 1u);
            if (slot < flagCap) {
                double *rec = flagged + (size_t) slot * VFT_SUPPORT_REC;
                rec[0] = (double) k;
#pragma unroll
                for (int j = 0; j < 6; j++) rec[1 + j] = raw[j];
            }
        } else if (support1 > 0 && support2 > 0) {
            atomicAdd(&nSupport, 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) counts[k] = nSupport;
}
