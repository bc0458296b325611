// One-vs-all sweeps for 20-state alphabets with a distance matrix (amino acids: BASELINE config C5, the NJ and
// minimum-evolution phases of every protein run).  setBestHit (NJ.tcc:3571-3646) / setOutDistance (NJ.tcc:1012-1083)
// over profileDist with the matrix branch of profileDistPiece (NJ.tcc:900-918) and seqDist's matrix branch
// (NJ.tcc:1614-1620).
//
// The nucleotide kernels of vft_kernels_nj.h are lane-per-target walks; for proteins that shape starves the chip: the
// target lists are short (C5: 50 000 leaves, a few ten thousand internal nodes), a vector is 20 values (160 bytes in
// double) and every column is a dependent mask -> offset -> stream round trip.  Here:
//   * k_aa_query_prep turns the query (a node, or the out-profile for MODE_OUTDIST) into three small tables:
//       wq[p]          its weight,
//       qvec[p][20]    the vector the reference would use on its side of the 3-way product (its own, or
//                      codeFreq[code]),
//       ptab[p][c]     profileDistPiece against a target that holds CODE c at column p: distances[cq][c], the
//                      codeDist entry of the out-profile, or vector_multiply3_sum(fq, codeFreq[c], eigenval) - a
//                      plain-code target column then costs one LDS read instead of a 20-term product;
//   * leaf targets (codes only): a wavefront per tile of 64 leaves, lane per leaf, 16 columns per 16-byte load, the
//     next chunk in flight while the current one is walked; per column one ptab read and two double adds, in column
//     order (the reference's sequence of additions);
//   * internal targets: a WORKGROUP of 16 wavefronts per tile of 64 nodes.  Wavefront w owns column w of every 16-column
//     chunk (the next chunk's loads in flight while this one is computed); each
//     computes (wgt, wgt * piece) for its column x 64 targets - vectors come straight from the tile's contiguous
//     stream (vft_layout.h), the 20-term products in numeric_t with the reference's four strided accumulators - and
//     parks them in LDS; wavefront 0 then adds each target's 16 addends in column order.  64 ordered sums advance in
//     parallel, nothing is ever reduced across lanes, results are bit-identical to the generic kernels and the CPU.
// Algorithmic bytes: leaf nPos + 2S, internal nPos (S + 1) + nvec 20 S + 2S, + S + 8 for the criterion.
#pragma once
#include "vft_kernels_nj.h"

#define VFT_AA_NC 20
#define VFT_AA_WG 1024

template <typename REAL>
struct AaQuery {
    const REAL *wq;     // [nPosPad]
    const REAL *qvec;   // [nPosPad][20]
    const REAL *ptab;   // [nPosPad][20]
};

// query < 0: the out-profile (its codeDist rows are the code table); otherwise node `query`
template <typename REAL>
__global__ void k_aa_query_prep(Arena<REAL> A, int64_t query, REAL *wq, REAL *qvec, REAL *ptab) {
    constexpr int NC = VFT_AA_NC;
    const int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nPosPad = (int64_t) A.d.nChunk * VFT_CHUNK;
    const int64_t p = t / NC;
    const int c = (int) (t % NC);
    if (p >= nPosPad) return;
    REAL w = 0, f[NC];
    int code = VFT_NOCODE_;
    bool vec = false;
#pragma unroll
    for (int k = 0; k < NC; k++) f[k] = 0;
    if (p < A.d.nPos) {
        if (query < 0) {
            w = A.outW[p];
            vec = w > 0;
#pragma unroll
            for (int k = 0; k < NC; k++) f[k] = A.outF[p * NC + k];
        } else {
            Col<REAL, NC> col;
            vft_load_col<REAL, NC>(A, query, p, col);
            w = col.w;
            code = col.code;
            vec = col.vec;
            if (vec) {
#pragma unroll
                for (int k = 0; k < NC; k++) f[k] = col.f[k];
            } else if (code != VFT_NOCODE_) {
#pragma unroll
                for (int k = 0; k < NC; k++) f[k] = A.dmCodeFreq[code * NC + k];
            }
        }
    }
    const bool live = w > 0 && (vec || code != VFT_NOCODE_);
    REAL piece = 0;
    if (live) {
        if (query < 0 && A.outCD) piece = A.outCD[p * NC + c];                       // codeDist2[code1], NJ.tcc:905
        else if (!vec) piece = A.dmDist[code * NC + c];                              // distances[code1][code2], :903
        else {                                                                       // f1 vs codeFreq[code2], :907-916
            REAL cf[NC], ev[NC];
#pragma unroll
            for (int k = 0; k < NC; k++) {
                cf[k] = A.dmCodeFreq[c * NC + k];
                ev[k] = A.dmEigenval[k];
            }
            piece = vft_red4_mul3<REAL, NC>(f, cf, ev);
        }
    }
    ptab[p * NC + c] = piece;
    qvec[p * NC + c] = live ? f[c] : (REAL) 0;
    if (c == 0) wq[p] = live ? w : (REAL) 0;
}

// epilogue shared by both target kinds: MODE_OUTDIST refreshes the node's out-distance, otherwise distance, weight and
// criterion of the (query, target) pair go to the sweep arrays
template <typename REAL, int MODE>
__device__ __forceinline__ void vft_aa_finish(const Arena<REAL> &A, const SweepArgs &s, const SweepOut<REAL> &O, int64_t j,
                                              double top, double denom, bool leafPair, REAL &cmin, REAL &cmax) {
    REAL dist = (REAL) (denom > 0 ? top / denom : 1.0);
    REAL weight = leafPair ? (REAL) denom : (REAL) (denom > 0 ? denom : 0.01);   // seqDist: weight = nUse (NJ.tcc:1622)
    if (MODE == MODE_OUTDIST) {
        const REAL od = vft_out_distance<REAL>(dist, weight, s.nActive, A.selfweight[j], A.selfdist[j], A.diameter[j], s.totdiam);
        A.outDist[j] = od;
        A.nOutActive[j] = (int32_t) s.nActive;
        A.mOutDist[j] = od;
        A.mNOut[j] = (int32_t) s.nActive;
        return;
    }
    if (!leafPair) {
        const REAL dd = A.diameter[s.query] + A.diameter[j];
        dist = dist - dd;
    }
    const REAL crit = vft_criterion<REAL>(dist, A.outDist[s.query], A.nOutActive[s.query], A.outDist[j], A.nOutActive[j], s.nActive);
    O.dist[j] = dist;
    O.weight[j] = weight;
    O.crit[j] = crit;
    cmin = crit < cmin ? crit : cmin;
    cmax = crit > cmax ? crit : cmax;
}

// does target j take part?  (MODE_OUTDIST: only active nodes staler than allowed, NJ.tcc:1092-1098, 1013-1015)
template <typename REAL, int MODE>
__device__ __forceinline__ bool vft_aa_wanted(const Arena<REAL> &A, const SweepArgs &s, const SweepOut<REAL> &O, int64_t j) {
    if (j >= s.hi || j < s.lo) return false;
    if (A.parent[j] >= 0) {
        if (MODE != MODE_OUTDIST) {
            O.dist[j] = (REAL) 1e20;
            O.crit[j] = (REAL) 1e20;
            O.weight[j] = 0;
        }
        return false;
    }
    if (MODE == MODE_OUTDIST) {
        if (!s.force && !((int64_t) A.nOutActive[j] - s.nActive > s.nDiffAllow)) return false;
        if ((int64_t) A.nOutActive[j] == s.nActive) return false;
    }
    return true;
}

// One column of one internal tile for the 64 lanes of a wavefront: what the reference's profileDist adds to
// (denom, top) at column p for target lane `lane` - loads are issued by vft_aa_col_load, consumed by vft_aa_col_addends,
// so that a caller can keep the next column's loads in flight while it works on this one.
template <typename REAL>
struct AaCol {
    REAL f2[VFT_AA_NC];
    REAL w2;
    uint32_t ct;
    bool hv;
};

template <typename REAL>
__device__ __forceinline__ void vft_aa_col_load(AaCol<REAL> &r, const uint4 *cT, vft_smask_t mM, vft_soff_t mO, const REAL *wT,
                                                const REAL *fT, int64_t p, int lane) {
    constexpr int NC = VFT_AA_NC;
    const vft_u4_t mk = mM[p];   // wave-uniform: scalar loads
    const vft_u2_t of = mO[p];
    const unsigned long long mv = ((unsigned long long) mk.y << 32) | mk.x;
    const unsigned long long mw = ((unsigned long long) mk.w << 32) | mk.z;
    const bool hv = __builtin_amdgcn_inverse_ballot_w64(mv), hw = __builtin_amdgcn_inverse_ballot_w64(mw);
    const uint32_t slotV = __builtin_amdgcn_mbcnt_hi(mk.y, __builtin_amdgcn_mbcnt_lo(mk.x, of.x));
    const uint32_t slotW = __builtin_amdgcn_mbcnt_hi(mk.w, __builtin_amdgcn_mbcnt_lo(mk.z, of.y));
    const uint4 codes = cT[(p >> 4) * VFT_TILE];
    r.ct = vft_byte(codes, (int) (p & 15));
    r.hv = hv;
    r.w2 = (hv || r.ct != VFT_NOCODE_) ? (REAL) 1 : (REAL) 0;   // implied weight (vft_layout.h)
    if (hw) r.w2 = wT[slotW];
    if (hv) {   // transposed block (vft_fidx): consecutive lanes read consecutive 16 bytes
        const int nvec = __popcll(mv);
        const int rank = (int) (slotV - of.x);
#pragma unroll
        for (int k = 0; k < NC; k++) r.f2[k] = fT[vft_fidx<REAL, NC>(of.x, nvec, rank, k)];
    }
}

template <typename REAL>
__device__ __forceinline__ double2 vft_aa_col_addends(const AaCol<REAL> &r, const AaQuery<REAL> &Q, const REAL *ptab,
                                                      const REAL *eigenval, int64_t p) {
    constexpr int NC = VFT_AA_NC;
    const REAL wq1 = vft_uniform_load<REAL>(Q.wq + p);
    double piece = 0.0;
    if (r.hv) {
        REAL qv[NC], ev[NC];
#pragma unroll
        for (int k = 0; k < NC; k += 4) {
            const typename UVec4<REAL>::type v = vft_uniform_load4<REAL>(Q.qvec + p * NC + k);
            const typename UVec4<REAL>::type e = vft_uniform_load4<REAL>(eigenval + k);
            qv[k] = v.x; qv[k + 1] = v.y; qv[k + 2] = v.z; qv[k + 3] = v.w;
            ev[k] = e.x; ev[k + 1] = e.y; ev[k + 2] = e.z; ev[k + 3] = e.w;
        }
        piece = (double) vft_red4_mul3<REAL, NC>(qv, r.f2, ev);
    } else if (r.ct != VFT_NOCODE_) {
        piece = (double) ptab[p * NC + r.ct];
    }
    double wgt = 0.0, term = 0.0;
    if (wq1 > 0 && r.w2 > 0) {   // NJ.tcc:1175-1182
        const REAL ww = wq1 * r.w2;
        wgt = (double) ww;
        term = wgt * piece;
    }
    return make_double2(wgt, term);
}

// grid: [0, nLeafWG) leaf workgroups of 16 tiles each (a wavefront per tile), then one workgroup per internal tile
// starting at tile `intTile0` (a wavefront per column of the 16-column chunk).
// dynamic LDS: ptab [nPosPad][20] REAL | wq [nPosPad] double | stage [2][16][64] double2 (internal workgroups)
template <typename REAL, int MODE>
__global__ __launch_bounds__(VFT_AA_WG) void k_sweep_aa(Arena<REAL> A, AaQuery<REAL> Q, SweepArgs s, SweepOut<REAL> O,
                                                       int64_t leafTile0, int32_t nLeafWG, int64_t intTile0) {
    constexpr int NC = VFT_AA_NC;
    extern __shared__ __attribute__((aligned(16))) unsigned char aaLds[];
    const int64_t nPosPad = (int64_t) A.d.nChunk * VFT_CHUNK;
    REAL *ptab = (REAL *) aaLds;
    double *wqD = (double *) (aaLds + (((size_t) nPosPad * NC * sizeof(REAL) + 15) & ~(size_t) 15));
    double2 *stage = (double2 *) (wqD + nPosPad);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int64_t t = tid; t < nPosPad * NC; t += VFT_AA_WG) ptab[t] = Q.ptab[t];
    for (int64_t t = tid; t < nPosPad; t += VFT_AA_WG) wqD[t] = (double) Q.wq[t];
    __syncthreads();
    REAL cmin = (REAL) 1e30, cmax = (REAL) -1e30;
    const int nChunk = A.d.nChunk;
    if ((int) blockIdx.x < nLeafWG) {
        // ---------------------------------------------------------------- leaves: a wavefront per tile, lane per leaf
        const int64_t tile = leafTile0 + (int64_t) blockIdx.x * (VFT_AA_WG / 64) + wave;
        const int64_t j = tile * VFT_TILE + lane;
        const bool want = j < A.d.nSeqs && vft_aa_wanted<REAL, MODE>(A, s, O, j);
        if (tile * VFT_TILE < A.d.nSeqs) {
            const bool leafPair = MODE != MODE_OUTDIST && s.query >= 0 && s.query < A.d.nSeqs;
            const uint4 *cT = A.leafT + vft_leaf_idx(A.d, tile, 0, lane);
            double top = 0, denom = 0;
            uint4 cur = cT[0];
            for (int c = 0; c < nChunk; c++) {
                const uint4 nxt = cT[(int64_t) (c + 1 < nChunk ? c + 1 : c) * VFT_TILE];
#pragma unroll
                for (int b = 0; b < VFT_CHUNK; b++) {
                    const int64_t p = (int64_t) c * VFT_CHUNK + b;
                    const uint32_t ct = vft_byte(cur, b);
                    const bool has = ct != VFT_NOCODE_;
                    const double wgt = has ? wqD[p] : 0.0;
                    const double piece = (double) ptab[p * NC + (has ? ct : 0u)];
                    denom += wgt;
                    top += wgt * piece;
                }
                cur = nxt;
            }
            if (want) vft_aa_finish<REAL, MODE>(A, s, O, j, top, denom, leafPair, cmin, cmax);
        }
    } else {
        // ---------------------------------------------------------------- internal nodes: a workgroup per tile
        const int64_t tile = intTile0 + ((int64_t) blockIdx.x - nLeafWG);
        const int64_t j = tile * VFT_TILE + lane;
        const int64_t pt = tile - A.d.firstProfTile;
        const bool mine = j >= A.d.nSeqs;   // the tile that straddles nSeqs also holds leaves (handled above)
        const bool want = wave == 0 && mine && vft_aa_wanted<REAL, MODE>(A, s, O, j);
        // a tile without a single wanted target skips the walk (workgroup-uniform decision)
        __shared__ int anyWanted;
        if (tid == 0) anyWanted = 0;
        __syncthreads();
        if (want) anyWanted = 1;
        __syncthreads();
        if (anyWanted) {
            const uint4 *cT = A.profC + vft_c_idx(A.d, pt, 0, lane);
            const vft_smask_t mM = (vft_smask_t) (A.colMask + vft_meta_idx(A.d, pt, 0));
            const vft_soff_t mO = (vft_soff_t) (A.colOff + vft_meta_idx(A.d, pt, 0));
            const REAL *wT = A.profW + vft_wstream_base(A.d, pt);
            const REAL *fT = A.profF + vft_fstream_base(A.d, pt);
            // wavefront `wave` owns column `wave` of every 16-column chunk; the loads of the next chunk's column are in
            // flight while this chunk's addends are computed, staged and summed
            double top = 0, denom = 0;
            AaCol<REAL> ca, cb;   // ping-pong register sets: one chunk's column in use, the next one's loads in flight
            vft_aa_col_load<REAL>(ca, cT, mM, mO, wT, fT, (int64_t) wave, lane);
            // staging is double-buffered: chunk c is summed by wavefront 0 while the others already stage chunk c + 1, so
            // ONE barrier per chunk orders both the hand-over and the reuse of a buffer two chunks later
            auto round = [&](AaCol<REAL> &use, AaCol<REAL> &fill, int c) {
                const int64_t p = (int64_t) c * VFT_CHUNK + wave;
                if (c + 1 < nChunk) vft_aa_col_load<REAL>(fill, cT, mM, mO, wT, fT, p + VFT_CHUNK, lane);
                double2 *buf = stage + (c & 1) * (VFT_CHUNK * VFT_TILE);
                buf[wave * VFT_TILE + lane] = vft_aa_col_addends<REAL>(use, Q, ptab, A.dmEigenval, p);
                __syncthreads();
                if (wave == 0) {
#pragma unroll
                    for (int b = 0; b < VFT_CHUNK; b++) {
                        const double2 v = buf[b * VFT_TILE + lane];
                        denom += v.x;
                        top += v.y;
                    }
                }
            };
            for (int c = 0; c < nChunk; c += 2) {
                round(ca, cb, c);
                if (c + 1 < nChunk) round(cb, ca, c + 1);
            }
            if (want) vft_aa_finish<REAL, MODE>(A, s, O, j, top, denom, false, cmin, cmax);
        }
    }
    if (MODE != MODE_OUTDIST) vft_block_minmax_n<REAL, VFT_AA_WG / 64>(cmin, cmax, O.partMin, O.partMax, (int) blockIdx.x);
}
