// NJ-phase kernels: one-vs-all sweeps (setBestHit), pair lists (transferBestHits & co), out-distances,
// average / out-profile maintenance and the top-k selection that replaces the reference's full sort.
//
// Parity rule for every distance: ONE lane owns ONE (query,target) pair and walks the alignment columns in
// order, accumulating `top` and `denom` in double exactly as profileDist does (NJ.tcc:1167-1190).  There is no
// cross-lane reduction, so results are bit-identical to the CPU and independent of launch geometry; the
// tile-transposed layout (vft_layout.h) is what makes that access pattern fully coalesced.
#pragma once
#include "vft_device.h"

#define VFT_WG 256

// ------------------------------------------------------------------------------------------------ query staging
// Row-major copy of the query profile, read wave-uniformly (scalar loads) by the sweep kernels.
template <typename REAL>
struct QueryBuf {
    REAL *w;        // [nPosPad]
    uint8_t *code;  // [nPosPad] reference codes
    REAL *f;        // [nPosPad][nCodes]; nt: one-hot for code columns (see k_extract_query)
    uint4 *enc;     // [nChunk] encoded leaf bytes (valid when the query is a leaf)
    double2 *tab;   // [nPosPad][5] nt: what a LEAF target adds to (top, denom) at this column, by its code 0..3 / gap
};

// A leaf column is a code with weight 1, so against a profile query its contribution depends only on (column, code):
//     code c:  denom += (double) wq,   top += (double) wq * (1.0 - (double) fq[c])     (NJ.tcc:1176-1182, 922-930)
//     gap (or wq <= 0):  +0.0 to both, which is exact
template <typename REAL>
__device__ __forceinline__ void vft_query_tab(const QueryBuf<REAL> &q, int64_t p, REAL wq, const REAL *f) {
#pragma unroll
    for (int c = 0; c < 5; c++) {
        double2 v = make_double2(0.0, 0.0);
        if (c < 4 && wq > 0) {
            const double wgt = (double) wq;
            const double piece = 1.0 - (double) f[c < 4 ? c : 0];
            v = make_double2(wgt * piece, wgt);
        }
        q.tab[p * 5 + c] = v;
    }
}

template <typename REAL, int NC>
__device__ __forceinline__ void vft_extract_query(const Arena<REAL> &A, int64_t node, const QueryBuf<REAL> &q) {
    const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nPosPad = (int64_t) A.d.nChunk * VFT_CHUNK;
    if (p >= nPosPad) return;
    REAL w = 0;
    int code = VFT_NOCODE_;
    REAL f[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) f[k] = 0;
    if (p < A.d.nPos) {
        Col<REAL, NC> c;
        vft_load_col<REAL, NC>(A, node, p, c);
        w = c.w;
        code = c.code;
        if (c.vec) {
#pragma unroll
            for (int k = 0; k < NC; k++) f[k] = c.f[k];
        } else if (NC == 4 && code != VFT_NOCODE_) {
#pragma unroll
            for (int k = 0; k < NC; k++) f[k] = (k == code) ? (REAL) 1 : (REAL) 0;
        }
    }
    q.w[p] = w;
    q.code[p] = (uint8_t) code;
#pragma unroll
    for (int k = 0; k < NC; k++) q.f[p * NC + k] = f[k];
    if (NC == 4) vft_query_tab<REAL>(q, p, w, f);
    if (node < A.d.nSeqs && p < A.d.nChunk) {
        q.enc[p] = A.leafT[vft_leaf_idx(A.d, node >> 6, (int) p, (int) (node & 63))];
    }
}
template <typename REAL, int NC>
__global__ void k_extract_query(Arena<REAL> A, int64_t node, QueryBuf<REAL> q) {
    vft_extract_query<REAL, NC>(A, node, q);
}
// the queries of a batch of seeds in ONE launch (vft_sweep_batch): blockIdx.y = seed, each into its own staging buffers
// mq (optional): the seed is one of mqS profile seeds that share a pass over the targets (k_sweep_nt_profq_multi) - its weights and
// vectors also go, as query number mqIdx, into the group's INTERLEAVED buffer: per column VFT_MQ_STRIDE(S) = 4 S + 4 numbers,
// [f of query 0 (4)] ... [f of query S-1 (4)] [w of queries 0..S-1, padded to 4] - one base pointer and two or three scalar loads
// per column for all S queries, where S separate buffers took 2 S pointers (SGPR pairs the kernel has no room for) and 2 S loads.
#define VFT_MQ_STRIDE(S) (4 * (S) + 4)
template <typename REAL>
struct QuerySlot {
    int64_t node;
    QueryBuf<REAL> q;
    REAL *mq;
    int32_t mqIdx, mqS;
};
template <typename REAL, int NC>
__global__ void k_extract_query_batch(Arena<REAL> A, const QuerySlot<REAL> *qs) {
    const QuerySlot<REAL> &sl = qs[blockIdx.y];
    vft_extract_query<REAL, NC>(A, sl.node, sl.q);
    if (NC == 4 && sl.mq) {   // (the thread re-reads what it has just written: its own column)
        const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
        if (p < (int64_t) A.d.nChunk * VFT_CHUNK) {
            REAL *dst = sl.mq + p * VFT_MQ_STRIDE(sl.mqS);
#pragma unroll
            for (int k = 0; k < 4; k++) dst[4 * sl.mqIdx + k] = sl.q.f[p * 4 + k];
            dst[4 * sl.mqS + sl.mqIdx] = sl.q.w[p];
        }
    }
}

// The out-profile as a query (every column NOCODE with a vector, NJ.tcc:743-747).
template <typename REAL, int NC>
__global__ void k_outprofile_as_query(Arena<REAL> A, QueryBuf<REAL> q) {
    const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nPosPad = (int64_t) A.d.nChunk * VFT_CHUNK;
    if (p >= nPosPad) return;
    const bool in = p < A.d.nPos;
    q.w[p] = in ? A.outW[p] : (REAL) 0;
    q.code[p] = VFT_NOCODE_;
    REAL f[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) {
        f[k] = in ? A.outF[p * NC + k] : (REAL) 0;
        q.f[p * NC + k] = f[k];
    }
    if (NC == 4) vft_query_tab<REAL>(q, p, in ? A.outW[p] : (REAL) 0, f);
}

// ------------------------------------------------------------------------------------------------ sweep (nt)
struct SweepArgs {
    int64_t query;       // node id, or -1 when the query is the out-profile (MODE_OUTDIST)
    int64_t lo, hi;      // target id range [lo, hi), lo % 64 == 0
    int64_t nActive, nDiffAllow;
    double totdiam;
    int32_t queryIsLeaf;
    int32_t force;       // MODE_OUTDIST: refresh every listed node regardless of staleness
    int32_t nLeafWG;     // workgroups of k_sweep_nt_table (VFT_LEAF_SPAN leaves each); 0 when the query is a leaf
    int32_t pad;
    int64_t leafEnd;     // k_sweep_nt_table covers the ids [lo, leafEnd)
    int64_t heavyLo;     // k_sweep_nt covers [heavyLo, hi) (a multiple of 64) except the ids below leafEnd
};

template <typename REAL>
struct SweepOut {
    REAL *dist, *weight, *crit;   // [maxNodes], indexed by target id; inactive targets hold the 1e20 sentinel
    REAL *partMin, *partMax;      // [gridDim.x] per-workgroup min / max criterion of the active targets, reduced
                                  // by k_select_hist: the sweep issues no global atomics at all
};

// per-workgroup (min,max) of the criteria produced by this launch; every thread of the workgroup must call it
template <typename REAL, int NWAVES>
__device__ __forceinline__ void vft_block_minmax_n(REAL cmin, REAL cmax, REAL *partMin, REAL *partMax, int part) {
    __shared__ REAL smin[NWAVES], smax[NWAVES];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const REAL a = __shfl_xor(cmin, off, 64), b = __shfl_xor(cmax, off, 64);
        cmin = a < cmin ? a : cmin;
        cmax = b > cmax ? b : cmax;
    }
    if ((threadIdx.x & 63) == 0) {
        smin[threadIdx.x >> 6] = cmin;
        smax[threadIdx.x >> 6] = cmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < NWAVES; w++) {
            cmin = smin[w] < cmin ? smin[w] : cmin;
            cmax = smax[w] > cmax ? smax[w] : cmax;
        }
        partMin[part] = cmin;
        partMax[part] = cmax;
    }
}
template <typename REAL>
__device__ __forceinline__ void vft_block_minmax(REAL cmin, REAL cmax, REAL *partMin, REAL *partMax, int part) {
    vft_block_minmax_n<REAL, VFT_WG / 64>(cmin, cmax, partMin, partMax, part);
}

// MODE_CRIT_LEAFQ is MODE_CRIT with the knowledge that the query is a leaf (every seed of setAllLeafTopHits): its
// columns are plain codes with weight 1, so against a profile column only ONE frequency matters (NJ.tcc:922-930)
enum { MODE_CRIT = 0, MODE_OUTDIST = 1, MODE_CRIT_LEAFQ = 2 };

// %-different distance of two leaves from their encoded bytes (seqDist, NJ.tcc:1601-1612): integer counts.
__device__ __forceinline__ void vft_seq_counts(const uint4 a, const uint4 b, int &nUse, int &nSame) {
    const uint32_t x = a.x & b.x, y = a.y & b.y, z = a.z & b.z, w = a.w & b.w;
    nUse += __popc(x & 0x10101010u) + __popc(y & 0x10101010u) + __popc(z & 0x10101010u) + __popc(w & 0x10101010u);
    nSame += __popc(x & 0x0F0F0F0Fu) + __popc(y & 0x0F0F0F0Fu) + __popc(z & 0x0F0F0F0Fu) + __popc(w & 0x0F0F0F0Fu);
}

// Nucleotide, no distance matrix (the -nt default).  NC == 4.
// ---- internal targets: one 16-column chunk of one tile, as registers of the lane that owns the target.
// Every column is brought to the "both sides hold a vector" form of profileDistPiece (NJ.tcc:933-937):
//     piece = 1 - f1[0]*f2[0] - f1[1]*f2[1] - f1[2]*f2[2] - f1[3]*f2[3]   (numeric_t products, double subtractions)
// with a plain code c replaced by its one-hot vector.  That is bit-identical to the reference's special cases
// (c1==c2 ? 0 : 1,  1 - f2[c1],  1 - f1[c2]; NJ.tcc:920-930): a product with 0 or 1 is exact and subtracting the
// resulting +-0.0 from the running double never changes it.  One code path instead of four, no per-lane selects.
#ifndef VFT_SUB
#define VFT_SUB 8   // columns loaded and consumed together (a divisor of VFT_CHUNK)
#endif
// code of query column p0 + b (p0 a multiple of 4, b a constant): scalar dword load + scalar bit-field extract
template <typename REAL>
__device__ __forceinline__ uint32_t vft_query_code(const QueryBuf<REAL> &Q, int64_t p0, int b) {
    const uint32_t w = vft_uniform_load<uint32_t>((const uint32_t *) (Q.code + p0 + (b & ~3)));
    return (w >> ((b & 3) * 8)) & 0xFFu;
}

// QLEAF (the query is a leaf, code cq per column): piece = 1 - f2[cq] (NJ.tcc:924) and f2 of a plain target code is
// (cd == cq), so one value per column is enough.
template <typename REAL, bool QLEAF>
struct IntChunk {
    REAL w[VFT_SUB];
    REAL f[VFT_SUB][QLEAF ? 1 : 4];
};

// the per-column metadata of a tile is wave-uniform: read through the scalar cache as (4 + 2)-dword loads
typedef unsigned int __attribute__((ext_vector_type(4))) vft_u4_t;
typedef unsigned int __attribute__((ext_vector_type(2))) vft_u2_t;
typedef const __attribute__((address_space(4))) vft_u4_t *vft_smask_t;   // ColMask {vec lo, vec hi, w lo, w hi}
typedef const __attribute__((address_space(4))) vft_u2_t *vft_soff_t;    // ColOff {vec, w}

// Kernel ablation switches (tools/ablate_sweep.py) exist only in builds made with -DVFT_ABLATE; in the product build
// the bits are a compile-time zero and every ablation branch folds away.
#ifdef VFT_ABLATE
#define VFT_ABLATE_BITS(s) ((s).pad)
#else
#define VFT_ABLATE_BITS(s) 0
#endif

// codes: the target's 16 codes of this chunk (loaded one chunk ahead so that the defaults below do not wait on it)
// wT / fT: the tile's explicit-weight / vector streams (vft_layout.h)
template <typename REAL, bool QLEAF>
__device__ __forceinline__ void vft_int_chunk_load(IntChunk<REAL, QLEAF> &r, int c, int sub, const uint4 codes,
                                                   const REAL *wT, vft_smask_t mM, vft_soff_t mO, const REAL *fT,
                                                   const QueryBuf<REAL> &Q, int dbg = 0) {
    const int64_t p0 = (int64_t) c * VFT_CHUNK + sub * VFT_SUB;
    vft_u4_t mk[VFT_SUB];
    vft_u2_t of[VFT_SUB];
#pragma unroll
    for (int b = 0; b < VFT_SUB; b++) {   // wave-uniform: scalar loads
        mk[b] = mM[p0 + b];
        of[b] = mO[p0 + b];
    }
#pragma unroll
    for (int b = 0; b < VFT_SUB; b++) {
        const unsigned long long mv = ((unsigned long long) mk[b].y << 32) | mk[b].x;
        const unsigned long long mw = ((unsigned long long) mk[b].w << 32) | mk[b].z;
        // the masks are wave-uniform SGPR pairs: inverse_ballot turns them into the lane predicate for free, and
        // mbcnt starts counting at the column's offset into the stream
        const bool hv = __builtin_amdgcn_inverse_ballot_w64(mv) && !(dbg & 1);
        const bool hw = __builtin_amdgcn_inverse_ballot_w64(mw) && !(dbg & 2);
        const uint32_t slotV = __builtin_amdgcn_mbcnt_hi(mk[b].y, __builtin_amdgcn_mbcnt_lo(mk[b].x, of[b].x));
        const uint32_t slotW = __builtin_amdgcn_mbcnt_hi(mk[b].w, __builtin_amdgcn_mbcnt_lo(mk[b].z, of[b].y));
        const uint32_t cd = vft_byte(codes, sub * VFT_SUB + b);
        r.w[b] = (hv || cd != VFT_NOCODE_) ? (REAL) 1 : (REAL) 0;   // implicit weight (vft_layout.h)
        // 32-bit byte offsets into the tile's streams (a tile's stream is < 4 GB: vft_create checks nPos): the loads
        // take the SGPR base + VGPR offset form instead of 64-bit per-lane address arithmetic
        const char *wB = (const char *) wT, *fB = (const char *) fT;
        if (hw) r.w[b] = *(const REAL *) (wB + slotW * (uint32_t) sizeof(REAL));
        if (QLEAF) {
            // wave-uniform query code; a gap (127) is masked out by its weight 0, any in-range index will do
            const uint32_t cq = vft_query_code<REAL>(Q, p0, b);
            r.f[b][0] = (cd == cq) ? (REAL) 1 : (REAL) 0;
            if (hv) r.f[b][0] = *(const REAL *) (fB + (slotV * 4u + (cq & 3u)) * (uint32_t) sizeof(REAL));
        } else {
            // one-hot of a plain code without compares: byte k of `oh` is (code == k), read by v_cvt_f32_ubyteK.
            // (a gap, 127, aliases code 3 here: harmless, its lane has weight 0 and any finite piece gives +0.0)
            uint32_t oh;   // (asm: keeps the compiler from turning the byte reads back into compare + select)
            asm("v_lshlrev_b32 %0, %1, 1" : "=v"(oh) : "v"((cd & 3u) * 8u));
#pragma unroll
            for (int k = 0; k < 4; k++) r.f[b][QLEAF ? 0 : k] = (REAL) ((oh >> (8 * k)) & 0xFFu);
            if (hv) {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    r.f[b][QLEAF ? 0 : k] = *(const REAL *) (fB + (slotV * 4u + (uint32_t) k) * (uint32_t) sizeof(REAL));
            }
        }
    }
}

// The reference's column loop (NJ.tcc:1172-1183) over the 16 columns, in order and branch-free.  The reference skips
// a column when either weight is <= 0; weights are never negative, so the product is +0.0 there and adding
// (+0.0, +0.0 * piece) to the two double sums is exact.  Padding columns (beyond nPos) have weight 0 on both sides.
template <typename REAL, bool QLEAF>
__device__ __forceinline__ void vft_int_chunk_consume(const IntChunk<REAL, QLEAF> &r, int c, int sub,
                                                      const QueryBuf<REAL> &Q, double &top, double &denom, int dbg = 0) {
    const int64_t p0 = (int64_t) c * VFT_CHUNK + sub * VFT_SUB;
    if (dbg & 4) {   // ablation: no arithmetic, just consume the loaded values
#pragma unroll
        for (int b = 0; b < VFT_SUB; b++) top += (double) r.w[b] + (double) r.f[b][0];
        return;
    }
#pragma unroll
    for (int b = 0; b < VFT_SUB; b++) {
        const int64_t p = p0 + b;
        if (QLEAF) {
            // the leaf's weight is 1 (0 at a gap): w1 * w2 is w2 (or 0) exactly (NJ.tcc:1176)
            const uint32_t cq = vft_query_code<REAL>(Q, p0, b);
            const double wgt = (double) (cq != VFT_NOCODE_ ? r.w[b] : (REAL) 0);
            const double piece = 1.0 - (double) r.f[b][0];
            denom += wgt;
            top += wgt * piece;
            continue;
        }
        const REAL wq = vft_uniform_load<REAL>(Q.w + p);
        const typename UVec4<REAL>::type fq = vft_uniform_load4<REAL>(Q.f + p * 4);
        const REAL ww = wq * r.w[b];   // numeric_t product, NJ.tcc:1176
        const double wgt = (double) ww;
        const REAL q0 = fq.x * r.f[b][0], q1 = fq.y * r.f[b][QLEAF ? 0 : 1], q2 = fq.z * r.f[b][QLEAF ? 0 : 2];
        const REAL q3 = fq.w * r.f[b][QLEAF ? 0 : 3];
        double piece = 1.0 - (double) q0;
        piece -= (double) q1;
        piece -= (double) q2;
        piece -= (double) q3;
        denom += wgt;
        top += wgt * piece;
    }
}

// Epilogue of one (query, target) distance: setDistCriterion's diameter correction and criterion (NJ.tcc:1115-1124)
// or, for the out-profile as query, setOutDistance (NJ.tcc:1046-1053).
template <typename REAL, int MODE>
__device__ __forceinline__ void vft_sweep_finish(const Arena<REAL> &A, const SweepArgs &s, const SweepOut<REAL> &O,
                                                 int64_t j, REAL dist, REAL weight, bool seqPair, REAL &cmin, REAL &cmax) {
    if (MODE == MODE_CRIT) {
        if (!seqPair) {
            const REAL dd = A.diameter[s.query] + A.diameter[j];
            dist = dist - dd;   // NJ.tcc:1120
        }
        const REAL crit = vft_criterion<REAL>(dist, A.outDist[s.query], A.nOutActive[s.query], A.outDist[j],
                                              A.nOutActive[j], s.nActive);
        O.dist[j] = dist;
        O.weight[j] = weight;
        O.crit[j] = crit;
        cmin = crit < cmin ? crit : cmin;
        cmax = crit > cmax ? crit : cmax;
    } else {
        const REAL od = vft_out_distance<REAL>(dist, weight, s.nActive, A.selfweight[j], A.selfdist[j], A.diameter[j],
                                               s.totdiam);
        A.outDist[j] = od;
        A.nOutActive[j] = (int32_t) s.nActive;
        A.mOutDist[j] = od;
        A.mNOut[j] = (int32_t) s.nActive;
    }
}

// does target j need a distance in this launch?  MODE_CRIT: every active node (the others get the reference's
// "illegal join" sentinel, NJ.tcc:3586-3590); MODE_OUTDIST: active nodes whose out-distance is too stale.
template <typename REAL, int MODE>
__device__ __forceinline__ bool vft_sweep_wants(const Arena<REAL> &A, const SweepArgs &s, const SweepOut<REAL> &O, int64_t j) {
    const bool active = A.parent[j] < 0;
    if (MODE == MODE_CRIT) {
        if (!active) {
            O.dist[j] = (REAL) 1e20;
            O.crit[j] = (REAL) 1e20;
            O.weight[j] = 0;
        }
        return active;
    }
    return active && (s.force || ((int64_t) A.nOutActive[j] - s.nActive > s.nDiffAllow)) &&
           (int64_t) A.nOutActive[j] != s.nActive;
}

// Leaf targets against a PROFILE query (an internal node or the out-profile): one workgroup per VFT_LEAF_SPAN
// consecutive leaves.
// A leaf column is a code with weight 1, so its contribution to (top, denom) depends only on (column, code):
//     code c:  denom += (double) wq,   top += (double) wq * (1.0 - (double) fq[c])     (NJ.tcc:1176-1182, 922-930)
//     gap:     nothing (adding +0.0 to the running double sums is exact)
// The 5 x nPos table of those addends is built once per workgroup in LDS; each lane then walks its leaf's columns
// in the reference's order doing one LDS read and two double adds per column - same bits as evaluating the
// products per lane, a fifth of the instructions.
// Joined leaves are scattered through the id space (half of them late in a run), so the workgroup first COMPACTS
// the leaves that need a distance into an LDS list and hands them out densely: no idle lanes in the column loop.
#define VFT_PTILE 256
#define VFT_LEAF_SPAN 1024
// one 16-column chunk of NBT leaves against the LDS table row
template <int NBT>
__device__ __forceinline__ void vft_leaf_table_chunk(const double2 *row, const uint4 *t, double *top, double *denom) {
    // groups of GC columns: all LDS reads of a group (~8) are issued before the first add needs one
    constexpr int GC = NBT == 1 ? 8 : NBT == 2 ? 4 : 2;
#pragma unroll
    for (int g = 0; g < VFT_CHUNK; g += GC) {
        double2 v[NBT][GC];
#pragma unroll
        for (int bt = 0; bt < NBT; bt++)
#pragma unroll
            for (int q = 0; q < GC; q++) {
                // stored byte: 0x10 | one-hot nibble, 0 for a gap -> table column 0..3, or 4 for a gap
                const int idx = __ffs((int) (vft_byte(t[bt], g + q) | 0x10u)) - 1;
                v[bt][q] = row[(g + q) * 5 + idx];   // padding columns hold zeros
            }
#pragma unroll
        for (int q = 0; q < GC; q++)
#pragma unroll
            for (int bt = 0; bt < NBT; bt++) {
                denom[bt] += v[bt][q].y;
                top[bt] += v[bt][q].x;
            }
    }
}

template <typename REAL, int NBT>
__device__ __forceinline__ void vft_leaf_table_walk(const Arena<REAL> &A, const QueryBuf<REAL> &Q, double2 *tab,
                                                    const int64_t *tj, double *top, double *denom) {
    const int tid = threadIdx.x;
    const int64_t nPos = A.d.nPos;
    const uint4 *lp[NBT];
#pragma unroll
    for (int bt = 0; bt < NBT; bt++) lp[bt] = A.leafT + vft_leaf_idx(A.d, tj[bt] >> 6, 0, (int) (tj[bt] & 63));
    for (int64_t p0 = 0; p0 < nPos; p0 += VFT_PTILE) {
        __syncthreads();
        {   // the table of this position tile (written by k_extract_query): independent, coalesced 16-byte loads
            const int64_t nTab = ((int64_t) A.d.nChunk * VFT_CHUNK - p0) * 5;
            for (int e = tid; e < VFT_PTILE * 5; e += VFT_WG)
                tab[e] = e < nTab ? Q.tab[p0 * 5 + e] : make_double2(0.0, 0.0);
        }
        __syncthreads();
        const int c0 = (int) (p0 / VFT_CHUNK);
        const int c1 = (int) (((p0 + VFT_PTILE < nPos ? p0 + VFT_PTILE : nPos) + VFT_CHUNK - 1) / VFT_CHUNK);
        // the leaves' bytes run one chunk ahead of the table walk (a chunk is ~0.3 us of work, a load ~1 us away):
        // explicit ping-pong between two register sets, so that a wait only covers the older set (vmcnt is in-order)
        uint4 ta[NBT], tb[NBT];
#pragma unroll
        for (int bt = 0; bt < NBT; bt++) ta[bt] = lp[bt][(int64_t) c0 * VFT_TILE];
        for (int c = c0; c < c1; c += 2) {
            const int cb = c + 1 < c1 ? c + 1 : c, ca = c + 2 < c1 ? c + 2 : c;
#pragma unroll
            for (int bt = 0; bt < NBT; bt++) tb[bt] = lp[bt][(int64_t) cb * VFT_TILE];
            __builtin_amdgcn_sched_barrier(0);   // keep the loads above the chunk they overlap with
            vft_leaf_table_chunk<NBT>(tab + (int64_t) (c - c0) * VFT_CHUNK * 5, ta, top, denom);
#pragma unroll
            for (int bt = 0; bt < NBT; bt++) ta[bt] = lp[bt][(int64_t) ca * VFT_TILE];
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 < c1) vft_leaf_table_chunk<NBT>(tab + (int64_t) (c + 1 - c0) * VFT_CHUNK * 5, tb, top, denom);
        }
    }
}

template <typename REAL, int MODE>
__device__ __forceinline__ void vft_leaf_table_wg(const Arena<REAL> &A, const QueryBuf<REAL> &Q, const SweepArgs &s,
                                                  const SweepOut<REAL> &O, int64_t base, REAL &cmin, REAL &cmax) {
    constexpr int NB = VFT_LEAF_SPAN / VFT_WG, NW = VFT_WG / 64;
    __shared__ double2 tab[VFT_PTILE * 5];
    __shared__ unsigned short list[VFT_LEAF_SPAN];
    __shared__ int segCnt[NB * NW];
    const int tid = threadIdx.x, wave = tid >> 6;
    // 1. compaction (order-preserving, so neighbouring list entries are neighbours in memory)
    unsigned long long bal[NB];
#pragma unroll
    for (int r = 0; r < NB; r++) {
        const int64_t j = base + r * VFT_WG + tid;
        bal[r] = __ballot(j < s.leafEnd && vft_sweep_wants<REAL, MODE>(A, s, O, j));
        if ((tid & 63) == 0) segCnt[r * NW + wave] = __popcll(bal[r]);
    }
    __syncthreads();
    int nAct = 0, off[NB];
#pragma unroll
    for (int seg = 0; seg < NB * NW; seg++) {
#pragma unroll
        for (int r = 0; r < NB; r++)
            if (seg == r * NW + wave) off[r] = nAct;
        nAct += segCnt[seg];
    }
#pragma unroll
    for (int r = 0; r < NB; r++) {
        const int rank = (int) __builtin_amdgcn_mbcnt_hi((unsigned int) (bal[r] >> 32),
                               __builtin_amdgcn_mbcnt_lo((unsigned int) bal[r], 0u));
        if ((bal[r] >> (tid & 63)) & 1ull) list[off[r] + rank] = (unsigned short) (r * VFT_WG + tid);
    }
    __syncthreads();
    if (nAct == 0) return;   // workgroup-uniform
    const int nBatch = (nAct + VFT_WG - 1) / VFT_WG;
    // 2. this thread's leaves (one per batch) and their column walk; the number of batches is made a compile-time
    //    constant so that the loads running one chunk ahead can be waited for individually (vmcnt is in-order)
    int64_t tj[NB];
    double top[NB], denom[NB];
#pragma unroll
    for (int bt = 0; bt < NB; bt++) {
        const int idx = bt * VFT_WG + tid;
        tj[bt] = base + list[idx < nAct ? idx : 0];   // lanes beyond the list redo entry 0 and drop the result
        top[bt] = 0;
        denom[bt] = 0;
    }
    if (VFT_ABLATE_BITS(s) & 32) return;     // (ablation: compaction only)
    if (!(VFT_ABLATE_BITS(s) & 16)) switch (nBatch) {   // (ablation 16: no column walk)
    case 1: vft_leaf_table_walk<REAL, 1>(A, Q, tab, tj, top, denom); break;
    case 2: vft_leaf_table_walk<REAL, 2>(A, Q, tab, tj, top, denom); break;
    case 3: vft_leaf_table_walk<REAL, 3>(A, Q, tab, tj, top, denom); break;
    default: vft_leaf_table_walk<REAL, 4>(A, Q, tab, tj, top, denom); break;
    }
    // 3. epilogue
#pragma unroll
    for (int bt = 0; bt < NB; bt++) {
        if (bt * VFT_WG + tid < nAct) {
            const REAL weight = (REAL) (denom[bt] > 0 ? denom[bt] : 0.01);
            const REAL dist = (REAL) (denom[bt] > 0 ? top[bt] / denom[bt] : 1.0);
            vft_sweep_finish<REAL, MODE>(A, s, O, tj[bt], dist, weight, false, cmin, cmax);
        }
    }
}

// A sweep covers its targets with two kinds of workgroups: "heavy" ones, a target per lane, over every id that is not covered by a
// table workgroup (internal targets, all targets of a leaf query, range remainders) - HBM-bound, they want loads in flight - and
// "table" ones over s.nLeafWG spans of VFT_LEAF_SPAN leaves (profile query only) - LDS / VALU-bound.  part = index into the
// per-workgroup min/max: table workgroups [0, nLeafWG), heavy ones behind them.
//   k_sweep_nt        heavy workgroups only (leaf queries; MODE_OUTDIST's first launch)
//   k_sweep_nt_table  table workgroups only (MODE_OUTDIST's second launch)
//   k_sweep_nt_both   a profile seed's criteria in ONE launch: every third of the first 3 * nLeafWG workgroups is a table workgroup, so
//                     that the CUs hold both kinds side by side - the table walk (34 us on its own, bound by LDS reads) hides under
//                     the stream of the internal targets (80 us on its own, bound by HBM) instead of following it
template <typename REAL, int MODE>
__global__ __launch_bounds__(VFT_WG) void k_sweep_nt_table(Arena<REAL> A, QueryBuf<REAL> Q, SweepArgs s, SweepOut<REAL> O) {
    REAL cmin = (REAL) 1e30, cmax = (REAL) -1e30;
    vft_leaf_table_wg<REAL, MODE>(A, Q, s, O, s.lo + (int64_t) blockIdx.x * VFT_LEAF_SPAN, cmin, cmax);
    if (MODE == MODE_CRIT) vft_block_minmax<REAL>(cmin, cmax, O.partMin, O.partMax, (int) blockIdx.x);
}

// heavy workgroup number wg of a sweep (every thread of the workgroup must call)
template <typename REAL, int MODE_>
__device__ __forceinline__ void vft_sweep_heavy_wg(const Arena<REAL> &A, const QueryBuf<REAL> &Q, const SweepArgs &s, const SweepOut<REAL> &O, int wg) {
    constexpr bool QLEAF = MODE_ == MODE_CRIT_LEAFQ;
    constexpr int MODE = QLEAF ? MODE_CRIT : MODE_;
    REAL cmin = (REAL) 1e30, cmax = (REAL) -1e30;
    // highest ids first: the workgroups of internal targets (~10x the bytes of a leaf) start before the leaf ones,
    // which then fill the idle slots instead of running ahead of them
    const int64_t j = s.heavyLo + (int64_t) wg * VFT_WG + threadIdx.x;
    const int lane = (int) (j & 63);
    const int64_t tile = j >> 6;
    const bool work = j < s.hi && !(s.nLeafWG && j < s.leafEnd) && vft_sweep_wants<REAL, MODE>(A, s, O, j);
    if (work) {
        const int64_t nPos = A.d.nPos;
        const bool targetLeaf = j < A.d.nSeqs;
        REAL dist, weight;
        if (MODE == MODE_CRIT && s.queryIsLeaf && targetLeaf) {
            int nUse = 0, nSame = 0;
            for (int c = 0; c < A.d.nChunk; c++) {
                const uint4 t = A.leafT[vft_leaf_idx(A.d, tile, c, lane)];
                vft_seq_counts(t, Q.enc[c], nUse, nSame);
            }
            const double top = (double) (nUse - nSame);
            weight = (REAL) (double) nUse;
            dist = (REAL) (nUse > 0 ? top / (double) nUse : 1.0);
        } else {
            // leaf targets of a profile query belong to k_sweep_nt_table: everything left here is an internal node
            double top = 0, denom = 0;
            {
                // internal targets: codes dense; explicit weights and vectors from the tile's packed streams
                // the 64 lanes of a wave share one tile: readfirstlane makes every base address below wave-uniform
                // (SGPR base + 32-bit lane offset addressing; masks through the scalar cache)
                const int64_t pt = (int64_t) __builtin_amdgcn_readfirstlane((int) (tile - A.d.firstProfTile));
                const REAL *wT = A.profW + vft_wstream_base(A.d, pt);
                const uint4 *cT = A.profC + vft_c_idx(A.d, pt, 0, 0);
                const vft_smask_t mM = (vft_smask_t) (A.colMask + vft_meta_idx(A.d, pt, 0));
                const vft_soff_t mO = (vft_soff_t) (A.colOff + vft_meta_idx(A.d, pt, 0));
                const REAL *fT = A.profF + vft_fstream_base(A.d, pt);
                // VFT_SUB-column groups: every load of a group is issued before its first result is used; the codes
                // run one 16-column chunk ahead (issued after the group's loads so that nothing waits on them).
                IntChunk<REAL, QLEAF> ca;
                uint4 cur = cT[lane];
                const int nChunk = A.d.nChunk;
                for (int c = 0; c < nChunk; c++) {
                    uint4 nxt;
#pragma unroll
                    for (int sub = 0; sub < VFT_CHUNK / VFT_SUB; sub++) {
                        vft_int_chunk_load<REAL, QLEAF>(ca, c, sub, cur, wT, mM, mO, fT, Q, VFT_ABLATE_BITS(s));
                        if (sub == 0) nxt = cT[(int64_t) (c + 1 < nChunk ? c + 1 : c) * VFT_TILE + lane];
                        vft_int_chunk_consume<REAL, QLEAF>(ca, c, sub, Q, top, denom, VFT_ABLATE_BITS(s));
                    }
                    cur = nxt;
                }
            }
            weight = (REAL) (denom > 0 ? denom : 0.01);
            dist = (REAL) (denom > 0 ? top / denom : 1.0);
        }
        vft_sweep_finish<REAL, MODE>(A, s, O, j, dist, weight, s.queryIsLeaf && targetLeaf, cmin, cmax);
    }
    if (MODE == MODE_CRIT) vft_block_minmax<REAL>(cmin, cmax, O.partMin, O.partMax, s.nLeafWG + wg);
}

template <typename REAL, int MODE_>
__global__ __launch_bounds__(VFT_WG) void k_sweep_nt(Arena<REAL> A, QueryBuf<REAL> Q, SweepArgs s, SweepOut<REAL> O) {
    // highest ids first: the workgroups of internal targets (~10x the bytes of a leaf) start before the leaf ones,
    // which then fill the idle slots instead of running ahead of them
    vft_sweep_heavy_wg<REAL, MODE_>(A, Q, s, O, (int) gridDim.x - 1 - (int) blockIdx.x);
}

template <typename REAL>
__global__ __launch_bounds__(VFT_WG) void k_sweep_nt_both(Arena<REAL> A, QueryBuf<REAL> Q, SweepArgs s, SweepOut<REAL> O) {
    const int b = (int) blockIdx.x, nT = s.nLeafWG, nHeavy = (int) gridDim.x - nT;
    // block -> (kind, number): table workgroups at every third block while two heavy ones are there for each (else all of them first)
    const bool spread = nHeavy >= 2 * nT;
    const bool mixed = spread && b < 3 * nT;
    const bool isTable = spread ? (mixed && b % 3 == 0) : b < nT;
    if (isTable) {
        const int t = spread ? b / 3 : b;
        REAL cmin = (REAL) 1e30, cmax = (REAL) -1e30;
        vft_leaf_table_wg<REAL, MODE_CRIT>(A, Q, s, O, s.lo + (int64_t) t * VFT_LEAF_SPAN, cmin, cmax);
        vft_block_minmax<REAL>(cmin, cmax, O.partMin, O.partMax, t);
        return;
    }
    const int h = mixed ? b - b / 3 - 1 : b - nT;                      // heavy workgroups in launch order ...
    vft_sweep_heavy_wg<REAL, MODE_CRIT>(A, Q, s, O, nHeavy - 1 - h);   // ... take the highest ids first, as in k_sweep_nt
}

// ---- S leaf seeds in ONE pass over the targets (vft_sweep_batch: setAllLeafTopHits sweeps seed after seed and nothing changes
// in between, NJ.tcc:3798-3880).  A leaf seed's sweep is bound by the target stream (0.86 of the HBM peak by algorithmic bytes), and
// most of what a lane does per column does not depend on the query: the tile's masks and stream offsets, the lane's slots in
// the packed streams, its code byte, its weight, its vector.  Here a lane does that once and evaluates S queries on it - the
// arithmetic of every (query, target) pair is MODE_CRIT_LEAFQ's, operation for operation (same bits): one pass streams the targets
// for S sweeps.
//     piece = 1 - f2[cq]   (NJ.tcc:924; f2 of a plain target code is the one-hot of the code)
//     wgt   = the target's weight, or 0 at a gap of the query (the leaf's weight is 1: w1 * w2 is w2 exactly, NJ.tcc:1176)
template <typename REAL, int S>
struct MultiLeafQ {
    QueryBuf<REAL> Q[S];
    SweepOut<REAL> O[S];
    int64_t query[S];
    const REAL *mq;   // profile seeds: the S queries' weights and vectors interleaved by column (QuerySlot)
};

template <typename REAL, int SUB>
struct IntChunkAll {
    REAL w[SUB];
    typename UVec4<REAL>::type f[SUB];
};

// (Measured and dropped: the same loads as always-issued buffer loads - lanes without a vector asking beyond the stream, the answer
// OR-ed onto the implied value - so that no load sits in a branch and two column groups can alternate with exact wait counters:
// 142 -> 172-220 us per pass of four seeds.  Most (tile, column) pairs have no vector lane at all and the branch skips the
// instruction; issued unconditionally, the 128-bit loads cost the texture path more than the exposed latency they hide.)
template <typename REAL, int SUB>
__device__ __forceinline__ void vft_int_chunk_load_all(IntChunkAll<REAL, SUB> &r, int c, int sub, const uint4 codes, const REAL *wT,
                                                       vft_smask_t mM, vft_soff_t mO, const REAL *fT) {
    const int64_t p0 = (int64_t) c * VFT_CHUNK + sub * SUB;
    vft_u4_t mk[SUB];
    vft_u2_t of[SUB];
#pragma unroll
    for (int b = 0; b < SUB; b++) {   // wave-uniform: scalar loads
        mk[b] = mM[p0 + b];
        of[b] = mO[p0 + b];
    }
#pragma unroll
    for (int b = 0; b < SUB; b++) {
        const unsigned long long mv = ((unsigned long long) mk[b].y << 32) | mk[b].x;
        const unsigned long long mw = ((unsigned long long) mk[b].w << 32) | mk[b].z;
        const bool hv = __builtin_amdgcn_inverse_ballot_w64(mv);
        const bool hw = __builtin_amdgcn_inverse_ballot_w64(mw);
        const uint32_t slotV = __builtin_amdgcn_mbcnt_hi(mk[b].y, __builtin_amdgcn_mbcnt_lo(mk[b].x, of[b].x));
        const uint32_t slotW = __builtin_amdgcn_mbcnt_hi(mk[b].w, __builtin_amdgcn_mbcnt_lo(mk[b].z, of[b].y));
        const uint32_t cd = vft_byte(codes, sub * SUB + b);
        r.w[b] = (hv || cd != VFT_NOCODE_) ? (REAL) 1 : (REAL) 0;   // implicit weight (vft_layout.h)
        const char *wB = (const char *) wT, *fB = (const char *) fT;
        if (hw) r.w[b] = *(const REAL *) (wB + slotW * (uint32_t) sizeof(REAL));
        // a plain code as its one-hot vector: f[k] = (cd == k); entry cq of it is MODE_CRIT_LEAFQ's (cd == cq) for every code
        // cq, and at a gap of the query (cq = 127 reads entry 3) the value does not matter: the weight is 0 there
        uint32_t oh;   // (asm: keeps the compiler from turning the byte reads back into compare + select)
        asm("v_lshlrev_b32 %0, %1, 1" : "=v"(oh) : "v"((cd & 3u) * 8u));
        if (cd == VFT_NOCODE_) oh = 0;   // (a target's gap or vector column: no code equals it; MODE_CRIT_LEAFQ's cd == cq is false too)
        r.f[b].x = (REAL) (oh & 0xFFu);
        r.f[b].y = (REAL) ((oh >> 8) & 0xFFu);
        r.f[b].z = (REAL) ((oh >> 16) & 0xFFu);
        r.f[b].w = (REAL) ((oh >> 24) & 0xFFu);
        // the column's whole vector: the 16 (32) bytes a leaf query's single frequency sits in
        if (hv) r.f[b] = *(const typename UVec4<REAL>::type *) (fB + slotV * 4u * (uint32_t) sizeof(REAL));
    }
}

// per-workgroup (min, max) of S queries' criteria at once
template <typename REAL, int S>
__device__ __forceinline__ void vft_block_minmax_multi(REAL *cmin, REAL *cmax, const SweepOut<REAL> *O, int part) {
    constexpr int NW = VFT_WG / 64;
    __shared__ REAL smin[S][NW], smax[S][NW];
#pragma unroll
    for (int q = 0; q < S; q++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const REAL a = __shfl_xor(cmin[q], off, 64), b = __shfl_xor(cmax[q], off, 64);
            cmin[q] = a < cmin[q] ? a : cmin[q];
            cmax[q] = b > cmax[q] ? b : cmax[q];
        }
        if ((threadIdx.x & 63) == 0) {
            smin[q][threadIdx.x >> 6] = cmin[q];
            smax[q][threadIdx.x >> 6] = cmax[q];
        }
    }
    __syncthreads();
    if (threadIdx.x < S) {
        const int q = threadIdx.x;
        REAL lo = smin[q][0], hi = smax[q][0];
#pragma unroll
        for (int w = 1; w < NW; w++) {
            lo = smin[q][w] < lo ? smin[q][w] : lo;
            hi = smax[q][w] > hi ? smax[q][w] : hi;
        }
        O[q].partMin[part] = lo;
        O[q].partMax[part] = hi;
    }
}

// cond ? b : a for a wave-uniform condition, as ONE v_cndmask on a scalar lane mask.  (Written in C the compiler either branches
// on the scalar condition - a taken branch is ~40 cycles and every one of them ended in s_waitcnt vmcnt(0) - or, for an indexed
// vector, goes through scratch memory.)
__device__ __forceinline__ uint32_t vft_usel_b32(uint32_t a, uint32_t b, unsigned long long mask) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
    return r;
}
__device__ __forceinline__ float vft_usel(float a, float b, unsigned long long mask) {
    return __uint_as_float(vft_usel_b32(__float_as_uint(a), __float_as_uint(b), mask));
}
__device__ __forceinline__ double vft_usel(double a, double b, unsigned long long mask) {
    return __hiloint2double((int) vft_usel_b32((uint32_t) __double2hiint(a), (uint32_t) __double2hiint(b), mask),
                            (int) vft_usel_b32((uint32_t) __double2loint(a), (uint32_t) __double2loint(b), mask));
}

// S queries on one group of SUB columns
template <typename REAL, int S, int SUB>
__device__ __forceinline__ void vft_int_chunk_consume_all(const IntChunkAll<REAL, SUB> &ca, int64_t p0, const MultiLeafQ<REAL, S> &M, double *top, double *denom) {
#pragma unroll
    for (int b = 0; b < SUB; b++) {
        const double wd = (double) ca.w[b];
#pragma unroll
        for (int q = 0; q < S; q++) {
            const uint32_t cq = vft_query_code<REAL>(M.Q[q], p0, b);   // wave-uniform
            const unsigned long long m0 = (cq & 1u) ? ~0ull : 0ull, m1 = (cq & 2u) ? ~0ull : 0ull;
            const REAL fsel = vft_usel(vft_usel(ca.f[b].x, ca.f[b].y, m0), vft_usel(ca.f[b].z, ca.f[b].w, m0), m1);   // f2[cq & 3]
            // the target's weight, or +0.0 at a gap of the query: a product with a scalar 1.0 / 0.0 (exact: weights are finite and
            // not negative) - written as a select on the uniform condition the compiler branches around it
            const double wgt = wd * (cq != VFT_NOCODE_ ? 1.0 : 0.0);
            const double piece = 1.0 - (double) fsel;
            denom[q] += wgt;
            top[q] += wgt * piece;
        }
    }
}

template <typename REAL, int S>
__global__ __launch_bounds__(VFT_WG) void k_sweep_nt_leafq_multi(Arena<REAL> A, MultiLeafQ<REAL, S> M, SweepArgs s) {
    constexpr int SUB = 8;   // (groups of 4 columns: 142 -> 152 us per pass of four seeds)
    REAL cmin[S], cmax[S];
#pragma unroll
    for (int q = 0; q < S; q++) {
        cmin[q] = (REAL) 1e30;
        cmax[q] = (REAL) -1e30;
    }
    const int wg = (int) gridDim.x - 1 - (int) blockIdx.x;   // highest ids first, as in k_sweep_nt
    const int64_t j = s.heavyLo + (int64_t) wg * VFT_WG + threadIdx.x;
    const int lane = (int) (j & 63);
    const int64_t tile = j >> 6;
    bool work = false;
    if (j < s.hi) {
        work = A.parent[j] < 0;
        if (!work) {   // the reference's "illegal join" sentinel (NJ.tcc:3586-3590), as vft_sweep_wants
#pragma unroll
            for (int q = 0; q < S; q++) {
                M.O[q].dist[j] = (REAL) 1e20;
                M.O[q].crit[j] = (REAL) 1e20;
                M.O[q].weight[j] = 0;
            }
        }
    }
    if (work) {
        const bool targetLeaf = j < A.d.nSeqs;
        REAL dist[S], weight[S];
        if (targetLeaf) {
            int nUse[S], nSame[S];
#pragma unroll
            for (int q = 0; q < S; q++) nUse[q] = nSame[q] = 0;
            for (int c = 0; c < A.d.nChunk; c++) {
                const uint4 t = A.leafT[vft_leaf_idx(A.d, tile, c, lane)];
#pragma unroll
                for (int q = 0; q < S; q++) vft_seq_counts(t, M.Q[q].enc[c], nUse[q], nSame[q]);
            }
#pragma unroll
            for (int q = 0; q < S; q++) {
                const double top = (double) (nUse[q] - nSame[q]);
                weight[q] = (REAL) (double) nUse[q];
                dist[q] = (REAL) (nUse[q] > 0 ? top / (double) nUse[q] : 1.0);
            }
        } else {
            double top[S], denom[S];
#pragma unroll
            for (int q = 0; q < S; q++) top[q] = denom[q] = 0;
            const int64_t pt = (int64_t) __builtin_amdgcn_readfirstlane((int) (tile - A.d.firstProfTile));
            const uint4 *cT = A.profC + vft_c_idx(A.d, pt, 0, 0);
            const vft_smask_t mM = (vft_smask_t) (A.colMask + vft_meta_idx(A.d, pt, 0));
            const vft_soff_t mO = (vft_soff_t) (A.colOff + vft_meta_idx(A.d, pt, 0));
            const REAL *wT = A.profW + vft_wstream_base(A.d, pt);
            const REAL *fT = A.profF + vft_fstream_base(A.d, pt);
            uint4 cur = cT[lane];
            const int nChunk = A.d.nChunk;
            constexpr int NSUB = VFT_CHUNK / SUB;
            IntChunkAll<REAL, SUB> ca;
            for (int c = 0; c < nChunk; c++) {
                uint4 nxt;
#pragma unroll
                for (int sub = 0; sub < NSUB; sub++) {
                    vft_int_chunk_load_all<REAL, SUB>(ca, c, sub, cur, wT, mM, mO, fT);
                    if (sub == 0) nxt = cT[(int64_t) (c + 1 < nChunk ? c + 1 : c) * VFT_TILE + lane];
                    vft_int_chunk_consume_all<REAL, S, SUB>(ca, (int64_t) c * VFT_CHUNK + sub * SUB, M, top, denom);
                }
                cur = nxt;
            }
#pragma unroll
            for (int q = 0; q < S; q++) {
                weight[q] = (REAL) (denom[q] > 0 ? denom[q] : 0.01);
                dist[q] = (REAL) (denom[q] > 0 ? top[q] / denom[q] : 1.0);
            }
        }
#pragma unroll
        for (int q = 0; q < S; q++) {
            SweepArgs sq = s;
            sq.query = M.query[q];
            vft_sweep_finish<REAL, MODE_CRIT>(A, sq, M.O[q], j, dist[q], weight[q], targetLeaf, cmin[q], cmax[q]);
        }
    }
    vft_block_minmax_multi<REAL, S>(cmin, cmax, M.O, wg);
}

// ---- S PROFILE seeds (internal nodes) in one launch: the heavy workgroups stream the internal targets once and evaluate the S
// queries' profile x profile distances on every column they decode (MODE_CRIT's arithmetic, operation for operation: numeric_t
// products, double subtractions, NJ.tcc:933-937, :1172-1183); behind them nLeafWG table workgroups walk the leaf targets for all S
// queries at once (vft_leaf_table_wg_multi below; until round 6: S x nLeafWG workgroups, one query each).
// The S queries' columns are wave-uniform: scalar loads, 5 dwords per (query, column), from the group's interleaved buffer (QuerySlot).  Written as one loop over SUB x S the compiler
// hoists every load of the group to the front - 160 SGPRs at S = 4 - and spills them through v_writelane / v_readlane: 382 of the
// 1 781 VALU instructions of a 16-column chunk were that traffic (tools/isa_loops.py, round 6).  Here a column's scalars are loaded
// while the column before it is being consumed and the scheduler may not move anything across the column boundaries: two columns'
// scalars live, no spill.  The four products of a column are one vector multiply (v_pk_mul_f32 in single precision: two instructions,
// each lane's product IEEE as before).
template <typename REAL, int S>
struct QueryCol {
    REAL w[S];
    typename UVec4<REAL>::type f[S];
};
template <typename REAL, int S>
__device__ __forceinline__ void vft_query_col_load(QueryCol<REAL, S> &c, const MultiLeafQ<REAL, S> &M, int64_t p) {
    const REAL *src = M.mq + p * VFT_MQ_STRIDE(S);
    const typename UVec4<REAL>::type wv = vft_uniform_load4<REAL>(src + 4 * S);
#pragma unroll
    for (int q = 0; q < S; q++) {
        c.w[q] = q == 0 ? wv.x : q == 1 ? wv.y : q == 2 ? wv.z : wv.w;
        c.f[q] = vft_uniform_load4<REAL>(src + 4 * q);
    }
}

template <typename REAL, int S, int SUB>
__device__ __forceinline__ void vft_int_chunk_consume_prof(const IntChunkAll<REAL, SUB> &ca, int64_t p0, const MultiLeafQ<REAL, S> &M, double *top, double *denom) {
    QueryCol<REAL, S> cur, nxt;
    vft_query_col_load<REAL, S>(cur, M, p0);
#pragma unroll
    for (int b = 0; b < SUB; b++) {
        if (b + 1 < SUB) vft_query_col_load<REAL, S>(nxt, M, p0 + b + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < S; q++) {
            const REAL ww = cur.w[q] * ca.w[b];   // numeric_t product, NJ.tcc:1176
            const double wgt = (double) ww;
            const typename UVec4<REAL>::type pr = cur.f[q] * ca.f[b];   // the four numeric_t products of NJ.tcc:933-937
            double piece = 1.0 - (double) pr.x;
            piece -= (double) pr.y;
            piece -= (double) pr.z;
            piece -= (double) pr.w;
            denom[q] += wgt;
            top[q] += wgt * piece;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (b + 1 < SUB) cur = nxt;
    }
}

// ---- the leaf targets of S profile seeds in one walk (round 6): what vft_leaf_table_wg does for one seed - compaction of the span's
// active leaves, the (column, code) addend table in LDS, one LDS read and two double adds per column - with the compaction, the
// leaves' bytes and the table index of every (leaf, column) shared by the S seeds: per column a lane computes the index once and
// reads S tables.  Same addends in the same order per (seed, leaf) as the one-seed walk: same bits.  The S tables are staged
// VFT_PTILE_M positions at a time (S x 112 x 5 x 16 B = 35 KB: four workgroups of this kernel still share a CU's LDS), the span's
// compacted leaves are walked 2 x VFT_WG at a time (two leaves per lane: S x 2 reads in flight per column).
#define VFT_PTILE_M 112
template <int NBT, int S>
__device__ __forceinline__ void vft_leaf_table_chunk_multi(const double2 *row, const uint4 *t, double (*top)[S], double (*denom)[S]) {
    constexpr int GC = NBT * S >= 8 ? 1 : NBT * S >= 4 ? 2 : 4;
#pragma unroll
    for (int g = 0; g < VFT_CHUNK; g += GC) {
        double2 v[NBT][GC][S];
#pragma unroll
        for (int bt = 0; bt < NBT; bt++)
#pragma unroll
            for (int q = 0; q < GC; q++) {
                const int idx = __ffs((int) (vft_byte(t[bt], g + q) | 0x10u)) - 1;   // table column 0..3, or 4 for a gap (vft_leaf_table_chunk)
#pragma unroll
                for (int sd = 0; sd < S; sd++) v[bt][q][sd] = row[sd * (VFT_PTILE_M * 5) + (g + q) * 5 + idx];
            }
#pragma unroll
        for (int q = 0; q < GC; q++)
#pragma unroll
            for (int bt = 0; bt < NBT; bt++)
#pragma unroll
                for (int sd = 0; sd < S; sd++) {
                    denom[bt][sd] += v[bt][q][sd].y;
                    top[bt][sd] += v[bt][q][sd].x;
                }
    }
}

template <typename REAL, int NBT, int S>
__device__ __forceinline__ void vft_leaf_table_walk_multi(const Arena<REAL> &A, const MultiLeafQ<REAL, S> &M, double2 *tab, const int64_t *tj,
                                                          double (*top)[S], double (*denom)[S]) {
    const int tid = threadIdx.x;
    const int64_t nPos = A.d.nPos;
    const uint4 *lp[NBT];
#pragma unroll
    for (int bt = 0; bt < NBT; bt++) lp[bt] = A.leafT + vft_leaf_idx(A.d, tj[bt] >> 6, 0, (int) (tj[bt] & 63));
    for (int64_t p0 = 0; p0 < nPos; p0 += VFT_PTILE_M) {
        __syncthreads();
        {
            const int64_t nTab = ((int64_t) A.d.nChunk * VFT_CHUNK - p0) * 5;
#pragma unroll
            for (int sd = 0; sd < S; sd++)
                for (int e = tid; e < VFT_PTILE_M * 5; e += VFT_WG)
                    tab[sd * (VFT_PTILE_M * 5) + e] = e < nTab ? M.Q[sd].tab[p0 * 5 + e] : make_double2(0.0, 0.0);
        }
        __syncthreads();
        const int c0 = (int) (p0 / VFT_CHUNK);
        const int c1 = (int) (((p0 + VFT_PTILE_M < nPos ? p0 + VFT_PTILE_M : nPos) + VFT_CHUNK - 1) / VFT_CHUNK);
        uint4 ta[NBT], tb[NBT];   // the leaves' bytes one chunk ahead, ping-pong (vft_leaf_table_walk)
#pragma unroll
        for (int bt = 0; bt < NBT; bt++) ta[bt] = lp[bt][(int64_t) c0 * VFT_TILE];
        for (int c = c0; c < c1; c += 2) {
            const int cb = c + 1 < c1 ? c + 1 : c, ca = c + 2 < c1 ? c + 2 : c;
#pragma unroll
            for (int bt = 0; bt < NBT; bt++) tb[bt] = lp[bt][(int64_t) cb * VFT_TILE];
            __builtin_amdgcn_sched_barrier(0);
            vft_leaf_table_chunk_multi<NBT, S>(tab + (int64_t) (c - c0) * VFT_CHUNK * 5, ta, top, denom);
#pragma unroll
            for (int bt = 0; bt < NBT; bt++) ta[bt] = lp[bt][(int64_t) ca * VFT_TILE];
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 < c1) vft_leaf_table_chunk_multi<NBT, S>(tab + (int64_t) (c + 1 - c0) * VFT_CHUNK * 5, tb, top, denom);
        }
    }
}

template <typename REAL, int S>
__device__ __forceinline__ void vft_leaf_table_wg_multi(const Arena<REAL> &A, const MultiLeafQ<REAL, S> &M, const SweepArgs &s, int64_t base,
                                                        REAL *cmin, REAL *cmax) {
    constexpr int NB = VFT_LEAF_SPAN / VFT_WG, NW = VFT_WG / 64, NBT = 2;
    __shared__ double2 tab[S * VFT_PTILE_M * 5];
    __shared__ unsigned short list[VFT_LEAF_SPAN];
    __shared__ int segCnt[NB * NW];
    const int tid = threadIdx.x, wave = tid >> 6;
    // 1. compaction of the span's active leaves (order-preserving); the others get the "illegal join" sentinel of every seed
    unsigned long long bal[NB];
#pragma unroll
    for (int r = 0; r < NB; r++) {
        const int64_t j = base + r * VFT_WG + tid;
        bool active = false;
        if (j < s.leafEnd) {
            active = A.parent[j] < 0;
            if (!active) {
#pragma unroll
                for (int q = 0; q < S; q++) {
                    M.O[q].dist[j] = (REAL) 1e20;
                    M.O[q].crit[j] = (REAL) 1e20;
                    M.O[q].weight[j] = 0;
                }
            }
        }
        bal[r] = __ballot(active);
        if ((tid & 63) == 0) segCnt[r * NW + wave] = __popcll(bal[r]);
    }
    __syncthreads();
    int nAct = 0, off[NB];
#pragma unroll
    for (int seg = 0; seg < NB * NW; seg++) {
#pragma unroll
        for (int r = 0; r < NB; r++)
            if (seg == r * NW + wave) off[r] = nAct;
        nAct += segCnt[seg];
    }
#pragma unroll
    for (int r = 0; r < NB; r++) {
        const int rank = (int) __builtin_amdgcn_mbcnt_hi((unsigned int) (bal[r] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int) bal[r], 0u));
        if ((bal[r] >> (tid & 63)) & 1ull) list[off[r] + rank] = (unsigned short) (r * VFT_WG + tid);
    }
    __syncthreads();
    // 2. two leaves per lane at a time
    for (int r0 = 0; r0 < nAct; r0 += NBT * VFT_WG) {   // workgroup-uniform
        int64_t tj[NBT];
        double top[NBT][S], denom[NBT][S];
#pragma unroll
        for (int bt = 0; bt < NBT; bt++) {
            const int idx = r0 + bt * VFT_WG + tid;
            tj[bt] = base + list[idx < nAct ? idx : 0];   // lanes beyond the list redo entry 0 and drop the result
#pragma unroll
            for (int q = 0; q < S; q++) top[bt][q] = denom[bt][q] = 0;
        }
        if (nAct - r0 <= VFT_WG) vft_leaf_table_walk_multi<REAL, 1, S>(A, M, tab, tj, top, denom);
        else vft_leaf_table_walk_multi<REAL, NBT, S>(A, M, tab, tj, top, denom);
#pragma unroll
        for (int bt = 0; bt < NBT; bt++) {
            if (r0 + bt * VFT_WG + tid < nAct) {
#pragma unroll
                for (int q = 0; q < S; q++) {
                    const REAL weight = (REAL) (denom[bt][q] > 0 ? denom[bt][q] : 0.01);
                    const REAL dist = (REAL) (denom[bt][q] > 0 ? top[bt][q] / denom[bt][q] : 1.0);
                    SweepArgs sq = s;
                    sq.query = M.query[q];
                    vft_sweep_finish<REAL, MODE_CRIT>(A, sq, M.O[q], tj[bt], dist, weight, false, cmin[q], cmax[q]);
                }
            }
        }
    }
}

// (three wavefronts per SIMD in single precision: at four - 128 VGPRs - 23 registers of the column loop went to scratch, nine scratch
// accesses per 16-column chunk and 120 MB of extra write traffic per launch (rocprofv3 WRITE_SIZE 181 MB against 60 MB of results);
// with 137 registers nothing spills: 240 -> 215 us per launch of four profile seeds)
template <typename REAL, int S>
__global__ __launch_bounds__(VFT_WG, sizeof(REAL) == 4 ? 3 : 2) void k_sweep_nt_profq_multi(Arena<REAL> A, MultiLeafQ<REAL, S> M, SweepArgs s) {
    constexpr int SUB = 8;
    const int nT = s.nLeafWG, nHeavy = (int) gridDim.x - nT, blk = (int) blockIdx.x;
    REAL cmin[S], cmax[S];
#pragma unroll
    for (int q = 0; q < S; q++) {
        cmin[q] = (REAL) 1e30;
        cmax[q] = (REAL) -1e30;
    }
    if (blk >= nHeavy) {   // a table workgroup: span blk - nHeavy of the leaves against all S queries
        const int span = blk - nHeavy;
        vft_leaf_table_wg_multi<REAL, S>(A, M, s, s.lo + (int64_t) span * VFT_LEAF_SPAN, cmin, cmax);
        vft_block_minmax_multi<REAL, S>(cmin, cmax, M.O, span);
        return;
    }
    const int wg = nHeavy - 1 - blk;   // highest ids first, as in k_sweep_nt
    const int64_t j = s.heavyLo + (int64_t) wg * VFT_WG + threadIdx.x;
    const int lane = (int) (j & 63);
    const int64_t tile = j >> 6;
    bool work = false;
    if (j < s.hi && !(nT && j < s.leafEnd)) {
        work = A.parent[j] < 0;
        if (!work) {   // the reference's "illegal join" sentinel (NJ.tcc:3586-3590), as vft_sweep_wants
#pragma unroll
            for (int q = 0; q < S; q++) {
                M.O[q].dist[j] = (REAL) 1e20;
                M.O[q].crit[j] = (REAL) 1e20;
                M.O[q].weight[j] = 0;
            }
        }
    }
    if (work) {   // (an internal node: the leaves of a profile query belong to the table workgroups - or, without any, do not exist in [heavyLo, hi))
        double top[S], denom[S];
#pragma unroll
        for (int q = 0; q < S; q++) top[q] = denom[q] = 0;
        const int64_t pt = (int64_t) __builtin_amdgcn_readfirstlane((int) (tile - A.d.firstProfTile));
        const uint4 *cT = A.profC + vft_c_idx(A.d, pt, 0, 0);
        const vft_smask_t mM = (vft_smask_t) (A.colMask + vft_meta_idx(A.d, pt, 0));
        const vft_soff_t mO = (vft_soff_t) (A.colOff + vft_meta_idx(A.d, pt, 0));
        const REAL *wT = A.profW + vft_wstream_base(A.d, pt);
        const REAL *fT = A.profF + vft_fstream_base(A.d, pt);
        uint4 cur = cT[lane];
        const int nChunk = A.d.nChunk;
        IntChunkAll<REAL, SUB> ca;
        for (int c = 0; c < nChunk; c++) {
            uint4 nxt;
#pragma unroll
            for (int sub = 0; sub < VFT_CHUNK / SUB; sub++) {
                vft_int_chunk_load_all<REAL, SUB>(ca, c, sub, cur, wT, mM, mO, fT);
                if (sub == 0) nxt = cT[(int64_t) (c + 1 < nChunk ? c + 1 : c) * VFT_TILE + lane];
                vft_int_chunk_consume_prof<REAL, S, SUB>(ca, (int64_t) c * VFT_CHUNK + sub * SUB, M, top, denom);
            }
            cur = nxt;
        }
#pragma unroll
        for (int q = 0; q < S; q++) {
            const REAL weight = (REAL) (denom[q] > 0 ? denom[q] : 0.01);
            const REAL dist = (REAL) (denom[q] > 0 ? top[q] / denom[q] : 1.0);
            SweepArgs sq = s;
            sq.query = M.query[q];
            vft_sweep_finish<REAL, MODE_CRIT>(A, sq, M.O[q], j, dist, weight, false, cmin[q], cmax[q]);
        }
    }
    vft_block_minmax_multi<REAL, S>(cmin, cmax, M.O, nT + wg);
}

// ------------------------------------------------------------------------------------------------ generic pair
// profileDist / seqDist for an arbitrary (i, j), any alphabet, with or without a distance matrix.
// cdOut: the pair's second profile is the out-profile (row-major arrays in A.out*), used by setOutDistance.
template <typename REAL, int NC>
__device__ __forceinline__ void vft_pair_generic(const Arena<REAL> &A, int64_t i, int64_t j, bool jIsOut, REAL &dist,
                                                 REAL &weight) {
    const int64_t nPos = A.d.nPos;
    if (!jIsOut && i < A.d.nSeqs && j < A.d.nSeqs) {   // seqDist, NJ.tcc:1601-1624
        double top = 0;
        int64_t nUse = 0;
        int nDiff = 0;
        const int li = (int) (i & 63), lj = (int) (j & 63);
        for (int c = 0; c < A.d.nChunk; c++) {
            const uint4 a = A.leafT[vft_leaf_idx(A.d, i >> 6, c, li)];
            const uint4 b = A.leafT[vft_leaf_idx(A.d, j >> 6, c, lj)];
            for (int t = 0; t < VFT_CHUNK; t++) {
                if ((int64_t) c * VFT_CHUNK + t >= nPos) break;
                const int ca = vft_decode<NC>(vft_byte(a, t)), cb = vft_decode<NC>(vft_byte(b, t));
                if (ca != VFT_NOCODE_ && cb != VFT_NOCODE_) {
                    nUse++;
                    if (A.dmDist) top += (double) A.dmDist[ca * NC + cb];
                    else if (ca != cb) nDiff++;
                }
            }
        }
        if (!A.dmDist) top = (double) nDiff;
        weight = (REAL) (double) nUse;
        dist = (REAL) (nUse > 0 ? top / (double) nUse : 1.0);
        return;
    }
    double top = 0, denom = 0;
    for (int64_t p = 0; p < nPos; p++) {
        Col<REAL, NC> c1, c2;
        vft_load_col<REAL, NC>(A, i, p, c1);
        if (jIsOut) {
            c2.w = A.outW[p];
            c2.code = VFT_NOCODE_;
            c2.vec = c2.w > 0;
#pragma unroll
            for (int k = 0; k < NC; k++) c2.f[k] = A.outF[p * NC + k];
        } else {
            vft_load_col<REAL, NC>(A, j, p, c2);
        }
        if (c1.w > 0 && c2.w > 0) {
            const REAL ww = c1.w * c2.w;
            const double wgt = (double) ww;
            denom += wgt;
            const double piece = vft_piece<REAL, NC>(A, c1, c2, (jIsOut && A.outCD) ? A.outCD + p * NC : nullptr);
            top += wgt * piece;
        }
    }
    weight = (REAL) (denom > 0 ? denom : 0.01);
    dist = (REAL) (denom > 0 ? top / denom : 1.0);
}

// Wave-cooperative evaluation of ONE pair, for short lists (pair lists, single out-distances, self distances) where a
// lane-per-pair walk would be a chain of ~nPos dependent memory latencies.  The per-column addends
//     wgt_p = (double)(numeric_t)(w1*w2)   and   term_p = wgt_p * piece_p          (NJ.tcc:1176-1182)
// do not depend on each other, so the 64 lanes compute them for 64 columns at a time into LDS; lane 0 then adds them
// up IN COLUMN ORDER (denom += wgt_p; top += term_p), which is exactly the reference's sequence of double
// additions — columns it skips contribute +0.0, which is exact.  All 64 lanes must call; results are broadcast.
// sW / sT: this wave's LDS scratch, nPosPad doubles each.
// one column of the pair (i, j) — or (i, out-profile) — as the reference sees it
template <typename REAL, int NC>
__device__ __forceinline__ void vft_pair_load(const Arena<REAL> &A, int64_t i, int64_t j, bool jIsOut, int64_t p,
                                              Col<REAL, NC> &c1, Col<REAL, NC> &c2, bool iRow, bool jRow) {
    if (iRow) vft_load_row<REAL, NC>(A, i, p, c1);
    else vft_load_col<REAL, NC>(A, i, p, c1);
    if (jIsOut) {
        c2.w = A.outW[p];
        c2.code = VFT_NOCODE_;
        c2.vec = c2.w > 0;
#pragma unroll
        for (int k = 0; k < NC; k++) c2.f[k] = A.outF[p * NC + k];
    } else if (jRow) {
        vft_load_row<REAL, NC>(A, j, p, c2);
    } else {
        vft_load_col<REAL, NC>(A, j, p, c2);
    }
}
// does the node's current profile live in a plain row (refinement / ML phases, vft_layout.h)?  Looked up once per pair,
// outside the column loop, so that the loop's loads stay independent of it.
template <typename REAL>
__device__ __forceinline__ bool vft_is_row(const Arena<REAL> &A, int64_t node) {
#ifdef VFT_AB_NO_ROWS   // A/B builds only (tools): the pair kernels as they were before the rows existed
    return false;
#endif
    return node >= A.d.nSeqs && A.mlIs != nullptr && A.mlIs[node - A.d.nSeqs] != 0;
}
// (vft_pair_addends - a column's addends to (denom, top), parked in LDS for the in-order sum - lives in vft_device.h: the walk server's unit uses it too)
template <typename REAL, int NC>
__device__ __forceinline__ void vft_pair_wave(const Arena<REAL> &A, int64_t i, int64_t j, bool jIsOut, double *sW,
                                              double *sT, REAL &dist, REAL &weight) {
    const int lane = threadIdx.x & 63;
    const int64_t nPos = A.d.nPos;
    const bool leaves = !jIsOut && i < A.d.nSeqs && j < A.d.nSeqs;
    if (leaves && !A.dmDist && NC == 4) {   // seqDist, integer counts: any order
        int nUse = 0, nSame = 0;
        for (int c = lane; c < A.d.nChunk; c += 64)
            vft_seq_counts(A.leafT[vft_leaf_idx(A.d, i >> 6, c, (int) (i & 63))],
                           A.leafT[vft_leaf_idx(A.d, j >> 6, c, (int) (j & 63))], nUse, nSame);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            nUse += __shfl_xor(nUse, off, 64);
            nSame += __shfl_xor(nSame, off, 64);
        }
        weight = (REAL) (double) nUse;
        dist = (REAL) (nUse > 0 ? (double) (nUse - nSame) / (double) nUse : 1.0);
        return;
    }
    // two columns per lane and trip: both columns' loads are issued before the first is consumed (the loop is a
    // chain of dependent memory latencies otherwise: mask -> offsets -> stream)
    const bool iRow = vft_is_row<REAL>(A, i), jRow = !jIsOut && vft_is_row<REAL>(A, j);
    for (int64_t p = lane; p < nPos; p += 128) {
        const int64_t pb = p + 64;
        const bool hasB = pb < nPos;
        Col<REAL, NC> a1, a2, b1, b2;
        vft_pair_load<REAL, NC>(A, i, j, jIsOut, p, a1, a2, iRow, jRow);
        if (hasB) vft_pair_load<REAL, NC>(A, i, j, jIsOut, pb, b1, b2, iRow, jRow);
        vft_pair_addends<REAL, NC>(A, leaves, jIsOut, p, a1, a2, sW, sT);
        if (hasB) vft_pair_addends<REAL, NC>(A, leaves, jIsOut, pb, b1, b2, sW, sT);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // lane 0 adds the `top` terms, lane 1 the `denom` terms: two independent chains, each in column order
    // (the adds are one dependent chain, the LDS reads are not: 8 of them are in flight per group)
    double acc = 0;
    if (lane < 2) {
        const double *src = lane == 0 ? sT : sW;
        int64_t p = 0;
        for (; p + 8 <= nPos; p += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = src[p + u];
#pragma unroll
            for (int u = 0; u < 8; u++) acc += v[u];
        }
        for (; p < nPos; p++) acc += src[p];
    }
    const double top = __shfl(acc, 0, 64);
    const double denom = __shfl(acc, 1, 64);
    if (leaves) {
        weight = (REAL) denom;   // nUse
        dist = (REAL) (denom > 0 ? top / denom : 1.0);
    } else {
        weight = (REAL) (denom > 0 ? denom : 0.01);
        dist = (REAL) (denom > 0 ? top / denom : 1.0);
    }
}

// The second half of vft_pair_block: the addends of all columns are in sW / sT; every thread of the workgroup must call.
template <typename REAL>
__device__ __forceinline__ void vft_pair_block_sum(int64_t nPos, bool leaves, const double *sW, const double *sT, REAL &dist, REAL &weight) {
    __shared__ double res[2];
    __syncthreads();
    if (threadIdx.x < 2) {   // thread 0: `top`, thread 1: `denom`, each in column order
        const double *src = threadIdx.x == 0 ? sT : sW;
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
    const double top = res[0], denom = res[1];
    if (leaves) {
        weight = (REAL) denom;   // nUse
        dist = (REAL) (denom > 0 ? top / denom : 1.0);
    } else {
        weight = (REAL) (denom > 0 ? denom : 0.01);
        dist = (REAL) (denom > 0 ? top / denom : 1.0);
    }
}

// The same for ONE pair handled by a whole workgroup (single out-distances and self distances of the join loop: the
// wave version spends ~nPos/64 dependent memory round trips on them, this one nPos/blockDim.x).  Every thread of the
// workgroup must call; sW / sT: nPosPad doubles each; the results are broadcast.
template <typename REAL, int NC, typename DM>
__device__ __forceinline__ void vft_pair_block_t(const Arena<REAL> &A, int64_t i, int64_t j, bool jIsOut, double *sW,
                                               double *sT, REAL &dist, REAL &weight, bool rowsById, const DM &T) {
    const int64_t nPos = A.d.nPos;
    const bool leaves = !jIsOut && i < A.d.nSeqs && j < A.d.nSeqs;
    // two columns per thread and trip, loads of both issued before the first is consumed (as in vft_pair_wave)
    // (rowsById: the caller knows that every internal node has its row - the join engine, whose joins write rows - and spares
    //  the flag's memory round trip in front of the column loads)
    const bool iRow = rowsById ? i >= A.d.nSeqs : vft_is_row<REAL>(A, i), jRow = !jIsOut && (rowsById ? j >= A.d.nSeqs : vft_is_row<REAL>(A, j));
    for (int64_t p = threadIdx.x; p < nPos; p += 2 * (int64_t) blockDim.x) {
        const int64_t pb = p + blockDim.x;
        const bool hasB = pb < nPos;
        Col<REAL, NC> a1, a2, b1, b2;
        vft_pair_load<REAL, NC>(A, i, j, jIsOut, p, a1, a2, iRow, jRow);
        if (hasB) vft_pair_load<REAL, NC>(A, i, j, jIsOut, pb, b1, b2, iRow, jRow);
        vft_pair_addends<REAL, NC, DM>(A, leaves, jIsOut, p, a1, a2, sW, sT, T);
        if (hasB) vft_pair_addends<REAL, NC, DM>(A, leaves, jIsOut, pb, b1, b2, sW, sT, T);
    }
    vft_pair_block_sum<REAL>(nPos, leaves, sW, sT, dist, weight);
}
template <typename REAL, int NC>
__device__ __forceinline__ void vft_pair_block(const Arena<REAL> &A, int64_t i, int64_t j, bool jIsOut, double *sW,
                                               double *sT, REAL &dist, REAL &weight, bool rowsById = false) {
    vft_pair_block_t<REAL, NC, DmGlobal<REAL>>(A, i, j, jIsOut, sW, sT, dist, weight, rowsById, DmGlobal<REAL>(A));
}

// wave-per-item kernels: a workgroup holds blockDim.x / 64 items (4 by default; the host launches fewer waves per
// workgroup for long alignments so that the staging fits the 160 KB of LDS); dynamic LDS = waves * 2 * nPosPad doubles
#define VFT_PW_WAVES ((int) (blockDim.x >> 6))
__device__ __forceinline__ double *vft_pw_lds(double *base, int64_t nPosPad, int which) {
    return base + ((int64_t) (threadIdx.x >> 6) * 2 + which) * nPosPad;
}

// setOutDistance for a list of nodes, or (ids == nullptr) for every active node of [lo,hi).  Unless s.force is
// set only nodes staler than nDiffAllow are recomputed (setCriterion's lazy refresh, NJ.tcc:1092-1098).
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_WG) void k_out_distances(Arena<REAL> A, const int64_t *ids, int64_t n, SweepArgs s) {
    extern __shared__ __attribute__((aligned(16))) double pwLds[];
    const int64_t t = (int64_t) blockIdx.x * VFT_PW_WAVES + (threadIdx.x >> 6);   // wave-uniform item index
    int64_t v;
    if (ids) {
        if (t >= n) return;
        v = ids[t];
    } else {
        v = s.lo + t;
        if (v >= s.hi || A.parent[v] >= 0) return;
    }
    if (!s.force && !((int64_t) A.nOutActive[v] - s.nActive > s.nDiffAllow)) return;
    if ((int64_t) A.nOutActive[v] == s.nActive) return;   // NJ.tcc:1013-1015
    REAL d, w;
    vft_pair_wave<REAL, NC>(A, v, -1, true, vft_pw_lds(pwLds, A.d.nPosPad, 0), vft_pw_lds(pwLds, A.d.nPosPad, 1), d, w);
    if ((threadIdx.x & 63) != 0) return;
    const REAL od = vft_out_distance<REAL>(d, w, s.nActive, A.selfweight[v], A.selfdist[v], A.diameter[v], s.totdiam);
    A.outDist[v] = od;
    A.nOutActive[v] = (int32_t) s.nActive;
    A.mOutDist[v] = od;
    A.mNOut[v] = (int32_t) s.nActive;
}

// the same for ONE node passed by value (the query of a sweep, the nodes of a join): no id list to ship, and the
// whole workgroup walks the columns
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_WG) void k_out_distance_one(Arena<REAL> A, int64_t v, SweepArgs s) {
    extern __shared__ __attribute__((aligned(16))) double pwLds[];
    if (!s.force && !((int64_t) A.nOutActive[v] - s.nActive > s.nDiffAllow)) return;   // workgroup-uniform
    if ((int64_t) A.nOutActive[v] == s.nActive) return;
    REAL d, w;
    vft_pair_block<REAL, NC>(A, v, -1, true, pwLds, pwLds + A.d.nPosPad, d, w);
    if (threadIdx.x != 0) return;
    const REAL od = vft_out_distance<REAL>(d, w, s.nActive, A.selfweight[v], A.selfdist[v], A.diameter[v], s.totdiam);
    A.outDist[v] = od;
    A.nOutActive[v] = (int32_t) s.nActive;
    A.mOutDist[v] = od;
    A.mNOut[v] = (int32_t) s.nActive;
}

// The state change of one join (NJ.tcc:2904-2909, 3003-3007, 254): children get their parent, the new node its
// diameter and an "unreasonably stale" out-distance.  One 1-thread kernel instead of five range stores.
template <typename REAL>
__global__ void k_join_nodes(Arena<REAL> A, int64_t i, int64_t j, int64_t newnode, REAL diameter, int32_t staleStamp) {
    A.parent[i] = (int32_t) newnode;
    A.parent[j] = (int32_t) newnode;
    A.diameter[newnode] = diameter;
    A.outDist[newnode] = 0;
    A.nOutActive[newnode] = staleStamp;
}

// small stream-ordered state writes (no host synchronisation): dst[first + t] = src[t], with optional mirrors
template <typename T>
__global__ void k_store_range(T *dst, T *mirror, const T *src, int64_t first, int64_t count) {
    const int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    dst[first + t] = src[t];
    if (mirror) mirror[first + t] = src[t];
}

// Generic one-vs-all sweep (any alphabet / matrix): lane per target, query fixed.
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_WG) void k_sweep_generic(Arena<REAL> A, SweepArgs s, SweepOut<REAL> O) {
    const int64_t j = s.lo + (int64_t) blockIdx.x * VFT_WG + threadIdx.x;
    REAL cmin = (REAL) 1e30, cmax = (REAL) -1e30;
    if (j < s.hi) {
        if (A.parent[j] >= 0) {
            O.dist[j] = (REAL) 1e20;
            O.crit[j] = (REAL) 1e20;
            O.weight[j] = 0;
        } else {
            REAL dist, weight;
            vft_pair_generic<REAL, NC>(A, s.query, j, false, dist, weight);
            if (!(s.query < A.d.nSeqs && j < A.d.nSeqs)) {
                const REAL dd = A.diameter[s.query] + A.diameter[j];
                dist = dist - dd;
            }
            const REAL crit = vft_criterion<REAL>(dist, A.outDist[s.query], A.nOutActive[s.query], A.outDist[j],
                                                  A.nOutActive[j], s.nActive);
            O.dist[j] = dist;
            O.weight[j] = weight;
            O.crit[j] = crit;
            cmin = cmax = crit;
        }
    }
    vft_block_minmax<REAL>(cmin, cmax, O.partMin, O.partMax, (int) blockIdx.x);
}

// Generic one-vs-all sweep, wave-cooperative: amino-acid alignments are short lists of wide columns (C5: 37 500
// active targets x 300 columns x 20 codes), which a lane-per-target walk leaves 85 % of the machine idle on and turns
// into ~nPos dependent memory latencies per lane.  Here a WAVE owns a target (grid-stride over [lo, hi)): its lanes
// evaluate the columns' addends in parallel and lane 0 adds them in column order (vft_pair_wave, still bit-exact).
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_WG) void k_sweep_wave(Arena<REAL> A, SweepArgs s, SweepOut<REAL> O) {
    extern __shared__ __attribute__((aligned(16))) double pwLds[];
    const int lane = threadIdx.x & 63;
    REAL cmin = (REAL) 1e30, cmax = (REAL) -1e30;
    const int64_t stride = (int64_t) gridDim.x * VFT_PW_WAVES;
    for (int64_t j = s.lo + (int64_t) blockIdx.x * VFT_PW_WAVES + (threadIdx.x >> 6); j < s.hi; j += stride) {
        if (A.parent[j] >= 0) {   // wave-uniform
            if (lane == 0) {
                O.dist[j] = (REAL) 1e20;
                O.crit[j] = (REAL) 1e20;
                O.weight[j] = 0;
            }
            continue;
        }
        REAL dist, weight;
        vft_pair_wave<REAL, NC>(A, s.query, j, false, vft_pw_lds(pwLds, A.d.nPosPad, 0), vft_pw_lds(pwLds, A.d.nPosPad, 1),
                                dist, weight);
        if (lane == 0) {
            if (!(s.query < A.d.nSeqs && j < A.d.nSeqs)) {
                const REAL dd = A.diameter[s.query] + A.diameter[j];
                dist = dist - dd;
            }
            const REAL crit = vft_criterion<REAL>(dist, A.outDist[s.query], A.nOutActive[s.query], A.outDist[j],
                                                  A.nOutActive[j], s.nActive);
            O.dist[j] = dist;
            O.weight[j] = weight;
            O.crit[j] = crit;
            cmin = crit < cmin ? crit : cmin;
            cmax = crit > cmax ? crit : cmax;
        }
    }
    vft_block_minmax<REAL>(cmin, cmax, O.partMin, O.partMax, (int) blockIdx.x);
}

// Publication of a short list's results to host-mapped memory.  One pair per workgroup writing its three numbers
// straight over PCIe is 3 x n tiny posted writes and n system-scope fences: 35 ns per pair, 70 us of a 96 us call at 2 000
// pairs.  Instead every workgroup stores into a DEVICE staging buffer (stage[0..cap) dist, [cap..2 cap) weight,
// [2 cap..3 cap) criterion), and the last one to arrive - all of its threads - copies the three arrays to the host in
// coalesced wavefront-wide stores, fences once and raises the flag.  Every thread of every workgroup must call.
template <typename REAL>
__device__ __forceinline__ void vft_stage_store(REAL *stage, int64_t cap, int64_t t, REAL d, REAL w, REAL cr) {
    __hip_atomic_store(&stage[t], d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&stage[cap + t], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&stage[2 * cap + t], cr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// refDone / nRef: completion tags of refresh workgroups of the SAME launch (k_pairs_refresh_fused) that the publishing
// workgroup has to see before the host may: a forced refresh that is not an end of any listed pair is awaited by no pair
// workgroup, and the host reads its mirrors as soon as the flag moves.
template <typename REAL>
__device__ __forceinline__ void vft_publish_staged(const REAL *stage, int64_t cap, int64_t n, REAL *dist, REAL *weight, REAL *crit,
                                                   unsigned int *doneCtr, unsigned long long *flag, unsigned long long seq,
                                                   const unsigned int *refDone = nullptr, int64_t nRef = 0) {
    __shared__ int isLast;
    if (threadIdx.x == 0) {
        // The staged results were written by THIS thread with agent-scope atomic stores (vft_stage_store): they go to the
        // memory side of this 8-XCD part, and once they are acknowledged (vmcnt 0) the counter may move.  A release fence
        // here would be an L2 write-back per workgroup - 2 000 of them serialise into ~50 us, which was the whole per-pair
        // cost of a long list.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // two-level completion count: 2 000 read-modify-writes of ONE address serialise at the memory side (~20 ns each);
        // workgroup t counts in slot 1 + (t & 63), the workgroup that completes a slot counts in slot 0
        const int64_t t = blockIdx.x >= (unsigned int) (gridDim.x - n) ? (int64_t) blockIdx.x - ((int64_t) gridDim.x - n) : 0;
        const unsigned int sub = (unsigned int) (t & 63);
        const unsigned int inSub = (unsigned int) ((n - sub + 63) / 64), nSubs = (unsigned int) (n < 64 ? n : 64);
        int last = 0;
        if (__hip_atomic_fetch_add(&doneCtr[1 + sub], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == inSub - 1) {
            __hip_atomic_store(&doneCtr[1 + sub], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = __hip_atomic_fetch_add(&doneCtr[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nSubs - 1;
        }
        isLast = last;
    }
    __syncthreads();
    if (!isLast) return;
    for (int64_t b = threadIdx.x; b < nRef; b += blockDim.x)   // every refresh workgroup of this launch has published
        while (__hip_atomic_load(&refDone[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned int) seq) __builtin_amdgcn_s_sleep(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __threadfence();       // (one fence, in the one workgroup that goes on)
    for (int64_t t = threadIdx.x; t < n; t += blockDim.x) {
        dist[t] = __hip_atomic_load(&stage[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        weight[t] = __hip_atomic_load(&stage[cap + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (crit) crit[t] = __hip_atomic_load(&stage[2 * cap + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // The results must have LEFT the chip before the flag moves.  The explicit s_waitcnt is not redundant: ROCm 7.2 on
    // gfx950 can drop the wait that belongs to the fence's write-back when a returned atomic follows (MI355X_MICROARCH.md,
    // "Compiler hazard").
    __threadfence_system();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        *doneCtr = 0;   // launches on this stream are ordered: the next list starts from zero
        __threadfence_system();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// setDistCriterion over an explicit pair list whose out-distances k_refresh_list has brought up to date: distance and
// criterion per pair, results straight into (mapped) memory; the last wave to finish publishes `seq` to the host's
// flag, which replaces a trailing signal kernel (the small lists of the join loop are latency-bound: every launch
// saved is ~6 us of a ~50 us call).
// WGPAIR: one workgroup per pair instead of one wave (short lists: halves the dependent round trips per pair again)
template <typename REAL, int NC, bool WGPAIR>
__global__ __launch_bounds__(VFT_WG) void k_pairs_fused(Arena<REAL> A, const int64_t *pi, const int64_t *pj, int64_t n,
                                                        SweepArgs s, REAL *dist, REAL *weight, REAL *crit,
                                                        unsigned int *doneCtr, unsigned long long *flag,
                                                        unsigned long long seq, REAL *stage, int64_t stageCap) {
    extern __shared__ __attribute__((aligned(16))) double pwLds[];
    const int64_t t = WGPAIR ? (int64_t) blockIdx.x : (int64_t) blockIdx.x * VFT_PW_WAVES + (threadIdx.x >> 6);
    if (t >= n) return;
    const int64_t i = pi[t], j = pj[t];
    REAL d, w;
    if (WGPAIR) vft_pair_block<REAL, NC>(A, i, j, false, pwLds, pwLds + A.d.nPosPad, d, w);
    else vft_pair_wave<REAL, NC>(A, i, j, false, vft_pw_lds(pwLds, A.d.nPosPad, 0), vft_pw_lds(pwLds, A.d.nPosPad, 1), d, w);
    if (WGPAIR && flag && stage) {   // results through the device staging buffer (vft_publish_staged)
        if (threadIdx.x == 0) {
            REAL cr = 0;
            if (!s.force) {
                if (!(i < A.d.nSeqs && j < A.d.nSeqs)) {
                    const REAL dd = A.diameter[i] + A.diameter[j];
                    d = d - dd;
                }
                cr = vft_criterion<REAL>(d, A.outDist[i], A.nOutActive[i], A.outDist[j], A.nOutActive[j], s.nActive);
            }
            vft_stage_store<REAL>(stage, stageCap, t, d, w, cr);
        }
        vft_publish_staged<REAL>(stage, stageCap, n, dist, weight, s.force ? (REAL *) nullptr : crit, doneCtr, flag, seq);
        return;
    }
    if ((WGPAIR ? threadIdx.x : (threadIdx.x & 63)) != 0) return;
    if (s.force) {   // raw profileDist / seqDist (vft_profile_distances): no diameter correction, no criterion
        dist[t] = d;
        weight[t] = w;
    } else {
        if (!(i < A.d.nSeqs && j < A.d.nSeqs)) {
            const REAL dd = A.diameter[i] + A.diameter[j];
            d = d - dd;
        }
        dist[t] = d;
        weight[t] = w;
        crit[t] = vft_criterion<REAL>(d, A.outDist[i], A.nOutActive[i], A.outDist[j], A.nOutActive[j], s.nActive);
    }
    if (flag) {
        // The results above went to host-mapped memory; they must have LEFT the chip before the counter moves and,
        // for the last item, before the flag does.  The explicit s_waitcnt is not redundant: ROCm 7.2 on gfx950 can drop
        // the wait that belongs to the fence's write-back when a returned atomic follows (MI355X_MICROARCH.md, "Compiler
        // hazard") - observed here as a host that now and then read the previous call's numbers from the ring
        // (one 100 000-taxon tree in two came out different).
        __threadfence_system();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (atomicAdd(doneCtr, 1u) == (unsigned int) (n - 1)) {
            *doneCtr = 0;   // launches on this stream are ordered: the next list starts from zero
            __threadfence_system();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// (a step of a host-driven refinement walk - queued averages + the six distances of a quartet - is the walk server's, vft_kernels_walk.h)

// A short pair list and the out-distance refreshes it needs in ONE launch (the join loop makes three such calls per join
// and each is a host round trip: a second, dependent launch is ~8 us of a ~33 us call).  Workgroups [0, nStale) refresh
// the listed nodes exactly like k_refresh_list (the host put the distinct stale / forced nodes there) and then publish
// refDone[b] = seq, whether they recomputed or found the node fresh; workgroups [nStale, nStale + n) are k_pairs_fused's,
// one pair each.  A pair's distance does not depend on out-distances, only its criterion does: the pair workgroup computes
// the distance first and then - thread 0 - waits for the refresh workgroups of its two ends (wait[2 t], wait[2 t + 1]:
// their positions in the refresh list, or -1).  Refresh workgroups have the lowest block ids, are dispatched first and
// never wait, so the waiting ones always make progress; the host uses this kernel only while the whole grid fits the chip
// a few times over (n + nStale <= 4096).
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_WG) void k_pairs_refresh_fused(Arena<REAL> A, const int64_t *ids, int64_t nStale, int64_t nForced,
                                                                const int64_t *pi, const int64_t *pj, const int32_t *wait, int64_t n,
                                                                SweepArgs s, REAL *dist, REAL *weight, REAL *crit,
                                                                unsigned int *refDone, unsigned int *doneCtr,
                                                                unsigned long long *flag, unsigned long long seq, REAL *stage,
                                                                int64_t stageCap) {
    extern __shared__ __attribute__((aligned(16))) double pwLds[];
    const int64_t b = blockIdx.x;
    const unsigned int tag = (unsigned int) seq;
    if (b < nStale) {
        const int64_t v = ids[b];
        const int64_t st = A.nOutActive[v];
        const bool skip = b < nForced ? st == s.nActive : !(st - s.nActive > s.nDiffAllow);   // workgroup-uniform
        if (!skip) {
            REAL d, w;
            vft_pair_block<REAL, NC>(A, v, -1, true, pwLds, pwLds + A.d.nPosPad, d, w);
            if (threadIdx.x == 0) {
                const REAL od = vft_out_distance<REAL>(d, w, s.nActive, A.selfweight[v], A.selfdist[v], A.diameter[v], s.totdiam);
                // agent-scope atomic stores: they pair with the pair workgroups' agent-scope atomic loads below
                __hip_atomic_store(&A.outDist[v], od, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&A.nOutActive[v], (int32_t) s.nActive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                A.mOutDist[v] = od;
                A.mNOut[v] = (int32_t) s.nActive;
                __threadfence_system();   // the host-mapped mirrors must be out before any pair result that used them is
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        if (threadIdx.x == 0) __hip_atomic_store(&refDone[b], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int64_t t = b - nStale;
    if (t >= n) return;
    const int64_t i = pi[t], j = pj[t];
    REAL d, w;
    vft_pair_block<REAL, NC>(A, i, j, false, pwLds, pwLds + A.d.nPosPad, d, w);
    if (threadIdx.x == 0) {
        if (!(i < A.d.nSeqs && j < A.d.nSeqs)) {
            const REAL dd = A.diameter[i] + A.diameter[j];
            d = d - dd;
        }
        const int32_t wi = wait[2 * t], wj = wait[2 * t + 1];
        // (relaxed polls: what is read afterwards is read with agent-scope atomic loads, which go to the memory side where the
        //  refresh workgroup's fenced stores already are when its tag appears; an acquire per pair workgroup would be a cache
        //  invalidation per pair of a list in which every pair names the freshly joined node)
        if (wi >= 0)
            while (__hip_atomic_load(&refDone[wi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag) __builtin_amdgcn_s_sleep(2);
        if (wj >= 0)
            while (__hip_atomic_load(&refDone[wj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag) __builtin_amdgcn_s_sleep(2);
        // nothing below may be issued before the polls have returned (compiler and hardware): the loads that follow go to
        // different addresses and are relaxed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (out-distances of ends nobody refreshes in this launch were written by earlier launches)
        const REAL oi = __hip_atomic_load(&A.outDist[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const REAL oj = __hip_atomic_load(&A.outDist[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int32_t si = __hip_atomic_load(&A.nOutActive[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int32_t sj = __hip_atomic_load(&A.nOutActive[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        vft_stage_store<REAL>(stage, stageCap, t, d, w, vft_criterion<REAL>(d, oi, si, oj, sj, s.nActive));
    }
    vft_publish_staged<REAL>(stage, stageCap, n, dist, weight, crit, doneCtr, flag, seq, refDone, nStale);
}

// The cross product of two node lists (any mix of leaves and internal nodes): dist[a * nB + b] = the join distance of
// (idsA[a], idsB[b]), i.e. profileDist / seqDist minus the two diameters (setDistCriterion without the criterion, which is
// host arithmetic once the out-distances are current).  What a top-hits refresh recomputes (NJ.tcc:4477-4515:
// transferBestHits of the new node's 2m best hits to each of its m closest nodes, every transferred pair needing a new
// distance) is exactly such a block, and as a block it needs 3m ids in and m x 2m distances out instead of a list of
// 2m^2 pairs in and three arrays out.  Entries with a negative id or i == j are skipped (their slot is not written; the host gets whatever the scratch held).
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_WG) void k_pairs_block(Arena<REAL> A, const int64_t *idsA, int64_t nA, const int64_t *idsB,
                                                        int64_t nB, REAL *dist) {
    extern __shared__ __attribute__((aligned(16))) double pwLds[];
    const int64_t t = (int64_t) blockIdx.x * VFT_PW_WAVES + (threadIdx.x >> 6);
    if (t >= nA * nB) return;
    const int64_t i = idsA[t / nB], j = idsB[t % nB];
    if (i < 0 || j < 0 || i == j) return;   // wave-uniform
    REAL d, w;
    vft_pair_wave<REAL, NC>(A, i, j, false, vft_pw_lds(pwLds, A.d.nPosPad, 0), vft_pw_lds(pwLds, A.d.nPosPad, 1), d, w);
    if ((threadIdx.x & 63) != 0) return;
    if (!(i < A.d.nSeqs && j < A.d.nSeqs)) {
        const REAL dd = A.diameter[i] + A.diameter[j];
        d = d - dd;
    }
    dist[t] = d;
}

// The same block for 4-state alphabets, lane per pair: a lane owns one node of list B and VFT_PB_A nodes of list A at a time and
// walks the columns in order with VFT_PB_A pairs of double accumulators (top, denom: the reference's own sequence of additions,
// NJ.tcc:1168-1183, one lane per pair as in the sweeps) - B's column is loaded once per VFT_PB_A pairs, A's columns are staged
// in LDS by the workgroup.  The wave-per-pair kernel above re-reads both profiles for every pair and spends 64 lanes on 200
// columns: 2.9 ms for the 1 000 x 2 000 block of a top-hits refresh at a million sequences, 1.45 ns per pair; this one is
// bound by its arithmetic.  Leaf x leaf pairs come out as seqDist does: top = the number of differing columns, denom = the
// number of shared ones, both exact in double, and the same division.  Grid (ceil(nB / VFT_WG), ceil(nA / VFT_PB_A)).
// Dynamic LDS: VFT_PB_A * nPos * ((NC + 1) REALs + 1 int).
#define VFT_PB_A 8
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_WG) void k_pairs_block_tiled(Arena<REAL> A, const int64_t *idsA, int64_t nA, const int64_t *idsB,
                                                              int64_t nB, REAL *dist) {
    extern __shared__ __attribute__((aligned(16))) double pbLds[];
    const int64_t nPos = A.d.nPos;
    REAL *sF = (REAL *) pbLds;                                   // [VFT_PB_A][nPos][NC]
    REAL *sWt = sF + (int64_t) VFT_PB_A * nPos * NC;             // [VFT_PB_A][nPos]
    int32_t *sCode = (int32_t *) (sWt + (int64_t) VFT_PB_A * nPos);   // [VFT_PB_A][nPos]; bit 8: the column holds a vector
    const int64_t a0 = (int64_t) blockIdx.y * VFT_PB_A;
    for (int64_t e = threadIdx.x; e < (int64_t) VFT_PB_A * nPos; e += VFT_WG) {
        const int64_t a = e / nPos, p = e % nPos;
        const int64_t id = a0 + a < nA ? idsA[a0 + a] : -1;
        Col<REAL, NC> c;
        c.w = 0;
        c.code = VFT_NOCODE_;
        c.vec = false;
#pragma unroll
        for (int k = 0; k < NC; k++) c.f[k] = 0;
        if (id >= 0) vft_load_col_ml<REAL, NC>(A, id, p, c);
        sWt[e] = c.w;
        sCode[e] = c.code | (c.vec ? 256 : 0);
#pragma unroll
        for (int k = 0; k < NC; k++) sF[e * NC + k] = c.vec ? c.f[k] : (REAL) 0;
    }
    __syncthreads();
    const int64_t b = (int64_t) blockIdx.x * VFT_WG + threadIdx.x;
    const int64_t jb = b < nB ? idsB[b] : -1;
    if (jb < 0) return;
    double top[VFT_PB_A], den[VFT_PB_A];
#pragma unroll
    for (int a = 0; a < VFT_PB_A; a++) top[a] = den[a] = 0.0;
    for (int64_t p = 0; p < nPos; p++) {
        Col<REAL, NC> cb;
        vft_load_col_ml<REAL, NC>(A, jb, p, cb);
        if (!(cb.w > 0)) continue;
#pragma unroll
        for (int a = 0; a < VFT_PB_A; a++) {
            const int64_t e = (int64_t) a * nPos + p;
            Col<REAL, NC> ca;
            ca.w = sWt[e];
            if (!(ca.w > 0)) continue;
            const int32_t cc = sCode[e];
            ca.code = cc & 255;
            ca.vec = (cc & 256) != 0;
#pragma unroll
            for (int k = 0; k < NC; k++) ca.f[k] = sF[e * NC + k];
            const REAL ww = ca.w * cb.w;
            const double wgt = (double) ww;
            den[a] += wgt;
            top[a] += wgt * vft_piece<REAL, NC>(A, ca, cb, nullptr);
        }
    }
#pragma unroll
    for (int a = 0; a < VFT_PB_A; a++) {
        if (a0 + a >= nA) break;
        const int64_t i = idsA[a0 + a];
        if (i < 0 || i == jb) continue;
        REAL d = (REAL) (den[a] > 0 ? top[a] / den[a] : 1.0);
        if (!(i < A.d.nSeqs && jb < A.d.nSeqs)) {
            const REAL dd = A.diameter[i] + A.diameter[jb];
            d = d - dd;
        }
        dist[(a0 + a) * nB + b] = d;
    }
}

// Lazy out-distance refresh (setCriterion, NJ.tcc:1092-1098) of the DISTINCT stale nodes of a pair list; the host
// builds the list from its stamp mirror (vft_api.hip: pair_distances), the kernel looks at the real stamp again.
// Entries below nForced are unconditional refreshes that travel with the list (vft_pair_distances_refresh).
// WGPAIR: a workgroup per node (short lists), otherwise a wave.
template <typename REAL, int NC, bool WGPAIR>
__global__ __launch_bounds__(VFT_WG) void k_refresh_list(Arena<REAL> A, const int64_t *ids, int64_t n, int64_t nForced, SweepArgs s) {
    extern __shared__ __attribute__((aligned(16))) double pwLds[];
    const int64_t t = WGPAIR ? (int64_t) blockIdx.x : (int64_t) blockIdx.x * VFT_PW_WAVES + (threadIdx.x >> 6);
    if (t >= n) return;
    const int64_t v = ids[t];
    // the first nForced entries: setOutDistance itself (recomputed unless the stamp IS nActive, NJ.tcc:1012-1015); the
    // rest: setCriterion's lazy rule.  Uniform over the wave / workgroup.
    if (t < nForced ? (int64_t) A.nOutActive[v] == s.nActive : !((int64_t) A.nOutActive[v] - s.nActive > s.nDiffAllow)) return;
    REAL d, w;
    if (WGPAIR) vft_pair_block<REAL, NC>(A, v, -1, true, pwLds, pwLds + A.d.nPosPad, d, w);
    else vft_pair_wave<REAL, NC>(A, v, -1, true, vft_pw_lds(pwLds, A.d.nPosPad, 0), vft_pw_lds(pwLds, A.d.nPosPad, 1), d, w);
    if ((WGPAIR ? threadIdx.x : (threadIdx.x & 63)) != 0) return;
    const REAL od = vft_out_distance<REAL>(d, w, s.nActive, A.selfweight[v], A.selfdist[v], A.diameter[v], s.totdiam);
    A.outDist[v] = od;
    A.mOutDist[v] = od;
    A.mNOut[v] = (int32_t) s.nActive;
    A.nOutActive[v] = (int32_t) s.nActive;
}

// ------------------------------------------------------------------------------------------------ leaf blocks
// setDistCriterion for the cross product of two LEAF lists, nucleotides without a distance matrix: the close-neighbour
// transfers of setAllLeafTopHits (NJ.tcc:3957-3992 -> transferBestHits :4580-4613) evaluate every close neighbour of a
// seed against the seed's 2m best hits - up to m x 2m leaf pairs per seed, 2 x 10^9 over a million-sequence run, all of
// them seqDist (NJ.tcc:1601-1612): integer counts of "both present" / "equal", order-free.  A pair list (one wavefront
// per pair) is the wrong shape for that; here a lane owns a leaf of list B and keeps 2 x 16 counters for 16 leaves of
// list A, whose encoded chunks arrive wave-uniformly (scalar loads): 16 bytes of B per lane and chunk serve 16 pairs.
// Grid: (ceil(nB / 64), ceil(nA / 64)); workgroup = 4 wavefronts x 16 leaves of A.  out[a * nB + b].
// Entries whose id is negative (or the pair of a leaf with itself) still get numbers; the caller masks them.
#define VFT_LB_A 16
template <typename REAL>
__global__ __launch_bounds__(VFT_WG) void k_leaf_block(Arena<REAL> A, const int64_t *idsA, int64_t nA, const int64_t *idsB,
                                                       int64_t nB, SweepArgs s, REAL *dist, REAL *weight, REAL *crit) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int64_t tb = (int64_t) blockIdx.x * 64 + lane;
    const int64_t a0 = (int64_t) blockIdx.y * 64 + (int64_t) wave * VFT_LB_A;
    if (a0 >= nA) return;   // wave-uniform
    const int64_t b = tb < nB ? idsB[tb] : -1;
    const bool bOk = b >= 0 && b < A.d.nSeqs;
    const uint4 *bT = A.leafT + vft_leaf_idx(A.d, bOk ? b >> 6 : 0, 0, (int) (bOk ? b & 63 : 0));
    int64_t aId[VFT_LB_A];
    const uint4 *aT[VFT_LB_A];
#pragma unroll
    for (int u = 0; u < VFT_LB_A; u++) {
        int64_t a = a0 + u < nA ? idsA[a0 + u] : -1;   // wave-uniform
        if (a < 0 || a >= A.d.nSeqs) a = -1;
        aId[u] = a;
        aT[u] = A.leafT + vft_leaf_idx(A.d, a >= 0 ? a >> 6 : 0, 0, (int) (a >= 0 ? a & 63 : 0));
    }
    int nUse[VFT_LB_A], nSame[VFT_LB_A];
#pragma unroll
    for (int u = 0; u < VFT_LB_A; u++) nUse[u] = nSame[u] = 0;
    const int nChunk = A.d.nChunk;
    for (int c = 0; c < nChunk; c++) {
        const uint4 vb = bT[(int64_t) c * VFT_TILE];
#pragma unroll
        for (int u = 0; u < VFT_LB_A; u++) {
            typedef const __attribute__((address_space(4))) vft_u4_t *sp_t;
            const vft_u4_t q = *(sp_t) (aT[u] + (int64_t) c * VFT_TILE);   // wave-uniform address: scalar load
            uint4 va;
            va.x = q.x; va.y = q.y; va.z = q.z; va.w = q.w;
            vft_seq_counts(va, vb, nUse[u], nSame[u]);
        }
    }
    if (tb >= nB) return;
    const REAL outB = bOk ? A.outDist[b] : (REAL) 0;
    const int64_t nOutB = bOk ? A.nOutActive[b] : s.nActive;
#pragma unroll
    for (int u = 0; u < VFT_LB_A; u++) {
        if (a0 + u >= nA) break;
        const int64_t a = aId[u];
        REAL d = (REAL) 1e20, w = 0, cr = (REAL) 1e20;
        if (a >= 0 && bOk) {   // seqDist, NJ.tcc:1621-1623, and the criterion of NJ.tcc:1099-1107
            w = (REAL) (double) nUse[u];
            d = (REAL) (nUse[u] > 0 ? (double) (nUse[u] - nSame[u]) / (double) nUse[u] : 1.0);
            cr = vft_criterion<REAL>(d, A.outDist[a], A.nOutActive[a], outB, nOutB, s.nActive);
        }
        const int64_t o = (a0 + u) * nB + tb;
        dist[o] = d;
        weight[o] = w;
        crit[o] = cr;
    }
}

// ------------------------------------------------------------------------------------------------ top-k select
// The reference sorts all N hits of a sweep and keeps the first 2m (NJ.tcc:3810, 4541).  Here the k best are
// selected without sorting N records: criteria are mapped monotonically onto a 50-bit fixed-point value
//     VK(x) = floor((x - min) * 1024 / (max - min) * 2^40)        (monotone in x, equal x -> equal VK)
// and a radix-1024 select runs on VK from the top digit down: one histogram, the digit that holds the k-th
// smallest, a compaction of everything at or below it, and a rank sort of those few candidates by
// (criterion asc, id desc) — the reference's total order.  One round suffices unless more candidates than
// VFT_CAND_CAP share the threshold digit; then the host narrows to that digit and repeats (rare).
// No kernel issues more than one global atomic per workgroup (same-address atomics cost ~12 ns each on this
// chip, MI355X_MICROARCH.md "fanin").
#define VFT_NBINS 1024
#define VFT_DIGIT_BITS 10
#define VFT_VK_FRAC_BITS 40
#define VFT_MAX_LEVEL 4
#define VFT_SEL_WGS 128
#define VFT_CAND_CAP 8192

struct SelectState {
    double lo, scale;            // VK(x) = (x - lo) * scale, scale = 1024 * 2^40 / (max - min) (slightly shrunk)
    unsigned long long prefix;   // digits chosen so far (VK >> (shift + 10) must equal it to be binned)
    unsigned int level;          // 0 .. VFT_MAX_LEVEL
    unsigned int threshBin;      // digits <= threshBin are collected (VFT_NBINS: collect nothing from the range)
    unsigned int nCand;
    unsigned int overflow;
    unsigned int nBelow;         // #values with a smaller digit than threshBin in this round
    unsigned int nThresh;        // #values in threshBin
    unsigned int nIn;            // #values below the prefix (already known to be among the k smallest)
    unsigned int rankDone;       // workgroups of k_select_rank that have stored their hits (the last one publishes)
};

// Everything one seed's selection works on.  The selection kernels take an array of these and pick theirs by
// blockIdx.y: a batch of seeds (vft_sweep_batch) is selected by ONE set of launches and ONE host synchronisation.
struct SelectHeader;
struct SelSlot {
    const void *crit, *dist, *weight;   // the seed's sweep results, indexed by target id
    const void *partMin, *partMax;      // per-workgroup min / max criteria of its sweep
    SelectState *sel;
    unsigned int *slices;               // [VFT_NBINS]: the seed's histogram of the current round (zero between selections)
    uint64_t *candKey;                  // [VFT_CAND_CAP]
    int32_t *candId;
    void *hits;                         // device: the k records
    SelectHeader *hdr, *hostHdr;        // device / host-mapped header
    void *hostHits;                     // host-mapped copy of the k records
    int64_t query;
    int32_t nPart, pad;
};

__device__ __forceinline__ unsigned long long vft_vk(double x, double lo, double scale) {
    const double y = (x - lo) * scale;
    const double top = 1125899906842623.0;   // 2^50 - 1
    return y <= 0.0 ? 0ull : (y >= top ? (unsigned long long) top : (unsigned long long) y);
}
__device__ __forceinline__ unsigned int vft_level_shift(unsigned int level) {
    return (unsigned int) (VFT_VK_FRAC_BITS - VFT_DIGIT_BITS * (int) level);
}

// refinement (rare): descend into the threshold digit
__global__ void k_select_refine(const SelSlot *slots) {
    SelectState *S = slots[blockIdx.y].sel;
    S->nIn += S->nBelow;
    S->prefix = (S->prefix << VFT_DIGIT_BITS) | S->threshBin;
    S->level += 1;
    S->nCand = 0;
    S->overflow = 0;
    S->rankDone = 0;
}

// VFT_SEL_WGS workgroups, each writes its own histogram slice (plain stores).  first: round one - the key range comes from the
// sweep's per-workgroup (min, max) criteria, which EVERY workgroup reduces for itself (min and max are exact: all of them arrive at the
// same two numbers, a few microseconds of L2 reads side by side instead of a one-workgroup launch in front), and workgroup 0 sets the
// selection's state up for the kernels that follow.
template <typename REAL>
__global__ __launch_bounds__(VFT_WG) void k_select_hist(const SelSlot *slots, int64_t lo, int64_t hi, int first) {
    const SelSlot &sl = slots[blockIdx.y];
    const REAL *crit = (const REAL *) sl.crit;
    SelectState *S = sl.sel;
    unsigned int *slices = sl.slices;
    // four copies of the histogram, one per lane modulo 4: the criteria of a sweep crowd into a few digits, and LDS atomics of a
    // wavefront on one address run one after the other
    __shared__ unsigned int lh4[4][VFT_NBINS];
    unsigned int *lh = lh4[threadIdx.x & 3];
    __shared__ double smin[VFT_WG / 64], smax[VFT_WG / 64];
    for (int t = threadIdx.x; t < 4 * VFT_NBINS; t += VFT_WG) lh4[0][t] = 0;
    double vlo, scale;
    unsigned long long prefix;
    unsigned int level;
    if (first) {
        const REAL *partMin = (const REAL *) sl.partMin, *partMax = (const REAL *) sl.partMax;
        double cmin = 1e30, cmax = -1e30;
        for (int t = threadIdx.x; t < sl.nPart; t += VFT_WG) {
            const double a = (double) partMin[t], b = (double) partMax[t];
            cmin = a < cmin ? a : cmin;
            cmax = b > cmax ? b : cmax;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double a = __shfl_xor(cmin, off, 64), b = __shfl_xor(cmax, off, 64);
            cmin = a < cmin ? a : cmin;
            cmax = b > cmax ? b : cmax;
        }
        if ((threadIdx.x & 63) == 0) {
            smin[threadIdx.x >> 6] = cmin;
            smax[threadIdx.x >> 6] = cmax;
        }
        __syncthreads();
        cmin = smin[0];
        cmax = smax[0];
#pragma unroll
        for (int w = 1; w < VFT_WG / 64; w++) {
            cmin = smin[w] < cmin ? smin[w] : cmin;
            cmax = smax[w] > cmax ? smax[w] : cmax;
        }
        const double span = cmax > cmin ? cmax - cmin : 1.0;
        vlo = cmin;
        scale = 1023.9990234375 * 1099511627776.0 / span;   // (1024 - 2^-10) * 2^40 / span: VK < 2^50
        prefix = 0;
        level = 0;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            S->lo = vlo;
            S->scale = scale;
            S->prefix = 0;
            S->level = 0;
            S->nCand = 0;
            S->overflow = 0;
            S->nIn = 0;
            S->rankDone = 0;
        }
    } else {
        __syncthreads();
        vlo = S->lo;
        scale = S->scale;
        prefix = S->prefix;
        level = S->level;
    }
    const unsigned int shift = vft_level_shift(level);
    const int64_t stride = (int64_t) gridDim.x * VFT_WG;
    int64_t j = lo + (int64_t) blockIdx.x * VFT_WG + threadIdx.x;
    for (; j + 3 * stride < hi; j += 4 * stride) {   // four independent loads in flight per thread
        REAL c[4];
#pragma unroll
        for (int u = 0; u < 4; u++) c[u] = crit[j + u * stride];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (c[u] < (REAL) 1e20) {
                const unsigned long long vk = vft_vk((double) c[u], vlo, scale);
                if ((vk >> (shift + VFT_DIGIT_BITS)) == prefix) atomicAdd(&lh[(vk >> shift) & (VFT_NBINS - 1)], 1u);
            }
        }
    }
    for (; j < hi; j += stride) {
        const REAL c = crit[j];
        if (c < (REAL) 1e20) {
            const unsigned long long vk = vft_vk((double) c, vlo, scale);
            if ((vk >> (shift + VFT_DIGIT_BITS)) == prefix) atomicAdd(&lh[(vk >> shift) & (VFT_NBINS - 1)], 1u);
        }
    }
    __syncthreads();
    // into the seed's ONE histogram (round 6; zero on entry: k_select_rank's last workgroup leaves it so): the bins this workgroup touched,
    // one global atomic each - the threshold digit is then found by every workgroup of k_select_collect for itself (4 KB, one scan)
    // instead of by a one-workgroup launch over VFT_SEL_WGS slices in between
    for (int t = threadIdx.x; t < VFT_NBINS; t += VFT_WG) {
        const unsigned int v = lh4[0][t] + lh4[1][t] + lh4[2][t] + lh4[3][t];
        if (v) atomicAdd(&slices[t], v);
    }
}

template <typename REAL>
__global__ __launch_bounds__(VFT_WG) void k_select_collect(const SelSlot *slots, int64_t lo, int64_t hi, unsigned int k) {
    const SelSlot &sl = slots[blockIdx.y];
    const REAL *crit = (const REAL *) sl.crit;
    SelectState *S = sl.sel;
    uint64_t *candKey = sl.candKey;
    int32_t *candId = sl.candId;
    __shared__ unsigned int lcount, lbase;
    __shared__ uint64_t lkey[VFT_CAND_CAP / 8];
    __shared__ int32_t lid[VFT_CAND_CAP / 8];
    if (threadIdx.x == 0) lcount = 0;
    // ---- the digit that holds the need-th smallest value, from the seed's histogram (every workgroup for itself: the same 1 024
    //      numbers, the same answer; workgroup 0 leaves it in the state for a refinement round and the diagnostics)
    __shared__ unsigned int wsum[VFT_WG / 64], sTb, sBelow, sThresh;
    unsigned int tb;
    {
        constexpr int PER = VFT_NBINS / VFT_WG;
        const unsigned int *gh = sl.slices;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        unsigned int v[PER], mine = 0;
#pragma unroll
        for (int u = 0; u < PER; u++) {
            v[u] = gh[threadIdx.x * PER + u];
            mine += v[u];
        }
        unsigned int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        if (threadIdx.x == 0) {
            sTb = VFT_NBINS;   // need == 0: nothing (more) to pick from this range
            sBelow = 0;
            sThresh = 0;
        }
        __syncthreads();
        unsigned int before = incl - mine, total = 0;
#pragma unroll
        for (int w = 0; w < VFT_WG / 64; w++) {
            if (w < wave) before += wsum[w];
            total += wsum[w];
        }
        unsigned int need = k > S->nIn ? k - S->nIn : 0;
        if (need > total) need = total;
        if (need > 0 && before < need && before + mine >= need) {   // exactly one thread
            unsigned int b = before;
#pragma unroll
            for (int u = 0; u < PER; u++) {
                if (b < need && b + v[u] >= need) {
                    sTb = (unsigned int) (threadIdx.x * PER + u);
                    sBelow = b;
                    sThresh = v[u];
                }
                b += v[u];
            }
        }
        __syncthreads();
        tb = sTb;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            S->threshBin = tb;
            S->nBelow = sBelow;
            S->nThresh = sThresh;
        }
    }
    const double vlo = S->lo, scale = S->scale;
    const unsigned long long prefix = S->prefix;
    const unsigned int shift = vft_level_shift(S->level);
    const int64_t stride = (int64_t) gridDim.x * VFT_WG;
    auto take = [&](int64_t j, REAL c) {
        if (!(c < (REAL) 1e20)) return;
        const unsigned long long vk = vft_vk((double) c, vlo, scale);
        const unsigned long long up = vk >> (shift + VFT_DIGIT_BITS);
        if (up < prefix || (up == prefix && tb < VFT_NBINS && ((vk >> shift) & (VFT_NBINS - 1)) <= tb)) {
            const unsigned int slot = atomicAdd(&lcount, 1u);   // LDS atomic
            if (slot < VFT_CAND_CAP / 8) {
                // total-order key of the candidate: criterion ascending (ties resolved by id in the rank sort)
                lkey[slot] = sizeof(REAL) == 4 ? (uint64_t) vft_order_f32((float) c) : vft_order_f64((double) c);
                lid[slot] = (int32_t) j;
            }
        }
    };
    int64_t j = lo + (int64_t) blockIdx.x * VFT_WG + threadIdx.x;
    for (; j + 3 * stride < hi; j += 4 * stride) {   // four independent loads in flight per thread
        REAL c[4];
#pragma unroll
        for (int u = 0; u < 4; u++) c[u] = crit[j + u * stride];
#pragma unroll
        for (int u = 0; u < 4; u++) take(j + u * stride, c[u]);
    }
    for (; j < hi; j += stride) take(j, crit[j]);
    __syncthreads();
    const unsigned int n = lcount;
    if (n > VFT_CAND_CAP / 8) {
        if (threadIdx.x == 0) S->overflow = 1;
        return;
    }
    if (threadIdx.x == 0) lbase = n ? atomicAdd(&S->nCand, n) : 0;   // one global atomic per workgroup
    __syncthreads();
    const unsigned int base = lbase;
    if (base + n > VFT_CAND_CAP) {
        if (threadIdx.x == 0) S->overflow = 1;
        return;
    }
    for (unsigned int t = threadIdx.x; t < n; t += VFT_WG) {
        candKey[base + t] = lkey[t];
        candId[base + t] = lid[t];
    }
}

// What the host reads back after a sweep: one small block, one copy, one synchronisation.
struct SelectHeader {
    unsigned int nCand, overflow, shift, pad;
    long long bestJ;
    long long pad2;
};

// hit records cross workgroups inside k_select_rank: 8-byte agent-scope words (write-through stores, L1-bypassing loads)
template <typename HIT>
__device__ __forceinline__ void vft_hit_publish(HIT *dst, const HIT &h) {
    static_assert(sizeof(HIT) % 8 == 0, "hit records are whole 8-byte words");
    unsigned long long w[sizeof(HIT) / 8];
    __builtin_memcpy(w, &h, sizeof(HIT));
#pragma unroll
    for (unsigned int t = 0; t < sizeof(HIT) / 8; t++)
        __hip_atomic_store((unsigned long long *) dst + t, w[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename HIT>
__device__ __forceinline__ HIT vft_hit_fetch(const HIT *src) {
    unsigned long long w[sizeof(HIT) / 8];
#pragma unroll
    for (unsigned int t = 0; t < sizeof(HIT) / 8; t++)
        w[t] = __hip_atomic_load((const unsigned long long *) src + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    HIT h;
    __builtin_memcpy(&h, w, sizeof(HIT));
    return h;
}

// Rank sort of the candidates: a candidate's position in the (criterion asc, id desc) order is the number of
// candidates that precede it.  n is a few thousand, so the n^2 comparisons are cheap when spread wide: 16 lanes
// share one candidate (each scans 1/16 of the list, staged through LDS), partial ranks are summed with shuffles.
// A single-workgroup bitonic sort of the same list takes ~50 us (80 barriers); this takes a few.
// The LAST workgroup to finish (a counter in the selection's state) also finishes the selection: bestjoin
// (NJ.tcc:3625-3637: strict '<' while scanning ids upwards => the smallest id among the minimal criteria, the query itself
// excluded), header + hits into the host-mapped result block (the host gets its answer without a DMA copy: a 32 KB hipMemcpy
// D2H goes through SDMA here and costs hundreds of microseconds of latency; zero-copy stores over PCIe cost a few) and the
// completion word `seq` in the host header that vft_sweep(_batch) waits for - no kernel behind this one.  Hits travel
// between the workgroups as write-through 8-byte words + a drained counter increment (no release fence: round 2 tried this
// fold with fences - every workgroup's agent-scope release is an L2 write-back on this 8-XCD part - and the kernel doubled).
#define VFT_RANK_TILE 2048
#define VFT_RANK_LANES 16
template <typename REAL, typename HIT>
__global__ __launch_bounds__(VFT_WG) void k_select_rank(const SelSlot *slots, int32_t k, int64_t lo, int64_t hi, long long seq) {
    const SelSlot &sl = slots[blockIdx.y];
    SelectState *S = sl.sel;
    const uint64_t *candKey = sl.candKey;
    const int32_t *candId = sl.candId;
    const REAL *dist = (const REAL *) sl.dist, *weight = (const REAL *) sl.weight, *crit = (const REAL *) sl.crit;
    HIT *hits = (HIT *) sl.hits;
    HIT *hostHits = (HIT *) sl.hostHits;
    __shared__ uint64_t sk[VFT_RANK_TILE];
    __shared__ int32_t si[VFT_RANK_TILE];
    // An overflowed collection (more candidates than the buffer takes: the host narrows the key range and repeats) has counted
    // candidates it never stored - those entries hold whatever the memory held before, and an id read from there indexed the
    // result arrays (a memory access fault once the allocator handed out recycled blocks: the seed batches of a second tree in
    // one process).  Its records are discarded anyway: rank nothing, publish empty records and the overflow flag.
    const unsigned int n = S->overflow ? 0u : (S->nCand < VFT_CAND_CAP ? S->nCand : VFT_CAND_CAP);
    const unsigned int perWg = VFT_WG / VFT_RANK_LANES;
    const unsigned int span = n > (unsigned int) k ? n : (unsigned int) k;
    if (blockIdx.x * perWg >= span) return;   // whole workgroup idle (not counted below)
    const unsigned int cand = blockIdx.x * perWg + threadIdx.x / VFT_RANK_LANES;
    const unsigned int sub = threadIdx.x % VFT_RANK_LANES;
    const bool mine = cand < n;
    const uint64_t myKey = mine ? candKey[cand] : 0;
    const int32_t myId = mine ? candId[cand] : 0;
    unsigned int rank = 0;
    for (unsigned int base = 0; base < n; base += VFT_RANK_TILE) {
        __syncthreads();
        for (unsigned int u = threadIdx.x; u < VFT_RANK_TILE; u += VFT_WG) {
            sk[u] = base + u < n ? candKey[base + u] : ~0ull;
            si[u] = base + u < n ? candId[base + u] : -1;
        }
        __syncthreads();
        const unsigned int lim = n - base < VFT_RANK_TILE ? n - base : VFT_RANK_TILE;
        if (mine) {
#pragma unroll 8
            for (unsigned int u = sub; u < lim; u += VFT_RANK_LANES) {
                const uint64_t ku = sk[u];
                const int32_t iu = si[u];
                // u precedes me: smaller key, or equal key and larger id
                rank += (ku < myKey || (ku == myKey && iu > myId)) ? 1u : 0u;
            }
        }
    }
#pragma unroll
    for (int off = VFT_RANK_LANES / 2; off > 0; off >>= 1) rank += __shfl_xor(rank, off, VFT_RANK_LANES);
    if (sub == 0) {
        if (mine && rank < (unsigned int) k) {
            HIT h;
            h.j = myId;
            h.dist = dist[myId];
            h.weight = weight[myId];
            h.criterion = crit[myId];
            vft_hit_publish<HIT>(hits + rank, h);
        }
        if (cand >= n && cand < (unsigned int) k) {   // fewer candidates than requested: empty records
            HIT h;
            h.j = -1;
            h.dist = (REAL) 1e20;
            h.weight = 0;
            h.criterion = (REAL) 1e20;
            vft_hit_publish<HIT>(hits + cand, h);
        }
    }
    // ---- the last workgroup to get here finishes the selection
    __shared__ unsigned int sLast;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's records have left the chip's caches
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int active = (span + perWg - 1) / perWg;
        sLast = __hip_atomic_fetch_add(&S->rankDone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == active - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!sLast) return;
    const int64_t query = sl.query;
    SelectHeader *hdr = sl.hdr, *hostHdr = sl.hostHdr;
    __shared__ long long sBest;
    __shared__ REAL sCrit;
    __shared__ int sTruncatedTie;
    // the seed's histogram back to zero for the next selection's (or refinement round's) k_select_hist
    for (int t = threadIdx.x; t < VFT_NBINS; t += VFT_WG) sl.slices[t] = 0;
    // The records into the host's block, whole lines per wavefront.  (Measured and dropped in round 6: every workgroup writing its own
    // records there as it ranks them - 2 000 scattered 16-byte writes per seed over PCIe - took the kernel from 40 to 120 us.)
    for (int t0 = 0; t0 < k; t0 += 8 * VFT_WG) {   // (eight records per thread in flight: the fetches are L2 round trips)
        HIT tmp[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int t = t0 + u * VFT_WG + (int) threadIdx.x;
            tmp[u] = vft_hit_fetch<HIT>(hits + (t < k ? t : 0));
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int t = t0 + u * VFT_WG + (int) threadIdx.x;
            if (t < k) hostHits[t] = tmp[u];
        }
    }
    if (threadIdx.x == 0) {
        const unsigned int nn = n < (unsigned int) k ? n : (unsigned int) k;
        long long best = -1;
        REAL bc = (REAL) 1e20;
        unsigned int t = 0;
        for (; t < nn; t++) {
            const HIT h = vft_hit_fetch<HIT>(hits + t);
            const long long j = (long long) h.j;
            if (j == query || j < 0) continue;
            const REAL c = h.criterion;
            if (best < 0) {
                if (!(c < (REAL) 1e20)) break;
                best = j;
                bc = c;
            } else if (c == bc) {
                if (j < best) best = j;
            } else {
                break;
            }
        }
        sBest = best;
        sCrit = bc;
        // the list ended inside the run of minimal criteria and was cut at k: ids below the cut (the order is id
        // descending within a tie) may tie as well
        sTruncatedTie = (best >= 0 && t == nn && nn == (unsigned int) k) ? 1 : 0;
    }
    __syncthreads();
    if (sTruncatedTie) {   // rare: every listed hit ties at the minimum - scan the criteria themselves
        const REAL bc = sCrit;
        long long mineJ = sBest;
        for (int64_t j = lo + threadIdx.x; j < hi; j += VFT_WG)
            if (j != query && crit[j] == bc && j < mineJ) mineJ = j;
        atomicMin((unsigned long long *) &sBest, (unsigned long long) mineJ);
        __syncthreads();
    }
    // the host copies of the records must have LEFT the chip before the completion word moves (the explicit wait: see
    // vft_publish_staged); the header then rides in front of the completion word's own release - no second fence for it
    __threadfence_system();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x != 0) return;
    SelectHeader h;
    h.nCand = S->nCand;
    h.overflow = S->overflow;
    h.shift = S->level;
    h.pad = 0;
    h.bestJ = sBest;
    h.pad2 = 0;
    *hdr = h;
    hostHdr->nCand = h.nCand;
    hostHdr->overflow = h.overflow;
    hostHdr->shift = h.shift;
    hostHdr->pad = 0;
    hostHdr->bestJ = h.bestJ;
    __hip_atomic_store(&hostHdr->pad2, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Merge of per-shard hit lists (multi-GPU: each rank's sorted top-k, all-gathered): rank sort of the n = lists * k
// records by (criterion asc, id desc), empty records (j < 0) last; the first k go to `out` and to the host-mapped
// block.  Same 16-lanes-per-candidate scheme as k_select_rank.
// nSeeds > 1 (blockIdx.y = seed): `all` is [lists][nSeeds][k] - every rank's batch of lists, all-gathered - and the
// merged list of seed s goes to out + s * k.
template <typename REAL, typename HIT>
__global__ __launch_bounds__(VFT_WG) void k_merge_hits(const HIT *all, int32_t n, int32_t k, HIT *out, HIT *hostOut,
                                                       int32_t nSeeds) {
    __shared__ REAL sc[VFT_RANK_TILE];
    __shared__ long long sj[VFT_RANK_TILE];
    const unsigned int seed = blockIdx.y;
    auto rec = [&](unsigned int u) -> const HIT & {   // record u of this seed's n = lists * k candidates
        return all[((size_t) (u / (unsigned int) k) * (unsigned int) nSeeds + seed) * (unsigned int) k + u % (unsigned int) k];
    };
    out += (size_t) seed * k;
    hostOut += (size_t) seed * k;
    const unsigned int perWg = VFT_WG / VFT_RANK_LANES;
    const unsigned int cand = blockIdx.x * perWg + threadIdx.x / VFT_RANK_LANES;
    const unsigned int sub = threadIdx.x % VFT_RANK_LANES;
    const bool mine = cand < (unsigned int) n;
    const REAL myC = mine ? rec(cand).criterion : (REAL) 0;
    const long long myJ = mine ? (long long) rec(cand).j : -1;
    unsigned int rank = 0;
    for (unsigned int base = 0; base < (unsigned int) n; base += VFT_RANK_TILE) {
        __syncthreads();
        for (unsigned int u = threadIdx.x; u < VFT_RANK_TILE; u += VFT_WG) {
            const bool in = base + u < (unsigned int) n;
            sc[u] = in ? rec(base + u).criterion : (REAL) 0;
            sj[u] = in ? (long long) rec(base + u).j : -1;
        }
        __syncthreads();
        const unsigned int lim = (unsigned int) n - base < VFT_RANK_TILE ? (unsigned int) n - base : VFT_RANK_TILE;
        if (mine && myJ >= 0) {
#pragma unroll 8
            for (unsigned int u = sub; u < lim; u += VFT_RANK_LANES) {
                const long long ju = sj[u];
                const REAL cu = sc[u];
                // u precedes me: it is a real record with a smaller criterion, or the same criterion and a larger id
                rank += (ju >= 0 && (cu < myC || (cu == myC && ju > myJ))) ? 1u : 0u;
            }
        }
    }
#pragma unroll
    for (int off = VFT_RANK_LANES / 2; off > 0; off >>= 1) rank += __shfl_xor(rank, off, VFT_RANK_LANES);
    if (sub != 0 || !mine || myJ < 0) return;
    if (rank < (unsigned int) k) {
        out[rank] = rec(cand);
        hostOut[rank] = rec(cand);
    }
}

template <typename REAL, typename HIT>
__global__ void k_fill_empty_hits(HIT *out, HIT *hostOut, int32_t k) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= k) return;
    HIT h;
    h.j = -1;
    h.dist = (REAL) 1e20;
    h.weight = 0;
    h.criterion = (REAL) 1e20;
    out[t] = h;
    hostOut[t] = h;
}
