// A whole round of subtree-prune-regraft moves (SPR, NJ.tcc:6185-6404; traverseSPR with findSPRSteps :1805-1859 and
// unwindSPRStep :1861-1879, one thread, fast flavour) as ONE persistent workgroup.
//
// The reference walks the nodes in a fixed post-order; around each it tries up to four chains of at most maxSPRLength forced
// minimum-evolution NNIs, keeps the best prefix and unwinds the rest.  Every step is tiny - two averages and six profile
// distances over a few hundred to a few thousand columns - and depends on the one before through the tree, so the host-driven
// walk (MLLengths::doSPR) was one host<->device round trip per step: ~47 us each, 300 000 steps for two rounds on 10 000 taxa,
// three quarters of the whole pipeline once the NNI rounds ran as lanes.  Here the tree (parent / child arrays, the up-profile
// cache flags) lives in device memory and wave 0 of the workgroup IS the walk: it runs the reference's control flow, and hands the
// column work to the whole workgroup as commands through LDS (one command = the averages queued since the last one + the six
// distances of the quartet they lead up to) -
//     AVERAGES  a chain of unweighted averageProfile calls (recomputeProfile, up-profiles): column-parallel, a thread takes its
//               columns through the whole chain (as k_average_chain);
//     DISTANCES the six profile distances of a quartet: every thread forms the addends of its columns for all six pairs, twelve
//               lanes add them in column order (the sums of vft_pair_wave / vft_pair_block, bit-identical to every other pair
//               kernel); wave 0 log-corrects them with glibc's log (vft_glibc_log.h) as the host did.
// A step costs a few barriers and L2 hits instead of a PCIe round trip.  Same operations on the same values in the same order
// as the host walk: the trees stay byte-identical (tests/test_gpu_nni.py, tests/test_gpu_threads.py).
#ifndef VFT_KERNELS_SPR_H
#define VFT_KERNELS_SPR_H

#define VFT_SPR_WG 512            // 8 wavefronts: wave 0 walks, all of them work on the commands
#define VFT_SPR_CHAIN 16          // averages per command
#define VFT_SPR_MAXLEN 16         // chain length the kernel is built for (the reference's default maxSPRLength is 10)
#define VFT_SPR_NCACHE 1024       // entries of wave 0's node-record and cache-flag caches in LDS (direct mapped)
#define VFT_SPR_MISS 64           // missing up-profiles remembered per walk to the root

struct SprNode {                  // one 16-byte load tells everything about a node
    int32_t parent, c0, c1, c2;   // -1 = none (c2: the root's third child)
};

struct SprState {
    SprNode *nodes;               // [nNodes]
    uint32_t *upEpoch;            // [nNodes]: the node's up-profile (slot node + nSeqs) is cached iff == the current epoch
    const int32_t *nodeList;      // the walk, fixed before anything moves
    int32_t *path;                // [nNodes] scratch of getUpProfile's way down from the root
    int64_t nList, nNodes, root;
    int32_t scoredist, maxLen, rowsById, pad;
    double tol;
    int64_t *out;                 // [0] accepted moves, [1] chain steps evaluated, [2] average ops, [3] last epoch,
                                  // [4] / [5] clock ticks (100 MHz) inside commands / in all, [6] commands
};

struct SprCmd {
    int32_t type;                 // 0: stop, 1: work
    int32_t n;                    // averages, in order
    int32_t hasDist, pad;
    int32_t out[VFT_SPR_CHAIN], a[VFT_SPR_CHAIN], b[VFT_SPR_CHAIN];
    // where an input comes from: 0 memory, untouched by this command (requested ahead of the chain); 1 the previous op's output
    // (handed over in registers); 2 an earlier op's output (read back after that op's store - by the thread that stored it)
    uint8_t ka[VFT_SPR_CHAIN], kb[VFT_SPR_CHAIN];
    int32_t q[4];                 // A, B, C, D of the distances
    uint8_t kq[4];                // 0: untouched by this command's averages, 2: written by them
    double dist[6];
    double sum[12];
    // wave 0's chain bookkeeping (LDS instead of scratch: the arrays are indexed at run time)
    int32_t stepA[VFT_SPR_MAXLEN], stepB[VFT_SPR_MAXLEN];
    double stepDelta[VFT_SPR_MAXLEN];
    int32_t miss[VFT_SPR_MISS];
    long long tick[6];            // thread 0's clock ticks per phase of a command (diagnostics): columns, barrier, sums, barrier, end
};

// One column of a node for the SPR walk: a leaf's code or an internal id's plain row.  Every internal id (>= nSeqs: nodes and
// up-profile slots) IS a row in the refinement phase once vft_set_profile_rows has copied the tree over (the host checks), so
// neither the row flag nor the "is this a vector?" test stands between the thread and its loads: weight, code and vector are
// requested together (one memory round instead of three; the vector is simply ignored under a code).  Deliberately small: the
// generic loader with its tile-stream decoding, inlined at every use, made the command loop 47 KB of code - more than the
// instruction cache two CUs share - and every command streamed it from L2 again.
#define VFT_GLOBAL __attribute__((address_space(1)))
#define VFT_LDS __attribute__((address_space(3)))
// the four arrays a column comes from, as GLOBAL pointers: inside a real function the arena's pointers are plain (flat) pointers,
// and a flat load counts on the LDS counter as well - every wait for an LDS read (the command's ids) then waits for all the
// memory loads in flight before it
typedef unsigned int vft_u32x4 __attribute__((ext_vector_type(4)));
template <typename REAL>
struct SprRows {
    const VFT_GLOBAL vft_u32x4 *leafT;
    __device__ __forceinline__ uint4 leaf(int64_t i) const {
        const vft_u32x4 v = leafT[i];
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    VFT_GLOBAL REAL *mlW, *mlF;
    VFT_GLOBAL uint8_t *mlC;
    VftDims d;
};
template <typename REAL, int NC>
__device__ __forceinline__ void vft_spr_load(const SprRows<REAL> &R, int64_t node, int64_t p, Col<REAL, NC> &c) {
    if (node >= R.d.nSeqs) {
        const int64_t idx = (node - R.d.nSeqs) * R.d.nPos + p;
        const VFT_GLOBAL REAL *src = R.mlF + idx * NC;
#pragma unroll
        for (int k = 0; k < NC; k++) c.f[k] = src[k];
        c.w = R.mlW[idx];
        c.code = (int) R.mlC[idx];
        c.vec = c.w > 0 && c.code == VFT_NOCODE_;
    } else {
        const uint4 t = R.leaf(vft_leaf_idx(R.d, node >> 6, (int) (p >> 4), (int) (node & (VFT_TILE - 1))));
        c.code = vft_decode<NC>(vft_byte(t, (int) (p & 15)));
        c.w = c.code != VFT_NOCODE_ ? (REAL) 1 : (REAL) 0;
        c.vec = false;
    }
}

// vft_pair_addends (vft_kernels_nj.h) with the two staging arrays typed as LDS: the addends of one pair at one column
template <typename REAL, int NC>
__device__ __forceinline__ void vft_spr_addends(const Arena<REAL> &A, bool leaves, int64_t p, const Col<REAL, NC> &c1, const Col<REAL, NC> &c2,
                                                VFT_LDS double *sW, VFT_LDS double *sT) {
    double wgt = 0.0, term = 0.0;
    if (leaves) {   // seqDist with a distance matrix (NJ.tcc:1614-1620): top += distances[c1][c2], in order
        if (c1.code != VFT_NOCODE_ && c2.code != VFT_NOCODE_) {
            wgt = 1.0;
            term = A.dmDist ? (double) A.dmDist[c1.code * NC + c2.code] : (c1.code != c2.code ? 1.0 : 0.0);
        }
    } else if (c1.w > 0 && c2.w > 0) {
        const REAL ww = c1.w * c2.w;
        wgt = (double) ww;
        term = wgt * vft_piece<REAL, NC>(A, c1, c2, nullptr);
    }
    sW[p] = wgt;
    sT[p] = term;
}

// the column-ordered sum of n doubles in LDS (n a multiple of 16; the tail beyond the alignment holds +0.0): the adds are one
// dependent chain, so the reads of the next sixteen are in flight while the current sixteen are added
__device__ __forceinline__ double vft_spr_chain_sum(const VFT_LDS double *src, int64_t n) {
    double acc = 0, b0[16], b1[16];
#pragma unroll
    for (int u = 0; u < 16; u++) b0[u] = src[u];
    int64_t p = 16;
    for (; p + 16 <= n; p += 32) {
#pragma unroll
        for (int u = 0; u < 16; u++) b1[u] = src[p + u];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 16; u++) acc += b0[u];
        __builtin_amdgcn_sched_barrier(0);
        if (p + 32 <= n) {
#pragma unroll
            for (int u = 0; u < 16; u++) b0[u] = src[p + 16 + u];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 16; u++) acc += b1[u];
        __builtin_amdgcn_sched_barrier(0);
        if (p + 32 > n) return acc;   // (b0 was not refilled: everything is added)
    }
#pragma unroll
    for (int u = 0; u < 16; u++) acc += b0[u];   // the last sixteen when n / 16 is odd (or n == 16)
    return acc;
}

// the data half of a command, executed by every thread of the workgroup (one copy in the binary: the walk calls it from
// several places)
// Its arguments are LDS objects and say so in their types: handed over as plain pointers they become FLAT accesses - every
// cmd.a[k] in front of a load, every addend parked for the sums, every double the twelve summing lanes read went through the
// vector-memory path instead of a ds_read (the in-order sum of 1 000 doubles took 13-20 us instead of 3).
template <typename REAL, int NC>
__device__ __noinline__ void vft_spr_exec(const VFT_LDS Arena<REAL> *Ap, const VFT_LDS SprState *Sp, VFT_LDS SprCmd *cmdp, VFT_LDS double *pwp) {
    const Arena<REAL> A = *(const Arena<REAL> *) Ap;   // (fields in registers; the casts are resolved back to LDS reads)
    const double tolS = ((const SprState *) Sp)->tol;
    VFT_LDS double *pwLds = pwp;
    SprRows<REAL> R;
    R.leafT = (const VFT_GLOBAL vft_u32x4 *) A.leafT;
    R.mlW = (VFT_GLOBAL REAL *) A.mlW;
    R.mlF = (VFT_GLOBAL REAL *) A.mlF;
    R.mlC = (VFT_GLOBAL uint8_t *) A.mlC;
    R.d = A.d;
    constexpr int PF = NC == 4 ? 2 : 1;   // ops whose memory inputs are requested together
    const int n = cmdp->n;
    const bool hasDist = cmdp->hasDist != 0;
    const int64_t nPos = A.d.nPos, nPosPad = A.d.nPosPad;
    int64_t id[4];
    bool leaf[4];
#pragma unroll
    for (int x = 0; x < 4; x++) {
        id[x] = hasDist ? cmdp->q[x] : 0;
        leaf[x] = id[x] < A.d.nSeqs;
    }
    const bool counts = !A.dmDist && NC == 4;   // leaf pairs by integer counts (seqDist, any order)
    long long tk0 = wall_clock64();
    for (int64_t p = threadIdx.x; p < nPos; p += VFT_SPR_WG) {
        // the quartet's columns that this command does not write: on their way before the averages start
        // (4-state columns only: four 20-state columns next to the averages' operands do not fit the registers)
        constexpr int NQ = NC == 4 ? 4 : 1;
        Col<REAL, NC> cq[NQ];
        if constexpr (NC == 4) {
            if (hasDist) {
#pragma unroll
                for (int x = 0; x < 4; x++)
                    if (cmdp->kq[x] == 0) vft_spr_load<REAL, NC>(R, id[x], p, cq[x]);
            }
        }
        // ---- the averages, in order; a thread takes its column through the whole chain
        Col<REAL, NC> prev;
        prev.w = 0;
        prev.code = VFT_NOCODE_;
        prev.vec = false;
#pragma unroll
        for (int q = 0; q < NC; q++) prev.f[q] = 0;
        for (int g0 = 0; g0 < n; g0 += PF) {
            Col<REAL, NC> ea[PF], eb[PF];
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const int k = g0 + u;
                if (k < n) {
                    if (cmdp->ka[k] == 0) vft_spr_load<REAL, NC>(R, cmdp->a[k], p, ea[u]);
                    if (cmdp->kb[k] == 0) vft_spr_load<REAL, NC>(R, cmdp->b[k], p, eb[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const int k = g0 + u;
                if (k < n) {
                    Col<REAL, NC> c1, c2;
                    const int ka = cmdp->ka[k], kb = cmdp->kb[k];
                    if (ka == 0) c1 = ea[u];
                    else if (ka == 1) c1 = prev;
                    else vft_spr_load<REAL, NC>(R, cmdp->a[k], p, c1);
                    if (kb == 0) c2 = eb[u];
                    else if (kb == 1) c2 = prev;
                    else vft_spr_load<REAL, NC>(R, cmdp->b[k], p, c2);
                    REAL wo, f[NC];
                    int co;
                    vft_average_col<REAL, NC>(A, c1, c2, 0.5, tolS, wo, co, f);
                    {   // the row's vector slot is written whatever the column holds: the loads above never test before they read
                        const int64_t idx = ((int64_t) cmdp->out[k] - A.d.nSeqs) * nPos + p;
                        R.mlW[idx] = wo;
                        R.mlC[idx] = (uint8_t) co;
                        VFT_GLOBAL REAL *dst = R.mlF + idx * NC;
#pragma unroll
                        for (int q = 0; q < NC; q++) dst[q] = f[q];
                    }
                    prev.w = wo;
                    prev.code = co;
                    prev.vec = wo > 0 && co == VFT_NOCODE_;
#pragma unroll
                    for (int q = 0; q < NC; q++) prev.f[q] = f[q];
                }
            }
        }
        // ---- the addends of the six pairs AB AC AD BC BD CD at this column (chooseNNI, NJ.tcc:4836-4846): exactly what the
        // pair kernels park in LDS (vft_pair_addends); members written above are read back by the thread that wrote them
        if (hasDist) {
            if constexpr (NC == 4) {
#pragma unroll
                for (int x = 0; x < 4; x++)
                    if (cmdp->kq[x] != 0) vft_spr_load<REAL, NC>(R, id[x], p, cq[x]);
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const int x = k < 3 ? 0 : k < 5 ? 1 : 2, y = k == 0 ? 1 : (k == 1 || k == 3) ? 2 : 3;
                    const bool ll = leaf[x] && leaf[y];
                    if (ll && counts) continue;
                    vft_spr_addends<REAL, NC>(A, ll, p, cq[x], cq[y], pwLds + (2 * k) * (int) nPosPad, pwLds + (2 * k + 1) * (int) nPosPad);
                }
            } else {
                Col<REAL, NC> c4[4];   // all four requested at once (one memory round), then the six pairs
#pragma unroll
                for (int x = 0; x < 4; x++) vft_spr_load<REAL, NC>(R, id[x], p, c4[x]);
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const int x = k < 3 ? 0 : k < 5 ? 1 : 2, y = k == 0 ? 1 : (k == 1 || k == 3) ? 2 : 3;
                    vft_spr_addends<REAL, NC>(A, leaf[x] && leaf[y], p, c4[x], c4[y], pwLds + (2 * k) * (int) nPosPad, pwLds + (2 * k + 1) * (int) nPosPad);
                }
            }
        }
    }
    // every output is a row from now on (the flags of internal nodes are set already in this phase; up-profile slots get
    // theirs here); other kernels read them after this one
    if (threadIdx.x < (unsigned) n && A.mlIs) A.mlIs[cmdp->out[threadIdx.x] - A.d.nSeqs] = 1;
    if (threadIdx.x == 0) {
        const long long t = wall_clock64();
        cmdp->tick[0] += t - tk0;
        tk0 = t;
    }
    if (!hasDist) return;
    {   // leaf x leaf pairs of a matrix-free nucleotide run: integer counts by the pair's wavefront (one round of loads)
        const int w = threadIdx.x >> 6;
        if (w < 6) {
            const int x = w < 3 ? 0 : w < 5 ? 1 : 2, y = w == 0 ? 1 : (w == 1 || w == 3) ? 2 : 3;
            if (leaf[x] && leaf[y] && counts) {   // seqDist (NJ.tcc:1601-1624) as vft_pair_wave counts it
                const int lane = threadIdx.x & 63;
                int nUse = 0, nSame = 0;
                for (int c = lane; c < A.d.nChunk; c += 64)
                    vft_seq_counts(R.leaf(vft_leaf_idx(A.d, id[x] >> 6, c, (int) (id[x] & 63))),
                                   R.leaf(vft_leaf_idx(A.d, id[y] >> 6, c, (int) (id[y] & 63))), nUse, nSame);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    nUse += __shfl_xor(nUse, off, 64);
                    nSame += __shfl_xor(nSame, off, 64);
                }
                const REAL d = (REAL) (nUse > 0 ? (double) (nUse - nSame) / (double) nUse : 1.0);
                if (lane == 0) cmdp->dist[w] = (double) d;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long t = wall_clock64();
        cmdp->tick[1] += t - tk0;
        tk0 = t;
    }
    {   // chain ch = 2 k + which (0: the pair's weights sW -> denom, 1: its terms sT -> top) on wave ch & 7, lane ch >> 3
        const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
        if (l < 2 && w + 8 * l < 12) cmdp->sum[w + 8 * l] = vft_spr_chain_sum(pwLds + (w + 8 * l) * (int) nPosPad, nPosPad);
    }
    if (threadIdx.x == 0) {
        const long long t = wall_clock64();
        cmdp->tick[2] += t - tk0;
        tk0 = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long t = wall_clock64();
        cmdp->tick[3] += t - tk0;
        tk0 = t;
    }
    if (threadIdx.x < 6) {
        const int w = threadIdx.x;
        const int x = w < 3 ? 0 : w < 5 ? 1 : 2, y = w == 0 ? 1 : (w == 1 || w == 3) ? 2 : 3;
        if (!(leaf[x] && leaf[y] && counts)) {
            const double denom = cmdp->sum[2 * w], top = cmdp->sum[2 * w + 1];
            const REAL d = (REAL) (denom > 0 ? top / denom : 1.0);
            cmdp->dist[w] = (double) d;
        }
    }
}

// wave 0's state: plain scalars handed to small helpers by reference (everything is inlined; nothing is indexed at run time)
struct SprWalk {
    uint32_t epoch;
    int nQueued;
    long long nAvg, tCmd, nCmd;
};

template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_SPR_WG) void k_spr_walk(Arena<REAL> A_, SprState S_) {
    extern __shared__ __attribute__((aligned(16))) double pwLds[];
    __shared__ SprCmd cmd;
    // The arena and the walk's state as LDS objects: vft_spr_exec is ONE function called from several places (not inlined), and
    // a reference to a kernel argument handed to a real function makes the compiler keep a copy of the argument in scratch
    // memory - every A.mlW, A.d.nPos in the column loops was then a trip to private memory in front of the load it feeds.
    // A reference to an LDS object costs an LDS read.
    __shared__ Arena<REAL> sArena;
    __shared__ SprState sState;
    // wave 0's view of the tree: the records and up-profile flags of recently touched nodes (a step works on a dozen nodes and
    // walks one path to the root; from L2 every one of those dependent reads is a memory round trip)
    __shared__ int4 ncache[VFT_SPR_NCACHE];
    __shared__ int32_t ntag[VFT_SPR_NCACHE], etag[VFT_SPR_NCACHE];
    __shared__ uint32_t eval[VFT_SPR_NCACHE];
    if (threadIdx.x == 0) {
        sArena = A_;
        sState = S_;
    }
    for (int t = threadIdx.x; t < VFT_SPR_NCACHE; t += VFT_SPR_WG) ntag[t] = etag[t] = -1;
    if (threadIdx.x < 6) cmd.tick[threadIdx.x] = 0;
    // the columns between the alignment's end and the staging stride hold +0.0 for good (vft_spr_chain_sum adds whole sixteens)
    for (int64_t t = threadIdx.x; t < 12 * (A_.d.nPosPad - A_.d.nPos); t += VFT_SPR_WG) {
        const int64_t arr = t / (A_.d.nPosPad - A_.d.nPos), off = t % (A_.d.nPosPad - A_.d.nPos);
        pwLds[arr * A_.d.nPosPad + A_.d.nPos + off] = 0.0;
    }
    __syncthreads();
    const Arena<REAL> &A = sArena;
    const SprState &S = sState;
    const int lane = threadIdx.x & 63;
    if ((threadIdx.x >> 6) != 0) {   // workers: wait for a command, do its data half, report back
        for (;;) {
            __syncthreads();
            if (cmd.type == 0) return;
            vft_spr_exec<REAL, NC>((const VFT_LDS Arena<REAL> *) &sArena, (const VFT_LDS SprState *) &sState, (VFT_LDS SprCmd *) &cmd, (VFT_LDS double *) pwLds);
            __syncthreads();
        }
    }
    // ---- wave 0: the walk.  All 64 lanes run it in step (uniform values); stores are lane 0's.
    const int32_t nSeqs = (int32_t) A_.d.nSeqs, root = (int32_t) S_.root;
    SprNode *nodes = S_.nodes;
    uint32_t *upEpoch = S_.upEpoch;
    int32_t *pathBuf = S_.path;
    const int32_t *nodeList = S_.nodeList;
    const int64_t nList = S_.nList;
    const int maxLen = S_.maxLen;
    const bool scoredist = S_.scoredist != 0;
    int64_t *outBuf = S_.out;
    SprWalk W;
    W.epoch = 1;
    W.nQueued = 0;
    W.nAvg = W.tCmd = W.nCmd = 0;
    int64_t nSPR = 0, nSteps = 0;
    const long long tStart = wall_clock64(), cStart = clock64();
    auto sync0 = [&]() {   // wave 0's own stores (lane 0) before its next loads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto ld = [&](int32_t v) -> SprNode {
        const int slot = v & (VFT_SPR_NCACHE - 1);
        int4 r;
        if (ntag[slot] == v) {
            r = ncache[slot];
        } else {
            r = *reinterpret_cast<const int4 *>(&nodes[v]);
            if (lane == 0) {
                ncache[slot] = r;
                ntag[slot] = v;
            }
        }
        SprNode n;
        n.parent = r.x;
        n.c0 = r.y;
        n.c1 = r.z;
        n.c2 = r.w;
        return n;
    };
    auto parentOf = [&](int32_t v) -> int32_t { return ld(v).parent; };
    auto epochOf = [&](int32_t v) -> uint32_t {
        const int slot = v & (VFT_SPR_NCACHE - 1);
        if (etag[slot] == v) return eval[slot];
        const uint32_t e = upEpoch[v];
        if (lane == 0) {
            eval[slot] = e;
            etag[slot] = v;
        }
        return e;
    };
    // hand the command in LDS to the workgroup and take part in it: the averages queued so far, then (withDist) the distances
    auto issue = [&](bool withDist) {
        if (lane == 0) {
            cmd.type = 1;
            cmd.n = W.nQueued;
            cmd.hasDist = withDist ? 1 : 0;
        }
        W.nAvg += W.nQueued;
        W.nQueued = 0;
        W.nCmd++;
        const long long t0 = wall_clock64();
        __syncthreads();
        vft_spr_exec<REAL, NC>((const VFT_LDS Arena<REAL> *) &sArena, (const VFT_LDS SprState *) &sState, (VFT_LDS SprCmd *) &cmd, (VFT_LDS double *) pwLds);
        __syncthreads();
        W.tCmd += wall_clock64() - t0;
    };
    auto producer = [&](int32_t v) -> int {   // 0: not written by the queued ops, 1: by the last one, 2: by an earlier one
        int kind = 0;
        for (int k = 0; k < W.nQueued; k++)
            if (cmd.out[k] == v) kind = k == W.nQueued - 1 ? 1 : 2;
        return kind;
    };
    auto queueAverage = [&](int32_t out, int32_t a, int32_t b) {
        if (W.nQueued == VFT_SPR_CHAIN) issue(false);
        const int ka = producer(a), kb = producer(b);
        if (lane == 0) {
            cmd.out[W.nQueued] = out;
            cmd.a[W.nQueued] = a;
            cmd.b[W.nQueued] = b;
            cmd.ka[W.nQueued] = (uint8_t) ka;
            cmd.kb[W.nQueued] = (uint8_t) kb;
        }
        W.nQueued++;
    };
    // the two children of the root that are not v
    auto rootOthers = [&](const SprNode &R, int32_t v, int32_t &x, int32_t &y) {
        if (R.c0 == v) {
            x = R.c1;
            y = R.c2;
        } else if (R.c1 == v) {
            x = R.c0;
            y = R.c2;
        } else {
            x = R.c0;
            y = R.c1;
        }
    };
    auto setUp = [&](int32_t v, uint32_t e) {
        if (lane == 0) {
            upEpoch[v] = e;
            eval[v & (VFT_SPR_NCACHE - 1)] = e;
            etag[v & (VFT_SPR_NCACHE - 1)] = v;
        }
    };
    auto replaceChild = [&](int32_t par, int32_t oldChild, int32_t newChild) {   // NJ.tcc:1929-1940
        SprNode P = ld(par);
        if (P.c0 == oldChild) P.c0 = newChild;
        else if (P.c1 == oldChild) P.c1 = newChild;
        else if (P.c2 == oldChild) P.c2 = newChild;
        if (lane == 0) {
            const int4 r = make_int4(P.parent, P.c0, P.c1, P.c2);
            *reinterpret_cast<int4 *>(&nodes[par]) = r;
            ncache[par & (VFT_SPR_NCACHE - 1)] = r;   // (ld above left par in its slot)
            nodes[newChild].parent = par;
            const int cs = newChild & (VFT_SPR_NCACHE - 1);
            if (ntag[cs] == newChild) ncache[cs].x = par;
        }
        sync0();
    };
    auto ensureUpProfile = [&](int32_t node) {   // getUpProfile (NJ.tcc:3382-3434): cached; missing ones from the root down
        if (epochOf(node) == W.epoch) return;
        // the way up: parents through the LDS cache; the cache flags of the nodes passed are requested on the way (they are not
        // part of the chase) and the missing ones remembered, deepest first
        int len = 0, nMiss = 0;
        for (int32_t x = node; x != root;) {
            const uint32_t e = epochOf(x);
            const int32_t up = parentOf(x);
            if (lane == 0) pathBuf[len] = x;
            if (e != W.epoch) {
                if (nMiss < VFT_SPR_MISS && lane == 0) cmd.miss[nMiss] = x;
                nMiss++;
            }
            len++;
            x = up;
        }
        sync0();
        auto build = [&](int32_t x) {
            const SprNode X = ld(x);
            const SprNode P = ld(X.parent);
            int32_t c0, c1;
            if (X.parent == root) {
                rootOthers(P, x, c0, c1);
            } else {
                c0 = P.c0 == x ? P.c1 : P.c0;
                c1 = X.parent + nSeqs;
            }
            queueAverage(x + nSeqs, c0, c1);
            setUp(x, W.epoch);
        };
        if (nMiss <= VFT_SPR_MISS) {
            for (int t = nMiss; t-- > 0;) build(cmd.miss[t]);   // from the root down
        } else {
            for (int t = len; t-- > 0;) {
                const int32_t x = pathBuf[t];
                if (epochOf(x) != W.epoch) build(x);
            }
        }
        sync0();
    };
    auto updateForNNI = [&](int32_t node) {   // NJ.tcc:1902-1926, fast flavour
        const SprNode N = ld(node);
        const int32_t ip = N.parent;
        const SprNode P = ld(ip);
        setUp(node, 0);
        setUp(N.c0, 0);
        setUp(N.c1, 0);
        if (ip == root) {
            int32_t x, y;
            rootOthers(P, node, x, y);
            setUp(x, 0);
            setUp(y, 0);
        } else {
            setUp(ip, 0);
            setUp(P.c0 == node ? P.c1 : P.c0, 0);
            if (P.parent != root) {   // the uncle
                const SprNode G = ld(P.parent);
                setUp(G.c0 == ip ? G.c1 : G.c0, 0);
            }
        }
        sync0();
        if (node >= nSeqs && node != root) queueAverage(node, N.c0, N.c1);   // recomputeProfile (NJ.tcc:3436-3473, no BIONJ weights)
        if (ip >= nSeqs && ip != root) queueAverage(ip, P.c0, P.c1);
    };
    auto movePivots = [&](int32_t node, int32_t &o0, int32_t &o1) {
        const SprNode N = ld(node);
        const SprNode P = ld(N.parent);
        if (N.parent == root) {
            rootOthers(P, node, o0, o1);
        } else {
            o0 = N.parent;
            o1 = P.c0 == node ? P.c1 : P.c0;
        }
    };
    auto logCorrect = [&](double dist) -> double {   // NJ.tcc:322-330
        const double maxscore = 3.0;
        if (!scoredist) dist = dist < 0.74 ? -0.75 * vft_glibc_log(1.0 - dist * 4.0 / 3.0) : maxscore;
        else dist = dist < 0.99 ? -1.3 * vft_glibc_log(1.0 - dist) : maxscore;
        return dist < maxscore ? dist : maxscore;
    };
    // chooseNNI's criteria around `node`, lower is better; q0..q3: setupABCD's nodes (NJ.tcc:1942-1975)
    auto meCriteria = [&](int32_t node, int32_t &q0, int32_t &q1, int32_t &q2, int32_t &q3, double &cr0, double &cr1, double &cr2) {
        const SprNode N = ld(node);
        const SprNode P = ld(N.parent);
        q0 = N.c0;
        q1 = N.c1;
        int32_t idD;
        if (N.parent == root) {
            rootOthers(P, node, q2, q3);
            idD = q3;
        } else {
            q2 = P.c0 == node ? P.c1 : P.c0;
            q3 = N.parent;
            ensureUpProfile(N.parent);
            idD = N.parent + nSeqs;
        }
        {
            const int k0 = producer(q0), k1 = producer(q1), k2 = producer(q2), k3 = producer(idD);
            if (lane == 0) {
                cmd.q[0] = q0;
                cmd.q[1] = q1;
                cmd.q[2] = q2;
                cmd.q[3] = idD;
                cmd.kq[0] = (uint8_t) (k0 ? 2 : 0);
                cmd.kq[1] = (uint8_t) (k1 ? 2 : 0);
                cmd.kq[2] = (uint8_t) (k2 ? 2 : 0);
                cmd.kq[3] = (uint8_t) (k3 ? 2 : 0);
            }
        }
        issue(true);
        const double c0 = logCorrect(cmd.dist[0]), c1 = logCorrect(cmd.dist[1]), c2 = logCorrect(cmd.dist[2]), c3 = logCorrect(cmd.dist[3]),
                     c4 = logCorrect(cmd.dist[4]), c5 = logCorrect(cmd.dist[5]);
        cr0 = c0 + c5;
        cr1 = c1 + c4;
        cr2 = c2 + c3;
        nSteps++;
    };

    for (int64_t it = 0; it < nList; it++) {
        const int32_t node = nodeList[it];
        if (node == root) continue;
        int32_t nodeAround0, nodeAround1;
        movePivots(node, nodeAround0, nodeAround1);
        bool bChanged = false;
        for (int iAround = 0; iAround < 2 && !bChanged; iAround++) {
            for (int acFirst = 0; acFirst < 2 && !bChanged; acFirst++) {
                int32_t around = iAround == 0 ? nodeAround0 : nodeAround1;
                int chainLength = 0;
                for (; chainLength < maxLen; chainLength++) {   // findSPRSteps
                    if (around < nSeqs || around == root) break;
                    int32_t q0, q1, q2, q3;
                    double cr0, cr1, cr2;
                    meCriteria(around, q0, q1, q2, q3, cr0, cr1, cr2);
                    int32_t n0, n1;
                    double delta;
                    if (chainLength == 0 ? acFirst != 0 : cr1 < cr2) {
                        delta = cr1 - cr0;   // swap B and C
                        n0 = q1;
                        n1 = q2;
                    } else {
                        delta = cr2 - cr0;   // swap A and C
                        n0 = q0;
                        n1 = q2;
                    }
                    if (lane == 0) {
                        cmd.stepA[chainLength] = n0;
                        cmd.stepB[chainLength] = n1;
                        cmd.stepDelta[chainLength] = delta;
                    }
                    const int32_t parAround = ld(around).parent;
                    replaceChild(around, n0, n1);
                    replaceChild(parAround, n1, n0);
                    updateForNNI(around);
                    int32_t nx0, nx1;
                    movePivots(node, nx0, nx1);
                    around = nx0 == around ? nx1 : nx0;
                }
                double dMinDelta = 0.0, dTotDelta = 0.0;
                int iCBest = -1;
                for (int iC = 0; iC < chainLength; iC++) {
                    dTotDelta += cmd.stepDelta[iC];
                    if (dTotDelta < dMinDelta) {
                        dMinDelta = dTotDelta;
                        iCBest = iC;
                    }
                }
                for (int iC = chainLength - 1; iC > iCBest; iC--) {   // unwindSPRStep
                    const int32_t n0 = cmd.stepA[iC], n1 = cmd.stepB[iC];
                    const int32_t p0 = ld(n0).parent, p1 = ld(n1).parent;
                    replaceChild(p0, n0, n1);
                    replaceChild(p1, n1, n0);
                    updateForNNI(ld(p0).parent == p1 ? p0 : p1);
                }
                if (iCBest >= 0) bChanged = true;
            }
        }
        if (bChanged) {
            nSPR++;
            W.epoch++;   // every cached up-profile is dropped (NJ.tcc:6275-6279)
            for (int32_t anc = ld(node).parent; anc >= 0;) {
                const SprNode X = ld(anc);
                if (anc >= nSeqs && anc != root) queueAverage(anc, X.c0, X.c1);
                anc = X.parent;
            }
        }
    }
    if (W.nQueued) issue(false);
    if (lane == 0) {
        outBuf[0] = nSPR;
        outBuf[1] = nSteps;
        outBuf[2] = W.nAvg;
        outBuf[3] = (int64_t) W.epoch;
        outBuf[4] = W.tCmd;
        outBuf[5] = wall_clock64() - tStart;
        outBuf[6] = W.nCmd;
        outBuf[7] = cmd.tick[0];
        outBuf[8] = cmd.tick[1];
        outBuf[9] = cmd.tick[2];
        outBuf[10] = cmd.tick[3];
        outBuf[11] = clock64() - cStart;   // shader clock cycles of the whole walk
        cmd.type = 0;
    }
    __syncthreads();   // releases the workers
}

#endif
