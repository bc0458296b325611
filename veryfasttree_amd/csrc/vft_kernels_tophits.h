// Top-hit lists in HBM and the per-join list operations of fastNJ's top-hits heuristic as device kernels.
//
// The reference keeps, for every active node, a list of its m = sqrt(N) best join partners (TopHits / TopHitsList / Hit,
// NJ.h:206-248) and walks such lists three times per join: getBestFromTopHits for both ends of the candidate join
// (NJ.tcc:4267-4298) and the merge of the two children's lists for the new node (topHitJoin -> uniqueBestHits ->
// sortSaveBestHits, NJ.tcc:4319-4330, 4786-4833, 4535-4578).  Round 2 walked them on the host: re-target every entry to its
// active ancestor, ship the pairs that need a distance, wait, compute criteria, sort - ~250 us per join at a million
// sequences.  Here the lists live on the device (hits[node][m], 8 bytes per entry in float precision: 16 GB for 2 x 10^6
// nodes x 1000 entries) and each of the three walks is ONE launch that returns only its result:
//
//   k_th_best   one workgroup per list entry: active ancestor of the partner (parent[] is device state), the distance of
//               a re-targeted pair (vft_pair_block: in-order column sums, bit-identical to every other pair kernel), the
//               lazy out-distance refresh of the partner (setCriterion, NJ.tcc:1092-1098), the criterion; the workgroup
//               that finishes last picks the first minimum in list order and publishes (j, dist, criterion).
//   k_th_join   one workgroup per entry of the two children's lists: ancestor, first-come ownership of each distinct
//               partner (an epoch-tagged mark per node), distance against the new node, lazy refresh, criterion; the last
//               workgroup sorts the distinct candidates by (criterion ascending, partner id descending) - the reference's
//               order: psort's tie rule on a list that is in ascending partner order - decides whether the
//               merged list is good enough (NJ.tcc:4342-4362), saves its first entries as the new node's list and
//               publishes the sorted candidates.
//
// Cross-workgroup hand-over follows vft_publish_staged (vft_kernels_nj.h): results are staged with agent-scope atomic
// stores, acknowledged (s_waitcnt) before a two-level completion count, and read back with agent-scope atomic loads by the
// one workgroup that completes the count.  No workgroup ever waits for another one.
#pragma once
#include "vft_kernels_nj.h"

template <typename REAL>
struct ThHit {          // Hit, NJ.h:206-209 (node ids are below 2^31)
    int32_t j;
    REAL dist;
};

template <typename REAL>
struct TopHits {
    ThHit<REAL> *hits;      // [nLists][m]
    int32_t *len;           // [nLists]
    int32_t m;
    int32_t cap;            // entries of the staging arrays (>= 2 m)
    int64_t nLists;
    int32_t *stJ;           // staging of one call: partner (or -1), distance, criterion per entry
    REAL *stD, *stC;
    unsigned int *mark;     // [maxNodes]: tag of the last k_th_join call that claimed the node as a candidate
    unsigned int *doneCtr;  // [65] two-level completion count
    int32_t *sorted;        // [cap + 1] join engine: staging indices of the candidates in sorted order; [cap] = how many
};

struct ThBestOut {      // what getBestFromTopHits returns (host-mapped)
    int32_t j, pos;
    double dist, crit;  // numeric_t values, widened
};

struct ThJoinInfo {     // header of k_th_join's host-mapped result block; the sorted (j, dist, criterion) arrays follow
    int32_t nUnique, useUnique, nSave, pad;
};

__device__ __forceinline__ int32_t vft_active_ancestor(const int32_t *parent, int32_t v) {   // NJ.tcc:536-544
    if (v < 0) return v;
    for (;;) {
        const int32_t p = parent[v];
        if (p < 0) return v;
        v = p;
    }
}

// Every thread of every workgroup calls; true in the workgroup that arrives last.  The caller's staged stores (thread 0,
// agent-scope atomics) are acknowledged before the count moves.
__device__ __forceinline__ bool vft_th_arrive(unsigned int *doneCtr) {
    __shared__ int thIsLast;
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int n = gridDim.x, sub = blockIdx.x & 63u;
        const unsigned int inSub = (n - sub + 63u) / 64u, nSubs = n < 64u ? n : 64u;
        int last = 0;
        if (__hip_atomic_fetch_add(&doneCtr[1 + sub], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == inSub - 1) {
            __hip_atomic_store(&doneCtr[1 + sub], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = __hip_atomic_fetch_add(&doneCtr[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nSubs - 1;
            if (last) __hip_atomic_store(&doneCtr[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // launches are stream-ordered
        }
        thIsLast = last;
    }
    __syncthreads();
    return thIsLast != 0;
}

// setDistCriterion of (i, j) for one workgroup (NJ.tcc:1115-1124), `i` being a node whose out-distance an earlier launch
// has brought up to date: optionally the distance (recompute), the lazy refresh of j (setCriterion's rule: recomputed when
// staler than nDiffAllow; several workgroups may refresh the same node - they store identical values, value before stamp),
// the criterion.  Every thread calls; the results are valid in thread 0.
// Optional behaviour of vft_th_pair for the speculative double walk of the join engine (k_nj_best_pairs2):
//   ovOut / ovStamp  the out-distance and stamp of node i to use instead of the arrays' (a forced refresh that is not committed yet);
//   claim / tag      one writer per refreshed node and launch: the workgroup that wins atomicExch(claim[j], tag) stores the
//                    refresh, the others compute the same value for themselves and store nothing;
//   log*             the winner records (node, old out-distance, old stamp) so that the launch can be undone.
template <typename REAL>
struct ThPairSpec {
    const REAL *ovOut;
    const int32_t *ovStamp;
    unsigned int *claim;
    unsigned int tag;
    int32_t *logNode, *logStamp, *logCount;
    REAL *logOut;
    bool rowsById = false;   // every internal node has its row (the join engine): no flag lookup in front of the column loads
};

template <typename REAL, int NC>
__device__ __forceinline__ void vft_th_pair(const Arena<REAL> &A, const SweepArgs &s, int64_t i, int64_t j, bool recompute,
                                            double *sW, double *sT, REAL &d, REAL &crit, const ThPairSpec<REAL> *spec = nullptr) {
    __shared__ int thStale;
    // thread 0 issues everything the criterion needs up front: the loads complete while the workgroup computes the distance
    int32_t sj = 0, si = 0;
    REAL oj = 0, oi = 0;
    if (threadIdx.x == 0) {
        sj = __hip_atomic_load(&A.nOutActive[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        oj = __hip_atomic_load(&A.outDist[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        si = spec && spec->ovStamp ? *spec->ovStamp : A.nOutActive[i];
        oi = spec && spec->ovOut ? *spec->ovOut : A.outDist[i];
    }
    if (recompute) {
        REAL w, diaI = 0, diaJ = 0;
        if (threadIdx.x == 0) {   // (asked for before the distance, not after it)
            diaI = A.diameter[i];
            diaJ = A.diameter[j];
        }
        vft_pair_block<REAL, NC>(A, i, j, false, sW, sT, d, w, spec && spec->rowsById);
        if (!(i < A.d.nSeqs && j < A.d.nSeqs)) {
            const REAL dd = diaI + diaJ;
            d = d - dd;   // (thread 0's value is the one used: the criterion and the staging store are thread 0's)
        }
    }
    if (threadIdx.x == 0) thStale = (int64_t) sj - s.nActive > s.nDiffAllow;
    __syncthreads();
    const bool stale = thStale != 0;
    if (stale) {
        REAL dd, ww;
        vft_pair_block<REAL, NC>(A, j, -1, true, sW, sT, dd, ww);
        if (threadIdx.x == 0) {
            const REAL oldOut = oj;
            const int32_t oldStamp = sj;
            oj = vft_out_distance<REAL>(dd, ww, s.nActive, A.selfweight[j], A.selfdist[j], A.diameter[j], s.totdiam);
            sj = (int32_t) s.nActive;
            // with claims: one writer (whoever gets there first); without: every workgroup that found the node stale stores
            // the same value - value first, stamp after it has been acknowledged: a reader that sees the new stamp sees the value
            const bool store = !(spec && spec->claim) || __hip_atomic_exchange(&spec->claim[j], spec->tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != spec->tag;
            if (store) {
                if (spec && spec->logCount) {   // (oldOut belongs to oldStamp: nobody else writes this node in this launch)
                    const int k = atomicAdd(spec->logCount, 1);
                    spec->logNode[k] = (int32_t) j;
                    spec->logOut[k] = oldOut;
                    spec->logStamp[k] = oldStamp;
                }
                __hip_atomic_store(&A.outDist[j], oj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                A.mOutDist[j] = oj;
                __threadfence_system();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&A.nOutActive[j], sj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                A.mNOut[j] = sj;
                __threadfence_system();   // the host-mapped mirrors are out before this workgroup counts itself
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
    } else if (threadIdx.x == 0 && (int64_t) sj == s.nActive) {
        // a stamp of this very step may have been written a moment ago by another workgroup of this launch (the same
        // partner listed twice): its value is read again, now certainly after the stamp
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        oj = __hip_atomic_load(&A.outDist[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) crit = vft_criterion<REAL>(d, oi, si, oj, sj, s.nActive);
}

template <typename REAL>
__device__ __forceinline__ void vft_th_stage(const TopHits<REAL> &T, int t, int32_t j, REAL d, REAL cr) {
    __hip_atomic_store(&T.stJ[t], j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&T.stD[t], d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&T.stC[t], cr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void vft_th_raise(unsigned long long *flag, unsigned long long seq) {
    // the results went to host-mapped memory: they must have left the chip before the flag moves (the explicit wait is not
    // redundant on ROCm 7.2 / gfx950, see vft_publish_staged)
    __threadfence_system();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// getBestFromTopHits(node) (NJ.tcc:4267-4298) over the device list of `node` (n entries, one workgroup each); the forced
// setOutDistance(node) of NJ.tcc:4273-4279 is an earlier launch on the same stream.  Dynamic LDS: 2 * nPosPad doubles.
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_WG) void k_th_best(Arena<REAL> A, TopHits<REAL> T, int64_t node, SweepArgs s, ThBestOut *out,
                                                    unsigned long long *flag, unsigned long long seq) {
    extern __shared__ __attribute__((aligned(16))) double pwLds[];
    const int t = (int) blockIdx.x, n = (int) gridDim.x;
    const ThHit<REAL> h = T.hits[node * T.m + t];                    // (workgroup-uniform loads)
    const int32_t j = vft_active_ancestor(A.parent, h.j);            // updateBestHit, NJ.tcc:1626-1648
    if (j < 0 || j == (int32_t) node) {
        if (threadIdx.x == 0) vft_th_stage<REAL>(T, t, -1, (REAL) 0, (REAL) 1e20);
    } else {
        REAL d = h.dist, cr = (REAL) 1e20;
        vft_th_pair<REAL, NC>(A, s, node, j, j != h.j, pwLds, pwLds + A.d.nPosPad, d, cr);
        if (threadIdx.x == 0) vft_th_stage<REAL>(T, t, j, d, cr);
    }
    if (!vft_th_arrive(T.doneCtr)) return;
    __threadfence();
    // the first strict minimum in list order: "if (bh.criterion < bestjoin.criterion) bestjoin = bh", starting from 1e20
    __shared__ double redC[VFT_WG];
    __shared__ int redT[VFT_WG];
    double bc = 1e20;
    int bt = 0x7FFFFFFF;
    for (int u = threadIdx.x; u < n; u += blockDim.x) {
        const int32_t ju = __hip_atomic_load(&T.stJ[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const REAL cu = __hip_atomic_load(&T.stC[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ju >= 0 && (double) cu < bc) {
            bc = (double) cu;
            bt = u;
        }
    }
    redC[threadIdx.x] = bc;
    redT[threadIdx.x] = bt;
    __syncthreads();
    for (int off = blockDim.x >> 1; off > 0; off >>= 1) {
        if ((int) threadIdx.x < off) {
            const double c2 = redC[threadIdx.x + off];
            const int t2 = redT[threadIdx.x + off];
            if (t2 != 0x7FFFFFFF && (redT[threadIdx.x] == 0x7FFFFFFF || c2 < redC[threadIdx.x] || (c2 == redC[threadIdx.x] && t2 < redT[threadIdx.x]))) {
                redC[threadIdx.x] = c2;
                redT[threadIdx.x] = t2;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int b = redT[0];
        if (b == 0x7FFFFFFF) {
            out->j = -1;
            out->pos = -1;
            out->dist = 1e20;
            out->crit = 1e20;
        } else {
            out->j = __hip_atomic_load(&T.stJ[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out->pos = b;
            out->dist = (double) __hip_atomic_load(&T.stD[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out->crit = (double) __hip_atomic_load(&T.stC[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    vft_th_raise(flag, seq);
}

// sort key of a candidate: ascending criterion (-0.0 = +0.0, as the reference's comparator sees them), then DESCENDING id
struct ThKey {
    unsigned long long key;
    uint32_t nj;
    int32_t src;
};
__device__ __forceinline__ unsigned long long vft_th_order(float x) {
    if (x == 0) x = 0;
    return (unsigned long long) vft_order_f32(x);
}
__device__ __forceinline__ unsigned long long vft_th_order(double x) {
    if (x == 0) x = 0;
    return vft_order_f64(x);
}
__device__ __forceinline__ bool vft_th_before(const ThKey &a, const ThKey &b) {
    return a.key != b.key ? a.key < b.key : a.nj < b.nj;
}

// ascending bitonic sort of P = 2^k keys in LDS by the whole workgroup
__device__ __forceinline__ void vft_th_bitonic(ThKey *keys, int P) {
    for (int size = 2; size <= P; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int u = threadIdx.x; u < (P >> 1); u += blockDim.x) {
                const int lo = 2 * u - (u & (stride - 1)), hi = lo + stride;   // pair (lo, lo + stride) of this stage
                const bool up = (lo & size) == 0;
                const ThKey a = keys[lo], b = keys[hi];
                if (vft_th_before(b, a) == up) {
                    keys[lo] = b;
                    keys[hi] = a;
                }
            }
        }
    __syncthreads();
}

// The merge of the two children's lists for the new node of a join: uniqueBestHits of the combined lists (every entry
// changes its first node, so every distance is recomputed: what is left is one candidate per distinct active ancestor of a
// partner, NJ.tcc:4319-4330 + 4786-4833), then the decision of NJ.tcc:4342-4362 and sortSaveBestHits (NJ.tcc:4535-4578).
// Grid: n0 + n1 workgroups.  The new node's own out-distance has been refreshed by an earlier launch.
// Dynamic LDS: max(2 * nPosPad doubles, P ThKeys) with P the power of two >= n0 + n1.
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_WG) void k_th_join(Arena<REAL> A, TopHits<REAL> T, int64_t newnode, int64_t c0, int32_t n0,
                                                    int64_t c1, SweepArgs s, unsigned int tag, int32_t nSaveMax, int32_t need,
                                                    int32_t ageOK, ThJoinInfo *info, int32_t *outJ, REAL *outD, REAL *outC,
                                                    unsigned long long *flag, unsigned long long seq) {
    extern __shared__ __attribute__((aligned(16))) double pwLds[];
    __shared__ int thOwner;
    const int t = (int) blockIdx.x, n = (int) gridDim.x;
    const ThHit<REAL> h = t < n0 ? T.hits[c0 * T.m + t] : T.hits[c1 * T.m + (t - n0)];
    const int32_t j = vft_active_ancestor(A.parent, h.j);
    if (threadIdx.x == 0)
        thOwner = j >= 0 && j != (int32_t) newnode && __hip_atomic_exchange(&T.mark[j], tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag;
    __syncthreads();
    if (!thOwner) {
        if (threadIdx.x == 0) vft_th_stage<REAL>(T, t, -1, (REAL) 0, (REAL) 1e20);
    } else {
        REAL d = 0, cr = (REAL) 1e20;
        vft_th_pair<REAL, NC>(A, s, newnode, j, true, pwLds, pwLds + A.d.nPosPad, d, cr);
        if (threadIdx.x == 0) vft_th_stage<REAL>(T, t, j, d, cr);
    }
    if (!vft_th_arrive(T.doneCtr)) return;
    __threadfence();
    ThKey *keys = (ThKey *) pwLds;
    __shared__ int thCount;
    int P = 1;
    while (P < n) P <<= 1;
    if (threadIdx.x == 0) thCount = 0;
    __syncthreads();
    for (int u = threadIdx.x; u < n; u += blockDim.x) {
        const int32_t ju = __hip_atomic_load(&T.stJ[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ju < 0) continue;
        const REAL cu = __hip_atomic_load(&T.stC[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ThKey k;
        k.key = vft_th_order(cu);
        k.nj = ~(uint32_t) ju;
        k.src = u;
        keys[atomicAdd(&thCount, 1)] = k;
    }
    __syncthreads();
    const int nU = thCount;
    for (int u = nU + threadIdx.x; u < P; u += blockDim.x) {
        ThKey k;
        k.key = ~0ull;
        k.nj = ~0u;
        k.src = -1;
        keys[u] = k;
    }
    vft_th_bitonic(keys, P);
    // NJ.tcc:4342-4362 (first-level lists): the merged list is used when it holds every other active node, or when it is
    // young enough and long enough; otherwise the caller refreshes the new node's list with a sweep
    const bool useUnique = (int64_t) nU == s.nActive - 1 || (ageOK && nU >= need);
    const int nSave = useUnique ? (nU < nSaveMax ? nU : nSaveMax) : 0;
    for (int r = threadIdx.x; r < nU; r += blockDim.x) {
        const int src = keys[r].src;
        const int32_t jr = __hip_atomic_load(&T.stJ[src], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const REAL dr = __hip_atomic_load(&T.stD[src], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const REAL cr = __hip_atomic_load(&T.stC[src], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        outJ[r] = jr;
        outD[r] = dr;
        outC[r] = cr;
        if (r < nSave) {
            ThHit<REAL> e;
            e.j = jr;
            e.dist = dr;
            T.hits[newnode * T.m + r] = e;
        }
    }
    if (threadIdx.x == 0) {
        if (useUnique) T.len[newnode] = nSave;
        info->nUnique = nU;
        info->useUnique = useUnique ? 1 : 0;
        info->nSave = nSave;
        info->pad = 0;
    }
    vft_th_raise(flag, seq);
}

// lists uploaded in one piece: packed[t][m] -> hits[nodes[t]][m], len[nodes[t]] = lens[t]; grid (count), any block size
template <typename REAL>
__global__ void k_th_scatter(TopHits<REAL> T, const int64_t *nodes, const int32_t *lens, const ThHit<REAL> *packed) {
    const int64_t t = blockIdx.x, node = nodes[t];
    const int32_t n = lens[t];
    for (int u = threadIdx.x; u < n; u += blockDim.x) T.hits[node * T.m + u] = packed[t * T.m + u];
    if (threadIdx.x == 0) T.len[node] = n;
}

// one list stored as it is: hits[node][0..n) = src[0..n), len[node] = n (the new node's own list of a top-hits refresh)
template <typename REAL>
__global__ void k_th_store(TopHits<REAL> T, int64_t node, const ThHit<REAL> *src, int32_t n) {
    for (int u = threadIdx.x; u < n; u += blockDim.x) T.hits[node * T.m + u] = src[u];
    if (threadIdx.x == 0) T.len[node] = n;
}

// The top-hits refresh of a join (topHitJoin's else-branch, NJ.tcc:4477-4515) for one of the new node's m closest nodes per
// workgroup.  The new node has just been swept (targets / tdist: the first nB records of its sorted hits - active nodes, or
// -1) and every out-distance is fresh enough (NJ.tcc:4451-4464 touches every active node first), so nothing below refreshes.
// For node x = nodes[t]: its own hits re-targeted to the active ancestors of their partners (updateBestHit), then the first
// 2 * nNew swept hits transferred to it (transferBestHits with updateDistances = false); uniqueBestHits keeps one record per
// partner - the LAST one in that order, the sort by partner id being stable under "ascending key, descending position" -
// recomputes the distance of every transferred record (block[t][u]: the cross product work x targets, k_pairs_block) and of
// the old records whose partner changed (or whose distance is negative, the reference's test) with vft_pair_block; criteria;
// sortSaveBestHits keeps the first nNew by (criterion ascending, partner id descending).  The swept node's own records keep
// their swept distances (x == newnode).  outLen / outFirst: the new length and first hit of every list (host-mapped: the
// caller's visible set).  Dynamic LDS: pair staging | P keys | 4 int arrays of E | E distances.
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_WG) void k_th_refresh(Arena<REAL> A, TopHits<REAL> T, int64_t newnode, const int64_t *nodes,
                                                       const int32_t *nNewArr, const int64_t *targets, const REAL *tdist, int32_t nB,
                                                       const REAL *block, SweepArgs s, int32_t P, int32_t E, int32_t *outLen,
                                                       ThHit<REAL> *outFirst, unsigned long long *flag, unsigned long long seq) {
    extern __shared__ __attribute__((aligned(16))) double pwLds[];
    double *sW = pwLds, *sT = pwLds + A.d.nPosPad;
    ThKey *keys = (ThKey *) (pwLds + 2 * A.d.nPosPad);
    int32_t *entJ = (int32_t *) (keys + P), *entSrc = entJ + E, *kept = entSrc + E, *todo = kept + E;
    REAL *entD = (REAL *) (todo + E);
    __shared__ int cnt, nKept, nTodo;
    const int t = (int) blockIdx.x;
    const int64_t x = nodes[t];
    const int nNew = nNewArr[t], nOld = T.len[x];
    if (threadIdx.x == 0) cnt = nKept = nTodo = 0;
    __syncthreads();
    for (int u = threadIdx.x; u < nOld; u += blockDim.x) {
        const ThHit<REAL> h = T.hits[x * T.m + u];
        const int32_t j = vft_active_ancestor(A.parent, h.j);
        if (j < 0 || j == (int32_t) x) continue;
        const int slot = atomicAdd(&cnt, 1);
        entJ[slot] = j;
        entSrc[slot] = (j == h.j && !(h.dist < (REAL) 0)) ? -1 : -2;
        entD[slot] = h.dist;
        ThKey k;
        k.key = (unsigned long long) (uint32_t) j;
        k.nj = ~(uint32_t) u;
        k.src = slot;
        keys[slot] = k;
    }
    const int nT = 2 * nNew < nB ? 2 * nNew : nB;
    for (int u = threadIdx.x; u < nT; u += blockDim.x) {
        const int64_t j = targets[u];
        if (j < 0 || j == x) continue;
        const int slot = atomicAdd(&cnt, 1);
        entJ[slot] = (int32_t) j;
        REAL d = (REAL) -1e20;
        int32_t src = u;
        if (x == newnode) {   // the swept node's own record: the distance is the sweep's
            d = tdist[u];
            src = d < (REAL) 0 ? -2 : -1;
        }
        entSrc[slot] = src;
        entD[slot] = d;
        ThKey k;
        k.key = (unsigned long long) (uint32_t) j;
        k.nj = ~(uint32_t) (nOld + u);
        k.src = slot;
        keys[slot] = k;
    }
    __syncthreads();
    const int n1 = cnt;
    int P1 = 2;
    while (P1 < n1) P1 <<= 1;
    for (int u = n1 + threadIdx.x; u < P1; u += blockDim.x) {
        ThKey k;
        k.key = ~0ull;
        k.nj = ~0u;
        k.src = -1;
        keys[u] = k;
    }
    vft_th_bitonic(keys, P1);
    // one record per partner: the first of every run (= the one that came last)
    for (int r = threadIdx.x; r < n1; r += blockDim.x) {
        if (r > 0 && keys[r - 1].key == keys[r].key) continue;
        const int slot = keys[r].src;
        kept[atomicAdd(&nKept, 1)] = slot;
        const int32_t src = entSrc[slot];
        if (src >= 0) entD[slot] = block[(int64_t) t * nB + src];
        else if (src == -2) todo[atomicAdd(&nTodo, 1)] = slot;
    }
    __syncthreads();
    const int n2 = nKept, nRe = nTodo;
    // setDistCriterion for the old hits whose partner changed (dozens to hundreds per list late in a run).  One workgroup-wide
    // pair evaluation after the other was most of this kernel (1.9 ms per refresh at a million sequences); all of them have x
    // on one side: x's columns go to LDS once - the sort keys' space is free here - and a LANE walks the columns of one pair in
    // order with its own double accumulators, as k_pairs_block_tiled does (the reference's sequence of additions, NJ.tcc:1168-1183).
    const int64_t nPos = A.d.nPos;
    if (nRe > 1 && (size_t) nPos * ((NC + 1) * sizeof(REAL) + 4) <= (size_t) P * sizeof(ThKey)) {
        REAL *sF = (REAL *) keys;                         // [nPos][NC]
        REAL *sWt = sF + nPos * NC;                       // [nPos]
        int32_t *sCode = (int32_t *) (sWt + nPos);        // [nPos]; bit 8: the column holds a vector
        for (int64_t p = threadIdx.x; p < nPos; p += blockDim.x) {
            Col<REAL, NC> c;
            c.w = 0;
            c.code = VFT_NOCODE_;
            c.vec = false;
#pragma unroll
            for (int k = 0; k < NC; k++) c.f[k] = 0;
            vft_load_col_ml<REAL, NC>(A, x, p, c);
            sWt[p] = c.w;
            sCode[p] = c.code | (c.vec ? 256 : 0);
#pragma unroll
            for (int k = 0; k < NC; k++) sF[p * NC + k] = c.vec ? c.f[k] : (REAL) 0;
        }
        __syncthreads();
        for (int k = threadIdx.x; k < nRe; k += blockDim.x) {
            const int slot = todo[k];
            const int64_t j = entJ[slot];
            double top = 0.0, den = 0.0;
            // (four columns of the partner per trip, their loads issued together: one column at a time this loop was a memory
            //  latency per column - 200 of them per pair)
            constexpr int U = 4;
            const bool jRow = vft_is_row<REAL>(A, j);
            for (int64_t p0 = 0; p0 < nPos; p0 += U) {
                Col<REAL, NC> cbs[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int64_t p = p0 + u < nPos ? p0 + u : nPos - 1;   // (clamped: unconditional loads)
                    if (jRow) vft_load_row<REAL, NC>(A, j, p, cbs[u]);
                    else vft_load_col<REAL, NC>(A, j, p, cbs[u]);
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int64_t p = p0 + u;
                    if (p >= nPos) break;
                    const Col<REAL, NC> &cb = cbs[u];
                    if (!(cb.w > 0)) continue;
                    Col<REAL, NC> ca;
                    ca.w = sWt[p];
                    if (!(ca.w > 0)) continue;
                    const int32_t cc = sCode[p];
                    ca.code = cc & 255;
                    ca.vec = (cc & 256) != 0;
#pragma unroll
                    for (int q = 0; q < NC; q++) ca.f[q] = sF[p * NC + q];
                    const REAL ww = ca.w * cb.w;
                    const double wgt = (double) ww;
                    den += wgt;
                    top += wgt * vft_piece<REAL, NC>(A, ca, cb, nullptr);
                }
            }
            REAL d = (REAL) (den > 0 ? top / den : 1.0);
            if (!(x < A.d.nSeqs && j < A.d.nSeqs)) {
                const REAL dd = A.diameter[x] + A.diameter[j];
                d = d - dd;
            }
            entD[slot] = d;
        }
    } else {
        for (int k = 0; k < nRe; k++) {
            const int slot = todo[k];
            const int64_t j = entJ[slot];
            REAL d, w;
            vft_pair_block<REAL, NC>(A, x, j, false, sW, sT, d, w);
            if (threadIdx.x == 0) {
                if (!(x < A.d.nSeqs && j < A.d.nSeqs)) {
                    const REAL dd = A.diameter[x] + A.diameter[j];
                    d = d - dd;
                }
                entD[slot] = d;
            }
        }
    }
    __syncthreads();
    const REAL oi = A.outDist[x];
    const int32_t si = A.nOutActive[x];
    for (int r = threadIdx.x; r < n2; r += blockDim.x) {
        const int slot = kept[r];
        const int32_t j = entJ[slot];
        const REAL cr = vft_criterion<REAL>(entD[slot], oi, si, A.outDist[j], A.nOutActive[j], s.nActive);
        ThKey k;
        k.key = vft_th_order(cr);
        k.nj = ~(uint32_t) j;
        k.src = slot;
        keys[r] = k;
    }
    int P2 = 2;
    while (P2 < n2) P2 <<= 1;
    for (int u = n2 + threadIdx.x; u < P2; u += blockDim.x) {
        ThKey k;
        k.key = ~0ull;
        k.nj = ~0u;
        k.src = -1;
        keys[u] = k;
    }
    vft_th_bitonic(keys, P2);
    const int nSave = n2 < nNew ? n2 : nNew;
    for (int r = threadIdx.x; r < nSave; r += blockDim.x) {
        const int slot = keys[r].src;
        ThHit<REAL> e;
        e.j = entJ[slot];
        e.dist = entD[slot];
        T.hits[x * T.m + r] = e;
        if (r == 0) outFirst[t] = e;
    }
    if (threadIdx.x == 0) {
        T.len[x] = nSave;
        outLen[t] = nSave;
        if (nSave == 0) {
            ThHit<REAL> e;
            e.j = -1;
            e.dist = (REAL) 1e20;
            outFirst[t] = e;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) __threadfence_system();   // this workgroup's host-mapped results are out before it counts itself
    if (!vft_th_arrive(T.doneCtr)) return;
    vft_th_raise(flag, seq);
}
