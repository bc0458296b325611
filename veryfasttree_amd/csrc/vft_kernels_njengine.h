// The join loop of fastNJ with top hits (NJ.tcc:2857-3047: topHitNJSearch :4137-4262, the join body :2897-3042, topHitJoin
// :4306-4438) as a stream of kernels that take their arguments from a device-resident state block: the host enqueues the
// same launches for every join without waiting for any of them, reads the join records later, and only steps in for the
// events the kernels flag - a top-visible reset, a top-hits refresh, a hill-climbing that needs another round.
//
// Round 2's loop made three host round trips per join (two list walks, one merge), with the host walking 1 000-entry lists
// in between: 377 us per join at a million sequences.  Here the visible set (visible[] / topvisible[], NJ.h:236-246), the
// ages of the lists and the scalars of the loop (nActive, maxnode, totdiam, the candidate join) live next to the lists
// (vft_kernels_tophits.h) and a join is the launch sequence
//     k_nj_scan                                   top-visible scan, candidate join, reset test           (:4137-4209)
//     k_nj_refresh_cur, k_nj_best_pairs, k_nj_best_tail   x2: getBestFromTopHits for both ends       (:4226-4261, 4267-4298)
//     k_nj_refresh_cur x2, k_nj_join              criterion on fresh out-distances, the join itself      (:2897-3042)
//     k_nj_refresh_new, k_nj_merge_pairs, k_nj_merge_tail   the new node's list, visible-set updates   (:4306-4438, 4633-4726)
// Every kernel starts by looking at state->halt and does nothing when an earlier kernel has raised it; the host resumes
// after it has dealt with the event.  Results are bit-identical to the host-driven loop (same formulas, same order of every
// floating-point operation; -ffp-contract=off).
#pragma once
#include "vft_kernels_tophits.h"

#define VFT_NJ_TAIL 1024   // threads of the single-workgroup kernels (scan, merge tail)

// halt reasons
#define VFT_NJ_HALT_RESET 1     // topHitNJSearch wants resetTopVisible (NJ.tcc:4156-4206); nothing of the join has happened
#define VFT_NJ_HALT_REFRESH 2   // topHitJoin wants a top-hits refresh for the new node (NJ.tcc:4440-4517); the join is done
#define VFT_NJ_HALT_CLIMB 3     // the hill climbing changed the candidate in its last enqueued round: another round
#define VFT_NJ_HALT_ERROR 4

struct NjJoinRec {       // one join as the host needs it (NJ.tcc:2904-2916, 2993-3007)
    int32_t i, j, newnode, pad;
    double dist, criterion, blI, blJ, diameter;
};

template <typename REAL>
struct NjState {
    long long nActive, maxnode, joinsDone;
    double totdiam;
    int32_t halt, haltJoin, changed, tvAge;
    int32_t curI, curJ;
    REAL curDist, curCrit;
    int32_t nUnique, pad;
};

template <typename REAL>
struct NjEngine {
    NjState<REAL> *st;
    int32_t *visJ;          // visible[node].j, visible[node].dist (NJ.h:236-240)
    REAL *visD;
    int32_t *topvis;        // topvisible[nTop] (NJ.h:242-246)
    int32_t *age;           // TopHitsList::age
    NjJoinRec *logDev, *logHost;
    volatile long long *hostStatus;   // host-mapped: [0] joinsDone, [1] halt, [2] haltJoin
    int32_t m, nTop, need, ageLimit, fastest, staleStamp;
    double staleOutLimit, tol;
    REAL *stash;            // vft_join_fused's pending stash
    int64_t *pendIds;
};

template <typename REAL>
__device__ __forceinline__ long long vft_nj_allow(const NjEngine<REAL> &E, long long nActive) {
    return (long long) ((double) nActive * E.staleOutLimit);   // nDiffAllow, NJ.tcc:1094 / Options.h:38
}

template <typename REAL>
__device__ __forceinline__ SweepArgs vft_nj_args(const NjEngine<REAL> &E, long long nActive, double totdiam) {
    SweepArgs s{};
    s.nActive = nActive;
    s.nDiffAllow = vft_nj_allow(E, nActive);
    s.totdiam = totdiam;
    return s;
}

// setOutDistance(v) by the whole workgroup (NJ.tcc:1012-1053); every thread calls
template <typename REAL, int NC>
__device__ __forceinline__ void vft_nj_out_distance(const Arena<REAL> &A, const SweepArgs &s, int64_t v, double *sW, double *sT) {
    REAL d, w;
    vft_pair_block<REAL, NC>(A, v, -1, true, sW, sT, d, w);
    if (threadIdx.x == 0) {
        const REAL od = vft_out_distance<REAL>(d, w, s.nActive, A.selfweight[v], A.selfdist[v], A.diameter[v], s.totdiam);
        A.outDist[v] = od;
        A.nOutActive[v] = (int32_t) s.nActive;
        A.mOutDist[v] = od;
        A.mNOut[v] = (int32_t) s.nActive;
    }
    __syncthreads();
}

// the lazy refreshes of a batch of setCriterion calls, single-workgroup kernels: `list` (LDS, n entries, duplicates allowed)
// holds nodes some thread found staler than allowed; each is looked at again and refreshed once
template <typename REAL, int NC>
__device__ __forceinline__ void vft_nj_refresh_listed(const Arena<REAL> &A, const SweepArgs &s, const int32_t *list, int n,
                                                      double *sW, double *sT) {
    for (int k = 0; k < n; k++) {
        const int32_t v = list[k];
        if (!((long long) A.nOutActive[v] - s.nActive > s.nDiffAllow)) continue;   // (uniform: every thread reads the same word)
        vft_nj_out_distance<REAL, NC>(A, s, v, sW, sT);
    }
}

// getVisible's test (NJ.tcc:546-557): node active with an active visible partner
template <typename REAL>
__device__ __forceinline__ bool vft_nj_visible_ok(const Arena<REAL> &A, const NjEngine<REAL> &E, int32_t node, int32_t &vj) {
    vj = -1;
    if (node < 0 || A.parent[node] >= 0) return false;
    vj = E.visJ[node];
    return vj >= 0 && A.parent[vj] < 0;
}

template <typename REAL>
__device__ __forceinline__ REAL vft_nj_crit(const Arena<REAL> &A, REAL dist, int32_t i, int32_t j, long long nActive) {
    return vft_criterion<REAL>(dist, A.outDist[i], A.nOutActive[i], A.outDist[j], A.nOutActive[j], nActive);
}

template <typename REAL>
__device__ __forceinline__ void vft_nj_publish(const NjEngine<REAL> &E, const NjState<REAL> *st) {
    E.hostStatus[0] = st->joinsDone;
    E.hostStatus[1] = st->halt;
    E.hostStatus[2] = st->haltJoin;
    __threadfence_system();
}

// ---------------------------------------------------------------------------------------------------------------
// topHitNJSearch up to the hill climbing (NJ.tcc:4137-4223).  One workgroup of VFT_NJ_TAIL threads.
// Dynamic LDS: 2 * nPosPad doubles | nTop x (int32 stale list x 2)
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_NJ_TAIL) void k_nj_scan(Arena<REAL> A, NjEngine<REAL> E, long long joinIndex) {
    extern __shared__ __attribute__((aligned(16))) double njLds[];
    NjState<REAL> *st = E.st;
    if (st->halt) return;
    double *sW = njLds, *sT = njLds + A.d.nPosPad;
    int32_t *staleList = (int32_t *) (njLds + 2 * A.d.nPosPad);
    __shared__ int nStale, nCand;
    __shared__ double redC[VFT_NJ_TAIL];
    __shared__ int redT[VFT_NJ_TAIL];
    const long long nActive = st->nActive;
    const SweepArgs s = vft_nj_args(E, nActive, st->totdiam);
    if (threadIdx.x == 0) nStale = nCand = 0;
    __syncthreads();
    // prefetchVisible(topvisible): the lazy refreshes of every getVisible of the scan
    for (int t = threadIdx.x; t < E.nTop; t += blockDim.x) {
        int32_t vj;
        const int32_t node = E.topvis[t];
        if (!vft_nj_visible_ok(A, E, node, vj)) continue;
        if ((long long) A.nOutActive[node] - nActive > s.nDiffAllow) staleList[atomicAdd(&nStale, 1)] = node;
        if ((long long) A.nOutActive[vj] - nActive > s.nDiffAllow) staleList[atomicAdd(&nStale, 1)] = vj;
    }
    __syncthreads();
    vft_nj_refresh_listed<REAL, NC>(A, s, staleList, nStale, sW, sT);
    __syncthreads();
    // the first minimum in array order ("bestNode < 0 || v.criterion < bestCrit")
    double bc = 0;
    int bt = 0x7FFFFFFF, mine = 0;
    for (int t = threadIdx.x; t < E.nTop; t += blockDim.x) {
        int32_t vj;
        const int32_t node = E.topvis[t];
        if (!vft_nj_visible_ok(A, E, node, vj)) continue;
        mine++;
        const double c = (double) vft_nj_crit<REAL>(A, E.visD[node], node, vj, nActive);
        if (bt == 0x7FFFFFFF || c < bc) {
            bc = c;
            bt = t;
        }
    }
    if (mine) atomicAdd(&nCand, mine);
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
        const int age = ++st->tvAge;
        const long long cand = nCand;
        if (2ll * age > E.m || (3 * cand < E.nTop && 3 * cand < nActive) || redT[0] == 0x7FFFFFFF) {
            st->halt = VFT_NJ_HALT_RESET;
            st->haltJoin = (int32_t) joinIndex;
            vft_nj_publish(E, st);
        } else {
            const int32_t node = E.topvis[redT[0]];
            st->curI = node;
            st->curJ = E.visJ[node];
            st->curDist = E.visD[node];
            st->curCrit = (REAL) redC[0];
            st->changed = 0;
        }
    }
}

// setOutDistance for one end of the candidate join (which: 0 = i, 1 = j): getBestFromTopHits' own node (NJ.tcc:4273-4279)
// and the two ends before the join (:2897-2898).  Recomputed unless the stamp is nActive.  One workgroup of VFT_WG threads.
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_WG) void k_nj_refresh_cur(Arena<REAL> A, NjEngine<REAL> E, int which, int onlyIfChanged) {
    extern __shared__ __attribute__((aligned(16))) double njLds[];
    const NjState<REAL> *st = E.st;
    if (st->halt || (onlyIfChanged && !st->changed)) return;
    const int64_t v = which ? st->curJ : st->curI;
    if ((long long) A.nOutActive[v] == st->nActive) return;
    const SweepArgs s = vft_nj_args(E, st->nActive, st->totdiam);
    vft_nj_out_distance<REAL, NC>(A, s, v, njLds, njLds + A.d.nPosPad);
}

// getBestFromTopHits for one end of the candidate (NJ.tcc:4267-4298): one workgroup per list entry (grid = m; workgroups
// beyond the list's length leave), results into the staging arrays; k_nj_best_tail picks.
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_WG) void k_nj_best_pairs(Arena<REAL> A, NjEngine<REAL> E, TopHits<REAL> T, int which, int onlyIfChanged) {
    extern __shared__ __attribute__((aligned(16))) double njLds[];
    const NjState<REAL> *st = E.st;
    if (st->halt || (onlyIfChanged && !st->changed)) return;
    const int64_t node = which ? st->curJ : st->curI;
    const int t = (int) blockIdx.x;
    if (t >= T.len[node]) return;
    const SweepArgs s = vft_nj_args(E, st->nActive, st->totdiam);
    const ThHit<REAL> h = T.hits[node * T.m + t];
    const int32_t j = vft_active_ancestor(A.parent, h.j);
    if (j < 0 || j == (int32_t) node) {
        if (threadIdx.x == 0) T.stJ[t] = -1;
        return;
    }
    REAL d = h.dist, cr = (REAL) 1e20;
    vft_th_pair<REAL, NC>(A, s, node, j, j != h.j, njLds, njLds + A.d.nPosPad, d, cr);
    if (threadIdx.x == 0) {
        T.stJ[t] = j;
        T.stD[t] = d;
        T.stC[t] = cr;
    }
}

// the end of getBestFromTopHits and the comparison of the hill climbing (NJ.tcc:4226-4260): the first strict minimum in list
// order; "if (best.j != join.<other end> && best.criterion < join.criterion) join = best".  which == 1 closes a round: when
// the candidate changed in it and `lastRound` is set, the host has to enqueue another round.  One workgroup of VFT_WG.
template <typename REAL>
__global__ __launch_bounds__(VFT_WG) void k_nj_best_tail(NjEngine<REAL> E, TopHits<REAL> T, int which, int onlyIfChanged, int lastRound,
                                                         long long joinIndex) {
    NjState<REAL> *st = E.st;
    if (st->halt || (onlyIfChanged && !st->changed)) return;
    __shared__ double redC[VFT_WG];
    __shared__ int redT[VFT_WG];
    const int64_t node = which ? st->curJ : st->curI;
    const int n = T.len[node];
    double bc = 1e20;
    int bt = 0x7FFFFFFF;
    for (int u = threadIdx.x; u < n; u += blockDim.x) {
        if (T.stJ[u] < 0) continue;
        const REAL cu = T.stC[u];
        if ((bt == 0x7FFFFFFF && cu < (REAL) 1e20) || (bt != 0x7FFFFFFF && (double) cu < bc)) {
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
    if (threadIdx.x != 0) return;
    if (which == 0 && !onlyIfChanged) st->changed = 0;     // (a round starts: "changed = false")
    else if (which == 0) st->changed = 0;
    const int b = redT[0];
    if (b != 0x7FFFFFFF) {
        const int32_t bj = T.stJ[b];
        const REAL bcr = T.stC[b];
        const int32_t other = which ? st->curI : st->curJ;
        if (bj != other && bcr < st->curCrit) {
            st->changed = 1;
            st->curI = (int32_t) node;
            st->curJ = bj;
            st->curDist = T.stD[b];
            st->curCrit = bcr;
        }
    }
    if (which == 1 && lastRound && st->changed) {
        st->halt = VFT_NJ_HALT_CLIMB;
        st->haltJoin = (int32_t) joinIndex;
        vft_nj_publish(E, st);
    }
}

// The join itself (NJ.tcc:2897-3042): criterion on the fresh out-distances of both ends (k_nj_refresh_cur ran for both),
// tree arrays, branch lengths, diameter, the new profile, its self distance, the incremental out-profile (vft_join_body),
// totdiam, the counters.  One workgroup of VFT_WG_PROF threads; dynamic LDS: 2 * nPosPad doubles.
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_WG_PROF) void k_nj_join(Arena<REAL> A, NjEngine<REAL> E, long long joinIndex, int32_t updateOut, int32_t slot) {
    extern __shared__ __attribute__((aligned(16))) double njLds[];
    NjState<REAL> *st = E.st;
    if (st->halt) return;
    __shared__ REAL sDiam;
    const long long nActive = st->nActive;
    const int64_t i = st->curI, j = st->curJ, newn = st->maxnode;
    if (threadIdx.x == 0) {
        const REAL dist = st->curDist;
        const REAL crit = vft_nj_crit<REAL>(A, dist, (int32_t) i, (int32_t) j, nActive);   // criterionFresh / setDistCriterion(join)
        // NJ.tcc:2911-2916, 3003-3007 (BIONJ off: weight 1/2)
        const double distIJ = (double) dist;
        const REAL od = A.outDist[i] - A.outDist[j];
        const double deltaDist = (double) od / (double) (nActive - 2);
        const REAL blI = (REAL) ((distIJ + deltaDist) / 2), blJ = (REAL) ((distIJ - deltaDist) / 2);
        const double bw = 0.5;
        const REAL bi = blI + A.diameter[i], bj = blJ + A.diameter[j];
        const REAL diam = (REAL) (bw * (double) bi + (1 - bw) * (double) bj);
        sDiam = diam;
        NjJoinRec r;
        r.i = (int32_t) i;
        r.j = (int32_t) j;
        r.newnode = (int32_t) newn;
        r.pad = 0;
        r.dist = (double) dist;
        r.criterion = (double) crit;
        r.blI = (double) blI;
        r.blJ = (double) blJ;
        r.diameter = (double) diam;
        E.logDev[joinIndex] = r;
        E.logHost[joinIndex] = r;
        if (updateOut) {   // (a full out-profile follows otherwise, and the host sets totdiam from the diameters)
            const REAL dd = diam - A.diameter[i] - A.diameter[j];
            st->totdiam += (double) dd;
        }
        A.mOutDist[newn] = 0;
        A.mNOut[newn] = E.staleStamp;
    }
    __syncthreads();
    vft_join_body<REAL, NC>(A, i, j, newn, sDiam, E.staleStamp, nActive, updateOut, E.tol, E.stash, E.pendIds, slot, njLds);
    if (threadIdx.x == 0) {
        st->maxnode = newn + 1;
        st->nActive = nActive - 1;
    }
}

// the new node's out-distance (the first setCriterion of topHitJoin refreshes it: its stamp is "unreasonably high")
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_WG) void k_nj_refresh_new(Arena<REAL> A, NjEngine<REAL> E) {
    extern __shared__ __attribute__((aligned(16))) double njLds[];
    const NjState<REAL> *st = E.st;
    if (st->halt) return;
    const SweepArgs s = vft_nj_args(E, st->nActive, st->totdiam);
    vft_nj_out_distance<REAL, NC>(A, s, st->maxnode - 1, njLds, njLds + A.d.nPosPad);
}

// uniqueBestHits of the two children's lists (k_th_join's first half; grid = 2 m, workgroups beyond the lists leave)
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_WG) void k_nj_merge_pairs(Arena<REAL> A, NjEngine<REAL> E, TopHits<REAL> T, long long joinIndex) {
    extern __shared__ __attribute__((aligned(16))) double njLds[];
    const NjState<REAL> *st = E.st;
    if (st->halt) return;
    __shared__ int thOwner;
    const NjJoinRec rec = E.logDev[joinIndex];
    const int64_t newnode = rec.newnode, c0 = rec.i < rec.j ? rec.i : rec.j, c1 = rec.i < rec.j ? rec.j : rec.i;
    const int n0 = T.len[c0], n1 = T.len[c1], t = (int) blockIdx.x;
    if (t >= n0 + n1) return;
    const SweepArgs s = vft_nj_args(E, st->nActive, st->totdiam);
    const ThHit<REAL> h = t < n0 ? T.hits[c0 * T.m + t] : T.hits[c1 * T.m + (t - n0)];
    const int32_t j = vft_active_ancestor(A.parent, h.j);
    const unsigned int tag = (unsigned int) (joinIndex + 1);
    if (threadIdx.x == 0)
        thOwner = j >= 0 && j != (int32_t) newnode && __hip_atomic_exchange(&T.mark[j], tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag;
    __syncthreads();
    if (!thOwner) {
        if (threadIdx.x == 0) T.stJ[t] = -1;
        return;
    }
    REAL d = 0, cr = (REAL) 1e20;
    vft_th_pair<REAL, NC>(A, s, newnode, j, true, njLds, njLds + A.d.nPosPad, d, cr);
    if (threadIdx.x == 0) {
        T.stJ[t] = j;
        T.stD[t] = d;
        T.stC[t] = cr;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The rest of topHitJoin for a merged list (NJ.tcc:4342-4438): the age of the new list, the decision, sortSaveBestHits, the
// new node's visible hit, updateTopVisible for it, updateVisible over the saved hits.  One workgroup of VFT_NJ_TAIL threads.
//
// updateTopVisible(iIn, hit) (NJ.tcc:4660-4726), by the whole workgroup on the LDS copy of topvisible[]:
//   1. the first slot that holds iIn already (done) or a dead / empty node (it takes the slot);
//   2. otherwise getVisible of every slot in order until one fails (iIn takes that slot) or shows the same pair from the other
//      side (done); the lazy refreshes of those getVisible calls only happen for the slots the reference's loop reaches;
//   3. otherwise the slot with the worst criterion (the last one among equals) is replaced if the new hit is better.
template <typename REAL, int NC>
__device__ __forceinline__ void vft_nj_update_top_visible(const Arena<REAL> &A, const NjEngine<REAL> &E, const SweepArgs &s,
                                                          int32_t *tv, int32_t iIn, int32_t hitJ, REAL hitDist, double *sW, double *sT,
                                                          int32_t *staleList, double *redC, int *redT) {
    __shared__ int first1, stop2, nStale2;
    const long long nActive = s.nActive;
    if (threadIdx.x == 0) {
        first1 = 0x7FFFFFFF;
        stop2 = 0x7FFFFFFF;
        nStale2 = 0;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < E.nTop; t += blockDim.x) {
        const int32_t node = tv[t];
        if (node == iIn || node < 0 || A.parent[node] >= 0) atomicMin(&first1, t);
    }
    __syncthreads();
    if (first1 != 0x7FFFFFFF) {
        if (threadIdx.x == 0 && tv[first1] != iIn) tv[first1] = iIn;
        __syncthreads();
        return;
    }
    // no free slot: every slot holds an active node.  Where does the reference's scan stop?
    for (int t = threadIdx.x; t < E.nTop; t += blockDim.x) {
        int32_t vj;
        const int32_t node = tv[t];
        const bool ok = vft_nj_visible_ok(A, E, node, vj);
        if (!ok || (node == hitJ && vj == iIn)) atomicMin(&stop2, t);
    }
    __syncthreads();
    const int stop = stop2;
    const bool stopOk = stop != 0x7FFFFFFF && ({ int32_t vj; vft_nj_visible_ok(A, E, tv[stop], vj); });   // the same pair from the other side
    // lazy refreshes of the getVisible calls the scan makes: slots before the stop, and the stopping slot if its getVisible succeeded
    const int reach = stop == 0x7FFFFFFF ? E.nTop : (stopOk ? stop + 1 : stop);
    for (int t = threadIdx.x; t < reach; t += blockDim.x) {
        int32_t vj;
        const int32_t node = tv[t];
        vft_nj_visible_ok(A, E, node, vj);
        if ((long long) A.nOutActive[node] - nActive > s.nDiffAllow) staleList[atomicAdd(&nStale2, 1)] = node;
        if ((long long) A.nOutActive[vj] - nActive > s.nDiffAllow) staleList[atomicAdd(&nStale2, 1)] = vj;
    }
    if (stop == 0x7FFFFFFF && threadIdx.x == 0) {   // the final setCriterion(iIn, hit.j)
        if ((long long) A.nOutActive[iIn] - nActive > s.nDiffAllow) staleList[atomicAdd(&nStale2, 1)] = iIn;
        if ((long long) A.nOutActive[hitJ] - nActive > s.nDiffAllow) staleList[atomicAdd(&nStale2, 1)] = hitJ;
    }
    __syncthreads();
    vft_nj_refresh_listed<REAL, NC>(A, s, staleList, nStale2, sW, sT);
    __syncthreads();
    if (stop != 0x7FFFFFFF) {
        if (!stopOk && threadIdx.x == 0) tv[stop] = iIn;
        __syncthreads();
        return;
    }
    // the worst slot: "vis.criterion >= critWorst" in slot order = the largest criterion, the last one among equals
    double wc = -1e20;
    int wt = -1;
    for (int t = threadIdx.x; t < E.nTop; t += blockDim.x) {
        const int32_t node = tv[t];
        const double c = (double) vft_nj_crit<REAL>(A, E.visD[node], node, E.visJ[node], nActive);
        if (c >= wc) {
            wc = c;
            wt = t;
        }
    }
    redC[threadIdx.x] = wc;
    redT[threadIdx.x] = wt;
    __syncthreads();
    for (int off = blockDim.x >> 1; off > 0; off >>= 1) {
        if ((int) threadIdx.x < off) {
            const double c2 = redC[threadIdx.x + off];
            const int t2 = redT[threadIdx.x + off];
            if (t2 >= 0 && (redT[threadIdx.x] < 0 || c2 > redC[threadIdx.x] || (c2 == redC[threadIdx.x] && t2 > redT[threadIdx.x]))) {
                redC[threadIdx.x] = c2;
                redT[threadIdx.x] = t2;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && redT[0] >= 0) {
        const REAL b = vft_nj_crit<REAL>(A, hitDist, iIn, hitJ, nActive);
        if ((double) b < redC[0]) tv[redT[0]] = iIn;
    }
    __syncthreads();
}

// Dynamic LDS: max(2 * nPosPad doubles | P keys | P distances, ...) laid out as: pair staging | ThKey[P] | REAL[P] (distances by
// staging index) | int32[nTop] topvisible | int32[2 * nTop + 2 * P] stale lists / pass list
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_NJ_TAIL) void k_nj_merge_tail(Arena<REAL> A, NjEngine<REAL> E, TopHits<REAL> T, long long joinIndex, int P) {
    extern __shared__ __attribute__((aligned(16))) double njLds[];
    NjState<REAL> *st = E.st;
    if (st->halt) return;
    double *sW = njLds, *sT = njLds + A.d.nPosPad;
    ThKey *keys = (ThKey *) (njLds + 2 * A.d.nPosPad);
    REAL *distL = (REAL *) (keys + P);
    int32_t *tv = (int32_t *) (distL + P);
    int32_t *staleList = tv + E.nTop;             // 2 * nTop + 2 entries (updateTopVisible), 2 * P (updateVisible)
    int32_t *passList = staleList + 2 * E.nTop + 2 * P + 2;   // P entries
    __shared__ int thCount, nStale, nPass;
    __shared__ double redC[VFT_NJ_TAIL];
    __shared__ int redT[VFT_NJ_TAIL];
    const NjJoinRec rec = E.logDev[joinIndex];
    const int32_t newnode = rec.newnode, c0 = rec.i < rec.j ? rec.i : rec.j, c1 = rec.i < rec.j ? rec.j : rec.i;
    const int n = T.len[c0] + T.len[c1];
    const long long nActive = st->nActive;
    const SweepArgs s = vft_nj_args(E, nActive, st->totdiam);
    if (threadIdx.x == 0) thCount = nStale = nPass = 0;
    for (int t = threadIdx.x; t < E.nTop; t += blockDim.x) tv[t] = E.topvis[t];
    __syncthreads();
    for (int u = threadIdx.x; u < n; u += blockDim.x) {
        const int32_t ju = T.stJ[u];
        if (ju < 0) continue;
        ThKey k;
        k.key = vft_th_order(T.stC[u]);
        k.nj = ~(uint32_t) ju;
        k.src = u;
        keys[atomicAdd(&thCount, 1)] = k;
        distL[u] = T.stD[u];
    }
    __syncthreads();
    const int nU = thCount;
    int P1 = 2;
    while (P1 < nU) P1 <<= 1;
    for (int u = nU + threadIdx.x; u < P1; u += blockDim.x) {
        ThKey k;
        k.key = ~0ull;
        k.nj = ~0u;
        k.src = -1;
        keys[u] = k;
    }
    vft_th_bitonic(keys, P1);
    // NJ.tcc:4342-4362
    const int32_t ageNew = (E.age[c0] + E.age[c1] + 1) / 2 + 1;
    const bool useUnique = (long long) nU == nActive - 1 || (ageNew <= E.ageLimit && nU >= E.need);
    if (!useUnique) {
        if (threadIdx.x == 0) {
            E.age[newnode] = ageNew;
            st->nUnique = nU;
            st->joinsDone = joinIndex + 1;
            st->halt = VFT_NJ_HALT_REFRESH;
            st->haltJoin = (int32_t) joinIndex;
            vft_nj_publish(E, st);
        }
        return;
    }
    const int nSave = nU < E.m ? nU : E.m;
    for (int r = threadIdx.x; r < nSave; r += blockDim.x) {
        ThHit<REAL> e;
        e.j = (int32_t) ~keys[r].nj;
        e.dist = distL[keys[r].src];
        T.hits[(int64_t) newnode * T.m + r] = e;
    }
    if (threadIdx.x == 0) {
        E.age[newnode] = ageNew;
        T.len[newnode] = nSave;
        E.visJ[newnode] = (int32_t) ~keys[0].nj;            // visible[newnode] = hits[newnode][0]
        E.visD[newnode] = distL[keys[0].src];
    }
    __syncthreads();
    vft_nj_update_top_visible<REAL, NC>(A, E, s, tv, newnode, (int32_t) ~keys[0].nj, distL[keys[0].src], sW, sT, staleList, redC, redT);
    // updateVisible (NJ.tcc:4633-4657) over the saved hits in order.  The getVisible of hit t looks at hit.j's own visible hit,
    // which only iteration t changes: all tests are made first (with their lazy refreshes), then the few hits that pass update
    // visible[] and the top-visible list one after the other.
    for (int r = threadIdx.x; r < nSave; r += blockDim.x) {
        int32_t vj;
        const int32_t node = (int32_t) ~keys[r].nj;
        if (!vft_nj_visible_ok(A, E, node, vj)) continue;
        if ((long long) A.nOutActive[node] - nActive > s.nDiffAllow) staleList[atomicAdd(&nStale, 1)] = node;
        if ((long long) A.nOutActive[vj] - nActive > s.nDiffAllow) staleList[atomicAdd(&nStale, 1)] = vj;
    }
    __syncthreads();
    vft_nj_refresh_listed<REAL, NC>(A, s, staleList, nStale, sW, sT);
    __syncthreads();
    for (int r = threadIdx.x; r < nSave; r += blockDim.x) {
        int32_t vj;
        const int32_t node = (int32_t) ~keys[r].nj;
        const REAL hitCrit = T.stC[keys[r].src];
        bool pass = true;
        if (vft_nj_visible_ok(A, E, node, vj)) pass = hitCrit < vft_nj_crit<REAL>(A, E.visD[node], node, vj, nActive);
        if (pass) passList[atomicAdd(&nPass, 1)] = r;
    }
    __syncthreads();
    const int np = nPass;
    // (in list order: a selection sort over the few entries by thread 0)
    if (threadIdx.x == 0)
        for (int a = 0; a < np; a++)
            for (int b = a + 1; b < np; b++)
                if (passList[b] < passList[a]) {
                    const int32_t x = passList[a];
                    passList[a] = passList[b];
                    passList[b] = x;
                }
    __syncthreads();
    for (int a = 0; a < np; a++) {
        const int r = passList[a];
        const int32_t node = (int32_t) ~keys[r].nj;
        const REAL d = distL[keys[r].src];
        if (threadIdx.x == 0) {
            E.visJ[node] = newnode;
            E.visD[node] = d;
        }
        __syncthreads();
        vft_nj_update_top_visible<REAL, NC>(A, E, s, tv, node, newnode, d, sW, sT, staleList, redC, redT);
    }
    for (int t = threadIdx.x; t < E.nTop; t += blockDim.x) E.topvis[t] = tv[t];
    if (threadIdx.x == 0) {
        st->nUnique = nU;
        st->joinsDone = joinIndex + 1;
        vft_nj_publish(E, st);
    }
}
