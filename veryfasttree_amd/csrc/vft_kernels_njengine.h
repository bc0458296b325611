// The join loop of fastNJ with top hits (NJ.tcc:2857-3047: topHitNJSearch :4137-4262, the join body :2897-3042, topHitJoin
// :4306-4438) as a stream of kernels that take their arguments from a device-resident state block: the host enqueues the
// same launches for every join without waiting for any of them, reads the join records later, and only steps in for the
// events the kernels flag - a top-visible reset, a top-hits refresh, a hill-climbing that needs another round.
//
// Round 2's loop made three host round trips per join (two list walks, one merge), with the host walking 1 000-entry lists
// in between: 377 us per join at a million sequences.  Here the visible set (visible[] / topvisible[], NJ.h:236-246), the
// ages of the lists and the scalars of the loop (nActive, maxnode, totdiam, the candidate join) live next to the lists
// (vft_kernels_tophits.h) and a join is the launch sequence
//     k_nj_best_pairs(i)     getBestFromTopHits for the first end: one workgroup per list entry       (:4267-4298)
//     k_nj_glue_best         its result against the candidate; setOutDistance of the (new) second end   (:4226-4240)
//     k_nj_best_pairs(j)     getBestFromTopHits for the second end
//     k_nj_glue_join         its result against the candidate (:4243-4260); the join itself: criterion on fresh
//                            out-distances, tree arrays, branch lengths, profile, out-profile (:2897-3042); the new node's
//                            out-distance
//     k_nj_merge_pairs       uniqueBestHits of the children's lists: one workgroup per entry           (:4319-4330, 4786-4833)
//     k_nj_glue_scan         the rest of topHitJoin (:4342-4438: decision, sortSaveBestHits, visible-set updates :4633-4726)
//                            AND topHitNJSearch of the next join up to the hill climbing (:4137-4223), setOutDistance of
//                            its first end
// - three wide kernels and three single-workgroup "glue" kernels per join (-fastest: no hill climbing, three kernels).
// Every kernel starts by looking at state->halt and does nothing when an earlier kernel has raised it; the host resumes
// after it has dealt with the event.  Results are bit-identical to the host-driven loop (same formulas, same order of every
// floating-point operation; -ffp-contract=off).
#pragma once
#include "vft_kernels_tophits.h"

// tools-only build (-DVFT_NJ_TIMING): thread 0 of k_nj_glue_scan accumulates the clock ticks between its phases
#ifdef VFT_NJ_TIMING
__device__ unsigned long long vftNjTicks[16];
#define VFT_NJ_TICK(k)                                                                  \
    do {                                                                                \
        if (threadIdx.x == 0) {                                                         \
            const unsigned long long now_ = wall_clock64();                             \
            atomicAdd(&vftNjTicks[k], now_ - tick_);                                    \
            tick_ = now_;                                                               \
        }                                                                               \
    } while (0)
#else
#define VFT_NJ_TICK(k) do { } while (0)
#endif
#define VFT_NJ_TAIL 256    // threads of k_nj_glue_scan: its cost is barriers and dependent loads, not arithmetic - four wavefronts
                           // synchronise several times faster than sixteen (measured: 1024 threads 47 us per call at 20 000 taxa)

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
    int32_t nUnique, runRound;   // runRound: the hill-climbing round whose kernels come next executes (NJ.tcc:4226-4262: "while (changed)")
    // the speculative double walk (k_nj_best_pairs2): setOutDistance(join.j) computed ahead but not stored yet
    REAL specOut;
    int32_t specStamp, specValid, logCount;
    // the merge that follows a join (written by k_nj_glue_join): the new node, its children in id order, their lists' lengths
    int32_t mergeNew, mergeC0, mergeC1, mergeN0, mergeN1, mergeAge;   // mergeAge: (age[c0] + age[c1] + 1) / 2 + 1, NJ.tcc:4342-4345
};

template <typename REAL>
struct NjEngine {
    NjState<REAL> *st;
    int32_t *visJ;          // visible[node].j, visible[node].dist (NJ.h:236-240)
    REAL *visD;
    int32_t *topvis;        // topvisible[nTop] (NJ.h:242-246)
    int32_t *age;           // TopHitsList::age
    NjJoinRec *logDev, *logHost;
    volatile long long *hostStatus;   // host-mapped status word (vft_nj_publish)
    int32_t m, nTop, need, ageLimit, fastest, staleStamp;
    double staleOutLimit, tol;
    REAL *stash;            // vft_join_fused's pending stash
    int64_t *pendIds;
    unsigned int *refClaim; // [maxNodes] one writer per refreshed node in a speculative double walk
    int32_t *logNode, *logStamp;   // [m] what the walk of the second end refreshed (undone when the first walk changes the candidate)
    REAL *logOut;
    // what k_nj_glue_scan wants to know about the slots of the top-visible list and about the merge's candidates, gathered by
    // the wide kernel in front of it (k_nj_merge_rank): slot t resp. rank r at [field * stride + index]
    int32_t *slotI;         // [6][nTopPad]: node, parent[node], visible[node].j, stamp(node), parent[partner], stamp(partner)
    REAL *slotR;            // [3][nTopPad]: visible[node].dist, outDist[node], outDist[partner]
    int32_t *candI;         // [5][capPad]: partner j of the candidate of rank r; for r < m: visible[j].j, stamp(j), parent[v], stamp(v)
    REAL *candR;            // [5][capPad]: distance, criterion; for r < m: visible[j].dist, outDist[j], outDist[v]
    int32_t nTopPad, capPad;
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

// Loads of state that a thread of the SAME workgroup may have rewritten earlier in the kernel (out-distances refreshed by
// thread 0, visible[] entries): served by the L2, never by a vector-L1 line cached before the write.
template <typename T>
__device__ __forceinline__ T vft_nj_ld(const T *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// the same value without storing it (thread 0 gets it); every thread calls
template <typename REAL, int NC>
__device__ __forceinline__ REAL vft_nj_out_value(const Arena<REAL> &A, const SweepArgs &s, int64_t v, double *sW, double *sT) {
    REAL d, w;
    vft_pair_block<REAL, NC>(A, v, -1, true, sW, sT, d, w);
    REAL od = 0;
    if (threadIdx.x == 0) od = vft_out_distance<REAL>(d, w, s.nActive, A.selfweight[v], A.selfdist[v], A.diameter[v], s.totdiam);
    __syncthreads();
    return od;
}

// the lazy refreshes of a batch of setCriterion calls, single-workgroup kernels: `list` (LDS, n entries, duplicates allowed)
// holds nodes some thread found staler than allowed; each is looked at again and refreshed once
template <typename REAL, int NC>
__device__ __forceinline__ void vft_nj_refresh_listed(const Arena<REAL> &A, const SweepArgs &s, const int32_t *list, int n,
                                                      double *sW, double *sT) {
    __shared__ int njGo;
    for (int k = 0; k < n; k++) {
        const int32_t v = list[k];
        __syncthreads();
        if (threadIdx.x == 0) njGo = (long long) vft_nj_ld(&A.nOutActive[v]) - s.nActive > s.nDiffAllow;   // one verdict for the workgroup
        __syncthreads();
        if (!njGo) continue;
        vft_nj_out_distance<REAL, NC>(A, s, v, sW, sT);
    }
}

// getVisible's test (NJ.tcc:546-557): node active with an active visible partner
template <typename REAL>
__device__ __forceinline__ bool vft_nj_visible_ok(const Arena<REAL> &A, const NjEngine<REAL> &E, int32_t node, int32_t &vj) {
    vj = -1;
    if (node < 0 || A.parent[node] >= 0) return false;
    vj = vft_nj_ld(&E.visJ[node]);
    return vj >= 0 && A.parent[vj] < 0;
}

template <typename REAL>
__device__ __forceinline__ REAL vft_nj_crit(const Arena<REAL> &A, REAL dist, int32_t i, int32_t j, long long nActive) {
    return vft_criterion<REAL>(dist, vft_nj_ld(&A.outDist[i]), vft_nj_ld(&A.nOutActive[i]), vft_nj_ld(&A.outDist[j]),
                               vft_nj_ld(&A.nOutActive[j]), nActive);
}

template <typename REAL>
__device__ __forceinline__ bool vft_nj_stale(const Arena<REAL> &A, const SweepArgs &s, int32_t v) {
    return (long long) vft_nj_ld(&A.nOutActive[v]) - s.nActive > s.nDiffAllow;
}

template <typename REAL>
__device__ __forceinline__ void vft_nj_publish(const NjEngine<REAL> &E, const NjState<REAL> *st) {
    // ONE 8-byte word, written in one piece (the host polls it): bits 0-30 joins completed, 31-33 halt reason, 34-63 its join.
    // No fence: the store drains by itself within the kernel's lifetime, and what the host reads once it has seen the word
    // (the join log) was written by earlier kernels.  (Two system fences here were 6 us of every join.)
    const unsigned long long w = ((unsigned long long) st->joinsDone & 0x7FFFFFFFull) | ((unsigned long long) (st->halt & 7) << 31) |
                                 ((unsigned long long) (uint32_t) st->haltJoin << 34);
    E.hostStatus[0] = (long long) w;
}

// (criterion, slot) reduction over the workgroup: the first minimum in slot order (MAXLAST = false; no candidate: slot
// 0x7FFFFFFF) or the last maximum (MAXLAST = true; no candidate: slot -1).  Wave shuffles, then one LDS exchange: three barriers
// instead of log2(threads).  Every thread calls; every thread gets the result.  redC / redT: blockDim.x / 64 entries.
template <bool MAXLAST>
__device__ __forceinline__ void vft_nj_arg_reduce(double &c, int &t, double *redC, int *redT) {
    constexpr int inv = MAXLAST ? -1 : 0x7FFFFFFF;
    auto take = [&](double c2, int t2) {
        const bool better = t2 != inv && (t == inv || (MAXLAST ? (c2 > c || (c2 == c && t2 > t)) : (c2 < c || (c2 == c && t2 < t))));
        if (better) {
            c = c2;
            t = t2;
        }
    };
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double c2 = __shfl_xor(c, off, 64);
        const int t2 = __shfl_xor(t, off, 64);
        take(c2, t2);
    }
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        redC[wave] = c;
        redT[wave] = t;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        c = (int) threadIdx.x < nw ? redC[threadIdx.x] : 0.0;
        t = (int) threadIdx.x < nw ? redT[threadIdx.x] : inv;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double c2 = __shfl_xor(c, off, 64);
            const int t2 = __shfl_xor(t, off, 64);
            take(c2, t2);
        }
        if (threadIdx.x == 0) {
            redC[0] = c;
            redT[0] = t;
        }
    }
    __syncthreads();
    c = redC[0];
    t = redT[0];
    __syncthreads();
}

// two first-minimum reductions at once (the two walks of a speculative hill-climbing round): redC / redT hold 2 x waves entries
__device__ __forceinline__ void vft_nj_arg_reduce2(double &c0, int &t0, double &c1, int &t1, double *redC, int *redT) {
    constexpr int inv = 0x7FFFFFFF;
    auto take = [](double &c, int &t, double c2, int t2) {
        if (t2 != inv && (t == inv || c2 < c || (c2 == c && t2 < t))) {
            c = c2;
            t = t2;
        }
    };
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double a = __shfl_xor(c0, off, 64), b = __shfl_xor(c1, off, 64);
        const int ta = __shfl_xor(t0, off, 64), tb = __shfl_xor(t1, off, 64);
        take(c0, t0, a, ta);
        take(c1, t1, b, tb);
    }
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        redC[2 * wave] = c0;
        redT[2 * wave] = t0;
        redC[2 * wave + 1] = c1;
        redT[2 * wave + 1] = t1;
    }
    __syncthreads();
    c0 = c1 = 0.0;
    t0 = t1 = inv;
    for (int w = 0; w < nw; w++) {   // (a handful of waves: every thread folds them in order)
        take(c0, t0, redC[2 * w], redT[2 * w]);
        take(c1, t1, redC[2 * w + 1], redT[2 * w + 1]);
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// The top-visible list in LDS with everything getVisible computes for its slots, so that updateTopVisible and the scan of the
// next join run without touching global memory: flags bits 0-1 = 0 empty / dead node, 1 active node without a usable visible
// hit, 2 usable (node and partner active); bit 2 / 3: the node's / the partner's out-distance is staler than allowed (a
// getVisible would refresh it first).  Any refresh inside the kernel reloads the whole cache.
template <typename REAL>
struct NjSlots {
    int32_t *node, *vj, *flags;
    REAL *dist;
    double *crit;
    int *anyStale;   // some slot carries a stale flag (set when the cache is loaded)
};

// The cache for all slots.  A thread owns several slots (1 500 slots on 256 threads at a million sequences) and every slot is
// two dependent rounds of loads - everything that hangs on the node, then everything that hangs on its partner: the loads of
// VFT_NJ_BATCH slots are issued together, round by round, instead of slot after slot.  Every thread calls; ends with a barrier.
#define VFT_NJ_BATCH 8
template <typename REAL>
__device__ __forceinline__ void vft_nj_slots_load(const Arena<REAL> &A, const NjEngine<REAL> &E, const SweepArgs &s, const NjSlots<REAL> &S) {
    __syncthreads();
    if (threadIdx.x == 0) *S.anyStale = 0;
    __syncthreads();
    for (int base = 0; base < E.nTop; base += VFT_NJ_BATCH * (int) blockDim.x) {
        int32_t node[VFT_NJ_BATCH], pn[VFT_NJ_BATCH], vj[VFT_NJ_BATCH], si[VFT_NJ_BATCH], pj[VFT_NJ_BATCH], sj[VFT_NJ_BATCH];
        REAL d0[VFT_NJ_BATCH], oi[VFT_NJ_BATCH], oj[VFT_NJ_BATCH];
#pragma unroll
        for (int k = 0; k < VFT_NJ_BATCH; k++) {
            const int t = base + k * (int) blockDim.x + (int) threadIdx.x;
            node[k] = t < E.nTop ? S.node[t] : -1;
            pn[k] = 0;
            vj[k] = -1;
            si[k] = 0;
            d0[k] = oi[k] = 0;
            if (node[k] >= 0) {
                pn[k] = A.parent[node[k]];
                vj[k] = vft_nj_ld(&E.visJ[node[k]]);
                d0[k] = vft_nj_ld(&E.visD[node[k]]);
                oi[k] = vft_nj_ld(&A.outDist[node[k]]);
                si[k] = vft_nj_ld(&A.nOutActive[node[k]]);
            }
        }
#pragma unroll
        for (int k = 0; k < VFT_NJ_BATCH; k++) {
            pj[k] = 0;
            sj[k] = 0;
            oj[k] = 0;
            if (node[k] >= 0 && pn[k] < 0 && vj[k] >= 0) {
                pj[k] = A.parent[vj[k]];
                oj[k] = vft_nj_ld(&A.outDist[vj[k]]);
                sj[k] = vft_nj_ld(&A.nOutActive[vj[k]]);
            }
        }
#pragma unroll
        for (int k = 0; k < VFT_NJ_BATCH; k++) {
            const int t = base + k * (int) blockDim.x + (int) threadIdx.x;
            if (t >= E.nTop) continue;
            int32_t f = 0, v = -1;
            REAL d = 0;
            double cr = 0;
            if (node[k] >= 0 && pn[k] < 0) {
                v = vj[k];
                f = 1;
                if (v >= 0 && pj[k] < 0) {
                    f = 2;
                    d = d0[k];
                    cr = (double) vft_criterion<REAL>(d, oi[k], si[k], oj[k], sj[k], s.nActive);
                    if ((long long) si[k] - s.nActive > s.nDiffAllow) f |= 4;
                    if ((long long) sj[k] - s.nActive > s.nDiffAllow) f |= 8;
                }
            }
            S.vj[t] = v;
            S.flags[t] = f;
            S.dist[t] = d;
            S.crit[t] = cr;
            if (f & 12) *S.anyStale = 1;   // (benign race: every writer stores 1)
        }
    }
    __syncthreads();
}

// The lazy refreshes the getVisible calls of slots [0, reach) would make (plus `extra` nodes, -1 = none), then the cache again.
// Every thread calls.  Returns (to every thread) whether anything was refreshed.
template <typename REAL, int NC>
__device__ __forceinline__ bool vft_nj_slots_refresh(const Arena<REAL> &A, const NjEngine<REAL> &E, const SweepArgs &s, const NjSlots<REAL> &S,
                                                     int reach, int32_t extra0, int32_t extra1, int32_t *staleList, double *sW, double *sT) {
    __shared__ int njStaleN;
    if (extra0 < 0 && extra1 < 0 && *S.anyStale == 0) return false;   // (uniform; the common case)
    __syncthreads();
    if (threadIdx.x == 0) {
        int n = 0;
        if (extra0 >= 0 && vft_nj_stale(A, s, extra0)) staleList[n++] = extra0;
        if (extra1 >= 0 && vft_nj_stale(A, s, extra1)) staleList[n++] = extra1;
        njStaleN = n;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < reach; t += blockDim.x) {
        const int32_t f = S.flags[t];
        if ((f & 3) != 2) continue;
        if (f & 4) staleList[atomicAdd(&njStaleN, 1)] = S.node[t];
        if (f & 8) staleList[atomicAdd(&njStaleN, 1)] = S.vj[t];
    }
    __syncthreads();
    const int n = njStaleN;
    if (n == 0) return false;
    vft_nj_refresh_listed<REAL, NC>(A, s, staleList, n, sW, sT);
    __syncthreads();
    vft_nj_slots_load(A, E, s, S);
    return true;
}

// updateTopVisible(iIn, hit) (NJ.tcc:4660-4726) on the cached list, by the whole workgroup:
//   1. the first slot that holds iIn already or a dead / empty node takes (iIn, hit);
//   2. otherwise getVisible of every slot in order until one fails (iIn takes that slot) or shows the same pair from the other
//      side (done); the lazy refreshes of those getVisible calls only happen for the slots the reference's loop reaches;
//   3. otherwise the slot with the worst criterion (the last one among equals) is replaced if the new hit is better.
// hitCrit: the criterion of (iIn, hit.j) as the merge computed it - valid while *nRefreshes (refreshes made by this kernel so
// far) is zero, recomputed otherwise.
template <typename REAL, int NC>
__device__ __forceinline__ void vft_nj_update_top_visible(const Arena<REAL> &A, const NjEngine<REAL> &E, const SweepArgs &s,
                                                          const NjSlots<REAL> &S, int32_t iIn, int32_t hitJ, REAL hitDist, REAL hitCrit,
                                                          int *nRefreshes, double *sW, double *sT, int32_t *staleList, double *redC, int *redT) {
    // ONE pass over the cached list collects what all three steps ask for - the first slot of step 1, the slot where step 2's scan
    // stops, step 3's worst slot - and one reduction delivers them (the steps as three passes were eight barriers per call,
    // 2.6 calls per join); only a refresh inside step 2 (rare) makes step 3 look again.
    __shared__ int first1, stop2;
    if (threadIdx.x == 0) {
        first1 = 0x7FFFFFFF;
        stop2 = 0x7FFFFFFF;
    }
    __syncthreads();
    int f1 = 0x7FFFFFFF, s2 = 0x7FFFFFFF, wt = -1;
    double wc = -1e20;
    for (int t = threadIdx.x; t < E.nTop; t += blockDim.x) {
        const int32_t node = S.node[t], fl = S.flags[t] & 3;
        if (node == iIn) {
            // visible[iIn] has just become `hit`: every slot that holds iIn shows it from now on (the list may hold a node more
            // than once: step 1 takes a dead slot in front of the node's own)
            S.vj[t] = hitJ;
            S.dist[t] = hitDist;
            S.crit[t] = (double) (*nRefreshes == 0 ? hitCrit : vft_nj_crit<REAL>(A, hitDist, iIn, hitJ, s.nActive));
            S.flags[t] = 2;
        }
        if ((node == iIn || fl == 0) && t < f1) f1 = t;
        if ((fl != 2 || (node == hitJ && S.vj[t] == iIn)) && t < s2) s2 = t;   // (meaningless when some slot holds iIn: step 1 ends the call)
        // "vis.criterion >= critWorst" in slot order = the largest criterion, the last one among equals
        const double c = S.crit[t];
        if (c >= wc) {
            wc = c;
            wt = t;
        }
    }
    if (f1 != 0x7FFFFFFF) atomicMin(&first1, f1);
    if (s2 != 0x7FFFFFFF) atomicMin(&stop2, s2);
    __syncthreads();
    const int fFirst = first1;
    auto takeSlot = [&](int t) {   // thread 0: slot t now shows (iIn -> hit.j); its criterion from current out-distances
        S.node[t] = iIn;
        S.vj[t] = hitJ;
        S.dist[t] = hitDist;
        S.crit[t] = (double) (*nRefreshes == 0 ? hitCrit : vft_nj_crit<REAL>(A, hitDist, iIn, hitJ, s.nActive));
        // (neither end can be staler than allowed: the merge of this join has just evaluated setCriterion on this very pair)
        S.flags[t] = 2;
    };
    if (fFirst != 0x7FFFFFFF) {
        if (threadIdx.x == 0) takeSlot(fFirst);   // (a slot that held iIn already shows its new visible hit from now on)
        __syncthreads();
        return;
    }
    const int stop = stop2;
    const bool stopOk = stop != 0x7FFFFFFF && (S.flags[stop] & 3) == 2;   // the same pair from the other side
    const int reach = stop == 0x7FFFFFFF ? E.nTop : (stopOk ? stop + 1 : stop);
    // (the final setCriterion(iIn, hit.j), made when the scan did not stop, cannot refresh anything: the merge of this join has
    //  just evaluated setCriterion on this very pair)
    const bool refreshed = vft_nj_slots_refresh<REAL, NC>(A, E, s, S, reach, -1, -1, staleList, sW, sT);
    if (refreshed) {
        if (threadIdx.x == 0) (*nRefreshes)++;
        __syncthreads();
    }
    if (stop != 0x7FFFFFFF) {
        if (!stopOk && threadIdx.x == 0) takeSlot(stop);
        __syncthreads();
        return;
    }
    if (refreshed) {   // the criteria have changed: the worst slot again
        wc = -1e20;
        wt = -1;
        for (int t = threadIdx.x; t < E.nTop; t += blockDim.x) {
            const double c = S.crit[t];
            if (c >= wc) {
                wc = c;
                wt = t;
            }
        }
    }
    vft_nj_arg_reduce<true>(wc, wt, redC, redT);   // (only step 3 needs the reduction: the two dead slots of a join end most calls in step 1)
    if (threadIdx.x == 0 && wt >= 0) {
        const REAL b = *nRefreshes == 0 ? hitCrit : vft_nj_crit<REAL>(A, hitDist, iIn, hitJ, s.nActive);
        if ((double) b < wc) takeSlot(wt);
    }
    __syncthreads();
}

// setOutDistance(v) unless its stamp is nActive (NJ.tcc:1012-1015); every thread calls, v uniform
template <typename REAL, int NC>
__device__ __forceinline__ void vft_nj_force_out_distance(const Arena<REAL> &A, const SweepArgs &s, int64_t v, double *sW, double *sT) {
    __shared__ int njForce;
    __syncthreads();
    if (threadIdx.x == 0) njForce = (long long) vft_nj_ld(&A.nOutActive[v]) != s.nActive;
    __syncthreads();
    if (njForce) vft_nj_out_distance<REAL, NC>(A, s, v, sW, sT);
}

// ---------------------------------------------------------------------------------------------------------------
// The candidates of a merge (k_nj_merge_pairs' staging arrays) in sortSaveBestHits' order: criterion ascending, ties by
// descending partner id (NJ.tcc:4535-4578 on a list in ascending partner order).  Sorting 2 m keys inside the
// single-workgroup glue kernel was the largest part of it (bitonic network: 23 us at 300 000 taxa); ranking by counting is
// 4 m^2 comparisons that spread over the chip: every workgroup holds all keys in LDS and ranks VFT_NJ_RANK_PER_WG of them,
// VFT_NJ_RANK_LANES lanes per key.  T.sorted[r] = staging index of the candidate of rank r, T.sorted[T.cap] = number of candidates.
// Dynamic LDS: P keys.
#define VFT_NJ_RANK_LANES 16
#define VFT_NJ_RANK_PER_WG (VFT_WG / VFT_NJ_RANK_LANES)
// Workgroups [0, rankBlocks) rank; workgroups beyond them gather, for k_nj_glue_scan, what it needs per slot of the top-visible
// list (a single workgroup gathering 1 500 slots x 8 words from 8 MB arrays is bound by its CU's one cache line per cycle: 20 us
// of that kernel); the lane that ranks a candidate also gathers the getVisible record updateVisible will test it against.
template <typename REAL>
__global__ __launch_bounds__(VFT_WG) void k_nj_merge_rank(Arena<REAL> A, NjEngine<REAL> E, TopHits<REAL> T, int rankBlocks) {
    extern __shared__ __attribute__((aligned(16))) double njLds[];
    const NjState<REAL> *st = E.st;
    const int32_t halt = st->halt;
    if ((int) blockIdx.x >= rankBlocks) {   // ---- slot records
        const int t = ((int) blockIdx.x - rankBlocks) * (int) blockDim.x + (int) threadIdx.x;
        const int32_t node = E.topvis[t < E.nTop ? t : E.nTop - 1];
        if (halt || t >= E.nTop) return;
        const int32_t nd = node >= 0 ? node : 0;
        int32_t pn = A.parent[nd], vj = E.visJ[nd], si = A.nOutActive[nd];
        REAL d0 = E.visD[nd], oi = A.outDist[nd];
        if (node < 0) {
            pn = 0;
            vj = -1;
            si = 0;
            d0 = oi = 0;
        }
        const bool ok = node >= 0 && pn < 0 && vj >= 0;
        const int32_t v = ok ? vj : 0;
        int32_t pj = A.parent[v], sj = A.nOutActive[v];
        REAL oj = A.outDist[v];
        if (!ok) {
            pj = 0;
            sj = 0;
            oj = 0;
        }
        const int S = E.nTopPad;
        E.slotI[t] = node;
        E.slotI[S + t] = pn;
        E.slotI[2 * S + t] = vj;
        E.slotI[3 * S + t] = si;
        E.slotI[4 * S + t] = pj;
        E.slotI[5 * S + t] = sj;
        E.slotR[t] = d0;
        E.slotR[S + t] = oi;
        E.slotR[2 * S + t] = oj;
        return;
    }
    const int n = st->mergeN0 + st->mergeN1;
    // (all loads of the staging arrays in flight together: the kernel is two memory round trips and some LDS work)
    int32_t ju[VFT_NJ_BATCH];
    REAL cu[VFT_NJ_BATCH];
#pragma unroll
    for (int q = 0; q < VFT_NJ_BATCH; q++) {   // (unconditional loads from clamped indices: see k_nj_glue_scan)
        const int u = q * (int) blockDim.x + (int) threadIdx.x, uc = u < T.cap ? u : T.cap - 1;
        ju[q] = T.stJ[uc];
        cu[q] = T.stC[uc];
    }
    if (halt) return;
    ThKey *keys = (ThKey *) njLds;
    __shared__ int nValid;
    if ((int) (blockIdx.x * VFT_NJ_RANK_PER_WG) >= n) {
        if (blockIdx.x == 0 && threadIdx.x == 0) T.sorted[T.cap] = 0;   // (two empty lists)
        return;
    }
    if (threadIdx.x == 0) nValid = 0;
    __syncthreads();
    int mine = 0;
#pragma unroll
    for (int q = 0; q < VFT_NJ_BATCH; q++) {
        const int u = q * (int) blockDim.x + (int) threadIdx.x;
        if (u >= n) continue;
        ThKey k;
        k.key = ju[q] < 0 ? ~0ull : vft_th_order(cu[q]);
        k.nj = ju[q] < 0 ? ~0u : ~(uint32_t) ju[q];
        k.src = ju[q] < 0 ? -1 : u;
        keys[u] = k;
        mine += ju[q] >= 0;
    }
    for (int u = VFT_NJ_BATCH * (int) blockDim.x + (int) threadIdx.x; u < n; u += blockDim.x) {   // (lists beyond 2 048 candidates)
        const int32_t j = T.stJ[u];
        ThKey k;
        k.key = j < 0 ? ~0ull : vft_th_order(T.stC[u]);
        k.nj = j < 0 ? ~0u : ~(uint32_t) j;
        k.src = j < 0 ? -1 : u;
        keys[u] = k;
        mine += j >= 0;
    }
    if (mine) atomicAdd(&nValid, mine);
    __syncthreads();
    const int e = (int) (blockIdx.x * VFT_NJ_RANK_PER_WG) + (int) (threadIdx.x / VFT_NJ_RANK_LANES), part = threadIdx.x % VFT_NJ_RANK_LANES;
    int rank = 0;
    ThKey ke;
    ke.key = ~0ull;
    ke.nj = ~0u;
    ke.src = -1;
    if (e < n) ke = keys[e];
    // (this candidate's distance, criterion and the getVisible record of its partner: asked for now, used after the ranking)
    const int32_t cj = ke.src >= 0 ? (int32_t) ~ke.nj : 0;
    REAL cD = 0, cC = 0, vd = 0, oi = 0;
    int32_t vj = -1, si = 0;
    if (part == 0 && ke.src >= 0) {
        cD = T.stD[e];
        cC = T.stC[e];
        vj = E.visJ[cj];
        vd = E.visD[cj];
        oi = A.outDist[cj];
        si = A.nOutActive[cj];
    }
    if (ke.src >= 0) {
        int u = part;
        for (; u + 7 * VFT_NJ_RANK_LANES < n; u += 8 * VFT_NJ_RANK_LANES) {   // eight keys of this lane's share per trip: the LDS reads go out together
            ThKey kk[8];
#pragma unroll
            for (int q = 0; q < 8; q++) kk[q] = keys[u + VFT_NJ_RANK_LANES * q];
#pragma unroll
            for (int q = 0; q < 8; q++) rank += vft_th_before(kk[q], ke) ? 1 : 0;
        }
        for (; u < n; u += VFT_NJ_RANK_LANES) rank += vft_th_before(keys[u], ke) ? 1 : 0;
    }
#pragma unroll
    for (int off = 1; off < VFT_NJ_RANK_LANES; off <<= 1) rank += __shfl_xor(rank, off, 64);
    if (part == 0 && ke.src >= 0) {
        T.sorted[rank] = e;
        const int S = E.capPad;
        E.candI[rank] = cj;
        E.candR[rank] = cD;
        E.candR[S + rank] = cC;
        if (rank < T.m) {
            const int32_t v = vj >= 0 ? vj : 0;
            int32_t pj = A.parent[v], sj = A.nOutActive[v];
            REAL oj = A.outDist[v];
            if (vj < 0) {
                pj = 0;
                sj = 0;
                oj = 0;
            }
            E.candI[S + rank] = vj;
            E.candI[2 * S + rank] = si;
            E.candI[3 * S + rank] = pj;
            E.candI[4 * S + rank] = sj;
            E.candR[2 * S + rank] = vd;
            E.candR[3 * S + rank] = oi;
            E.candR[4 * S + rank] = oj;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) T.sorted[T.cap] = nValid;
}

// ---------------------------------------------------------------------------------------------------------------
// setOutDistance's value for two nodes at once (either may be skipped): the columns of both against the out-profile are loaded
// together and four threads add the four column-ordered chains (vft_pair_block twice costs two rounds of dependent loads).
// Every thread calls; thread 0 gets the values.  LDS: sW / sT for v0, sW2 / sT2 for v1 (nPos doubles each).
template <typename REAL, int NC>
__device__ __forceinline__ void vft_nj_out_values2(const Arena<REAL> &A, const SweepArgs &s, int64_t v0, int64_t v1,
                                                   double *sW, double *sT, double *sW2, double *sT2, REAL &od0, REAL &od1, bool &need0, bool &need1) {
    // setOutDistance recomputes unless the stamp IS nActive (NJ.tcc:1012-1015) - which it almost never is here, so the columns are
    // loaded without waiting for the stamps; threads 0 / 1 fetch the stamps and the scalars of the closed form meanwhile
    __shared__ double res2[4];
    __shared__ int needS[2];
    const int64_t nPos = A.d.nPos;
    REAL selfW = 0, selfD = 0, dia = 0;
    if (threadIdx.x < 2) {
        const int64_t v = threadIdx.x ? v1 : v0;
        needS[threadIdx.x] = (long long) vft_nj_ld(&A.nOutActive[v]) != s.nActive;
        selfW = A.selfweight[v];
        selfD = A.selfdist[v];
        dia = A.diameter[v];
    }
    const bool row0 = vft_is_row<REAL>(A, v0), row1 = vft_is_row<REAL>(A, v1);
    for (int64_t p = threadIdx.x; p < nPos; p += blockDim.x) {
        Col<REAL, NC> a1, a2, b1, b2;
        vft_pair_load<REAL, NC>(A, v0, -1, true, p, a1, a2, row0, false);
        vft_pair_load<REAL, NC>(A, v1, -1, true, p, b1, b2, row1, false);
        vft_pair_addends<REAL, NC>(A, false, true, p, a1, a2, sW, sT);
        vft_pair_addends<REAL, NC>(A, false, true, p, b1, b2, sW2, sT2);
    }
    __syncthreads();
    if (threadIdx.x < 4) {   // top and denom of v0, top and denom of v1: each in column order
        const double *src = threadIdx.x == 0 ? sT : threadIdx.x == 1 ? sT2 : threadIdx.x == 2 ? sW : sW2;
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
        res2[threadIdx.x] = acc;   // [0] top(v0) [1] top(v1) [2] denom(v0) [3] denom(v1)
    }
    __syncthreads();
    REAL od = 0;
    if (threadIdx.x < 2) {
        const double top = res2[threadIdx.x], denom = res2[2 + threadIdx.x];
        const REAL w = (REAL) (denom > 0 ? denom : 0.01), d = (REAL) (denom > 0 ? top / denom : 1.0);
        od = vft_out_distance<REAL>(d, w, s.nActive, selfW, selfD, dia, s.totdiam);
    }
    od0 = od;                              // (thread 0's value is v0's)
    od1 = __shfl(od, 1, 64);               // (thread 0 gets thread 1's)
    need0 = needS[0] != 0;
    need1 = needS[1] != 0;
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// k_nj_glue_scan: [the rest of topHitJoin for join `doneJoin` (>= 0): NJ.tcc:4342-4438 - the age of the new list, the decision,
// sortSaveBestHits, the new node's visible hit, updateTopVisible for it, updateVisible over the saved hits] + [topHitNJSearch
// of join `nextJoin` (>= 0) up to the hill climbing, NJ.tcc:4137-4223: the lazy refreshes of the scan, the best visible hit,
// the reset test; then setOutDistance of the candidate's first end (both ends with -fastest, which does not climb)].
// One workgroup of THREADS threads; a single-workgroup kernel is a chain of dependent memory round trips (2-3 us each at a
// million sequences, where the node arrays do not fit the L2), so everything the kernel will want is loaded in FOUR rounds up
// front: (0) state, the join's record, the top-visible list, the ranks; (1) what hangs on a slot's node, the candidates behind
// the ranks, the children's ages; (2) what hangs on a slot's partner, getVisible of the hits that will be saved; (3) their
// partners.  A thread holds VFT_NJ_BATCH slots and ranks: nTop, P <= VFT_NJ_BATCH * THREADS (the host picks THREADS).
// Dynamic LDS: 2 x pair staging | ThKey[P] | REAL[P] | slot cache (3 int32 + REAL + double per slot) | int32[2 nTop + 2 P + 2]
// stale list | int32[P] pass list.
template <typename REAL, int NC, int THREADS>
__global__ __launch_bounds__(THREADS) void k_nj_glue_scan(Arena<REAL> A, NjEngine<REAL> E, TopHits<REAL> T, long long doneJoin,
                                                          long long nextJoin, int P) {
    extern __shared__ __attribute__((aligned(16))) double njLds[];
    constexpr int B = VFT_NJ_BATCH, TB = VFT_NJ_BATCH / 2;   // (saved hits: at most m <= P / 2 ranks)
    NjState<REAL> *st = E.st;
    double *sW = njLds, *sT = njLds + A.d.nPosPad, *sW2 = njLds + 2 * A.d.nPosPad, *sT2 = njLds + 3 * A.d.nPosPad;
    ThKey *keys = (ThKey *) (njLds + 4 * A.d.nPosPad);
    double *critS = (double *) (keys + P);
    REAL *distL = (REAL *) (critS + E.nTop), *critL = distL + P;   // by rank: distance and criterion of the sorted candidates
    NjSlots<REAL> S;
    S.crit = critS;
    S.dist = critL + P;
    S.node = (int32_t *) (S.dist + E.nTop);
    S.vj = S.node + E.nTop;
    S.flags = S.vj + E.nTop;
    int32_t *staleList = S.flags + E.nTop;
    int32_t *passList = staleList + 2 * E.nTop + 2 * P + 2;
    __shared__ int thCount, nPass, nRefreshes, nCand, slotsStale;
    S.anyStale = &slotsStale;
    __shared__ double redC[THREADS];
    __shared__ int redT[THREADS];
    const int tid = (int) threadIdx.x;
    const bool merge = doneJoin >= 0;
    // ---- the one round of loads: the state block, the join's record, and the slot / candidate records k_nj_merge_rank gathered
    // (all UNCONDITIONAL, from clamped indices, masked afterwards: a load inside a divergent branch makes the compiler wait for
    //  it before the next one is issued)
    const int32_t halt = st->halt;
    const long long nActive = st->nActive;
    const double totdiam = st->totdiam;
    const int32_t tvAge0 = st->tvAge, ageNew0 = st->mergeAge;
    NjJoinRec rec = E.logDev[merge ? doneJoin : 0];
    int nU = T.sorted[T.cap];
    if (!merge) {
        rec = NjJoinRec{};
        nU = 0;
    }
    int32_t sNode[B], sPn[B], sVj[B], sSi[B], sPj[B], sSj[B], rJ[B], tVj[TB], tSi[TB], tPj[TB], tSj[TB];
    REAL sD[B], sOi[B], sOj[B], rD[B], rC[B], tVd[TB], tOi[TB], tOj[TB];
    {
        const int S = E.nTopPad, C = E.capPad;
#pragma unroll
        for (int k = 0; k < B; k++) {
            const int t = k * THREADS + tid, ts = t < E.nTop ? t : E.nTop - 1, rs = t < T.cap ? t : T.cap - 1;
            sNode[k] = E.slotI[ts];
            sPn[k] = E.slotI[S + ts];
            sVj[k] = E.slotI[2 * S + ts];
            sSi[k] = E.slotI[3 * S + ts];
            sPj[k] = E.slotI[4 * S + ts];
            sSj[k] = E.slotI[5 * S + ts];
            sD[k] = E.slotR[ts];
            sOi[k] = E.slotR[S + ts];
            sOj[k] = E.slotR[2 * S + ts];
            rJ[k] = E.candI[rs];
            rD[k] = E.candR[rs];
            rC[k] = E.candR[C + rs];
        }
#pragma unroll
        for (int k = 0; k < TB; k++) {
            const int t = k * THREADS + tid, rs = t < T.m ? t : T.m - 1;
            tVj[k] = E.candI[C + rs];
            tSi[k] = E.candI[2 * C + rs];
            tPj[k] = E.candI[3 * C + rs];
            tSj[k] = E.candI[4 * C + rs];
            tVd[k] = E.candR[2 * C + rs];
            tOi[k] = E.candR[3 * C + rs];
            tOj[k] = E.candR[4 * C + rs];
        }
    }
    if (halt) return;
    const SweepArgs s = vft_nj_args(E, nActive, totdiam);
#ifdef VFT_NJ_TIMING
    unsigned long long tick_ = wall_clock64();
#endif
    if (tid == 0) thCount = nPass = nRefreshes = nCand = slotsStale = 0;
    const int32_t newnode = rec.newnode;
    const int nSave = nU < E.m ? nU : E.m;
#pragma unroll
    for (int k = 0; k < B; k++) {
        const int t = k * THREADS + tid;
        if (t >= E.nTop) {
            sNode[k] = -1;
            sPn[k] = 0;
            sVj[k] = -1;
            sSi[k] = sPj[k] = sSj[k] = 0;
            sD[k] = sOi[k] = sOj[k] = 0;
        }
        if (t >= nU) {
            rJ[k] = -1;
            rD[k] = rC[k] = 0;
        }
    }
#pragma unroll
    for (int k = 0; k < TB; k++) {
        if (k * THREADS + tid >= nSave) {
            tVj[k] = -1;
            tSi[k] = tPj[k] = tSj[k] = 0;
            tVd[k] = tOi[k] = tOj[k] = 0;
        }
    }
    // ---- the slot cache (as vft_nj_slots_load leaves it) and the candidates by rank, into LDS
    __syncthreads();
#pragma unroll
    for (int k = 0; k < B; k++) {
        const int t = k * THREADS + tid;
        if (t < E.nTop) {
            int32_t f = 0, v = -1;
            REAL d = 0;
            double cr = 0;
            if (sNode[k] >= 0 && sPn[k] < 0) {
                v = sVj[k];
                f = 1;
                if (v >= 0 && sPj[k] < 0) {
                    f = 2;
                    d = sD[k];
                    cr = (double) vft_criterion<REAL>(d, sOi[k], sSi[k], sOj[k], sSj[k], nActive);
                    if ((long long) sSi[k] - nActive > s.nDiffAllow) f |= 4;
                    if ((long long) sSj[k] - nActive > s.nDiffAllow) f |= 8;
                }
            }
            S.node[t] = sNode[k];
            S.vj[t] = v;
            S.flags[t] = f;
            S.dist[t] = d;
            S.crit[t] = cr;
            if (f & 12) slotsStale = 1;   // (benign race: every writer stores 1)
        }
        if (t < nU) {
            ThKey kk;
            kk.key = 0;
            kk.nj = ~(uint32_t) rJ[k];
            kk.src = t;
            keys[t] = kk;
            distL[t] = rD[k];
            critL[t] = rC[k];
        }
    }
    __syncthreads();
    VFT_NJ_TICK(0);
    if (merge) {
        VFT_NJ_TICK(1);
        VFT_NJ_TICK(2);
        // NJ.tcc:4342-4362 (the new list's age: k_nj_glue_join formed it from the children's)
        const int32_t ageNew = ageNew0;
        const bool useUnique = (long long) nU == nActive - 1 || (ageNew <= E.ageLimit && nU >= E.need);
        if (!useUnique) {
            if (tid == 0) {
                E.age[newnode] = ageNew;
                st->nUnique = nU;
                st->joinsDone = doneJoin + 1;
                st->halt = VFT_NJ_HALT_REFRESH;
                st->haltJoin = (int32_t) doneJoin;
                vft_nj_publish(E, st);
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < TB; k++) {
            const int r = k * THREADS + tid;
            if (r < nSave) {
                ThHit<REAL> e;
                e.j = rJ[k];
                e.dist = rD[k];
                T.hits[(int64_t) newnode * T.m + r] = e;
            }
        }
        const int32_t firstJ = (int32_t) ~keys[0].nj;
        const REAL firstD = distL[0], firstC = critL[0];
        if (tid == 0) {
            E.age[newnode] = ageNew;
            T.len[newnode] = nSave;
            E.visJ[newnode] = firstJ;            // visible[newnode] = hits[newnode][0]
            E.visD[newnode] = firstD;
        }
        __syncthreads();
        VFT_NJ_TICK(3);
        vft_nj_update_top_visible<REAL, NC>(A, E, s, S, newnode, firstJ, firstD, firstC, &nRefreshes, sW, sT, staleList, redC, redT);
        VFT_NJ_TICK(4);
        // updateVisible (NJ.tcc:4633-4657) over the saved hits in order.  The getVisible of hit t looks at hit.j's own visible
        // hit, which only iteration t changes: the lazy refreshes of all of them first, then all tests, then the few hits that
        // pass update visible[] and the top-visible list one after the other.
        {
            // one pass in the common case: getVisible of every hit's partner (its visible hit, both out-distances) and the test -
            // on what rounds 2 and 3 loaded unless something has been refreshed since; only when one of those out-distances is
            // staler than allowed (rare) are they refreshed and the pass repeated
            __shared__ int nStaleV;
            // the hit's criterion as updateVisible sees it: the merge's (hits are not re-evaluated, NJ.tcc:4640-4650)
            auto test = [&](int r, int32_t node, int32_t vj, REAL vd, REAL oi, int32_t si, int32_t pj, REAL oj, int32_t sj, bool collect) {
                bool pass = true;
                if (vj >= 0 && pj < 0) {   // (node itself is active: it is a candidate of this merge)
                    const bool staleI = (long long) si - nActive > s.nDiffAllow, staleJ = (long long) sj - nActive > s.nDiffAllow;
                    if (collect) {
                        if (staleI) staleList[atomicAdd(&nStaleV, 1)] = node;
                        if (staleJ) staleList[atomicAdd(&nStaleV, 1)] = vj;
                    }
                    pass = critL[r] < vft_criterion<REAL>(vd, oi, si, oj, sj, nActive);
                }
                if (pass) passList[atomicAdd(&nPass, 1)] = r;
            };
            const bool preloaded = nRefreshes == 0;   // (uniform: written before the last barrier)
            for (int attempt = 0; attempt < 2; attempt++) {
                __syncthreads();
                if (tid == 0) nStaleV = nPass = 0;
                __syncthreads();
                if (attempt == 0 && preloaded) {
#pragma unroll
                    for (int k = 0; k < TB; k++) {
                        const int r = k * THREADS + tid;
                        if (r < nSave) test(r, rJ[k], tVj[k], tVd[k], tOi[k], tSi[k], tPj[k], tOj[k], tSj[k], true);
                    }
                } else {
                    int32_t node[TB], vj[TB], si[TB], pj[TB], sj[TB];
                    REAL vd[TB], oi[TB], oj[TB];
#pragma unroll
                    for (int k = 0; k < TB; k++) {
                        const int r = k * THREADS + tid;
                        node[k] = r < nSave ? (int32_t) ~keys[r].nj : -1;
                        vj[k] = -1;
                        si[k] = 0;
                        vd[k] = oi[k] = 0;
                        if (node[k] >= 0) {
                            vj[k] = vft_nj_ld(&E.visJ[node[k]]);
                            vd[k] = vft_nj_ld(&E.visD[node[k]]);
                            oi[k] = vft_nj_ld(&A.outDist[node[k]]);
                            si[k] = vft_nj_ld(&A.nOutActive[node[k]]);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < TB; k++) {
                        pj[k] = 0;
                        sj[k] = 0;
                        oj[k] = 0;
                        if (vj[k] >= 0) {
                            pj[k] = A.parent[vj[k]];
                            oj[k] = vft_nj_ld(&A.outDist[vj[k]]);
                            sj[k] = vft_nj_ld(&A.nOutActive[vj[k]]);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < TB; k++) {
                        const int r = k * THREADS + tid;
                        if (r < nSave) test(r, node[k], vj[k], vd[k], oi[k], si[k], pj[k], oj[k], sj[k], attempt == 0);
                    }
                }
                __syncthreads();
                if (attempt == 1 || nStaleV == 0) break;   // (uniform)
                vft_nj_refresh_listed<REAL, NC>(A, s, staleList, nStaleV, sW, sT);
                __syncthreads();
                vft_nj_slots_load(A, E, s, S);
                if (tid == 0) nRefreshes++;
                __syncthreads();
            }
        }
        __syncthreads();
        VFT_NJ_TICK(5);
        const int np = nPass;
        if (tid == 0)   // (in list order: a selection sort over the few entries)
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
            const REAL d = distL[r], cr = critL[r];
            if (tid == 0) {
                E.visJ[node] = newnode;
                E.visD[node] = d;
            }
            __syncthreads();
            vft_nj_update_top_visible<REAL, NC>(A, E, s, S, node, newnode, d, cr, &nRefreshes, sW, sT, staleList, redC, redT);
        }
        VFT_NJ_TICK(6);
        for (int t = tid; t < E.nTop; t += THREADS) E.topvis[t] = S.node[t];
        if (tid == 0) {
            st->nUnique = nU;
            st->joinsDone = doneJoin + 1;
            if (nextJoin < 0) vft_nj_publish(E, st);
        }
        __syncthreads();
    }
    VFT_NJ_TICK(7);
    if (nextJoin < 0) return;
    // ---- topHitNJSearch(nextJoin): prefetchVisible(topvisible), the first minimum in slot order, the reset test
    vft_nj_slots_refresh<REAL, NC>(A, E, s, S, E.nTop, -1, -1, staleList, sW, sT);
    double bc = 0;
    int bt = 0x7FFFFFFF, mine = 0;
    for (int t = tid; t < E.nTop; t += THREADS) {
        if ((S.flags[t] & 3) != 2) continue;
        mine++;
        const double c = S.crit[t];
        if (bt == 0x7FFFFFFF || c < bc) {
            bc = c;
            bt = t;
        }
    }
    if (mine) atomicAdd(&nCand, mine);
    vft_nj_arg_reduce<false>(bc, bt, redC, redT);
    __shared__ int scanHalt, scanI, scanJ;
    if (tid == 0) {
        const int age = tvAge0 + 1;
        st->tvAge = age;
        const long long cand = nCand;
        scanHalt = 0;
        scanI = scanJ = -1;
        if (2ll * age > E.m || (3 * cand < E.nTop && 3 * cand < nActive) || bt == 0x7FFFFFFF) {
            st->halt = VFT_NJ_HALT_RESET;
            st->haltJoin = (int32_t) nextJoin;
            scanHalt = 1;
        } else {
            const int b = bt;
            st->curI = scanI = S.node[b];
            st->curJ = scanJ = S.vj[b];
            st->curDist = S.dist[b];
            st->curCrit = (REAL) S.crit[b];
            st->changed = 0;
            st->runRound = 1;
        }
        vft_nj_publish(E, st);
    }
    __syncthreads();
    VFT_NJ_TICK(8);
    if (scanHalt) return;
    // getBestFromTopHits' setOutDistance(join.i) (NJ.tcc:4273-4279); with -fastest the two ends before the join (:2897-2898).
    // Without -fastest the hill climbing walks both ends' lists in ONE launch (k_nj_best_pairs2), i.e. the second walk starts
    // before the first one has had its say.  The reference's second walk begins with setOutDistance(join.j) - which the first
    // walk must not see (its own hit (i, j) is evaluated with j's out-distance as it was): computed here, kept in the state
    // block, stored by k_nj_glue_join once the first walk has confirmed the candidate.
    REAL odI, odJ;
    bool needI, needJ;
    const int64_t vI = scanI, vJ = scanJ;
    vft_nj_out_values2<REAL, NC>(A, s, vI, vJ, sW, sT, sW2, sT2, odI, odJ, needI, needJ);
    if (tid == 0) {
        if (needI) {
            A.outDist[vI] = odI;
            A.nOutActive[vI] = (int32_t) nActive;
            A.mOutDist[vI] = odI;
            A.mNOut[vI] = (int32_t) nActive;
        }
        if (E.fastest) {
            if (needJ) {
                A.outDist[vJ] = odJ;
                A.nOutActive[vJ] = (int32_t) nActive;
                A.mOutDist[vJ] = odJ;
                A.mNOut[vJ] = (int32_t) nActive;
            }
        } else {
            st->specValid = needJ;
            st->specOut = odJ;
            st->specStamp = (int32_t) nActive;
            st->logCount = 0;
        }
    }
    VFT_NJ_TICK(9);
}

// getBestFromTopHits for one end of the candidate (NJ.tcc:4267-4298): one workgroup per list entry (grid = m; workgroups
// beyond the list's length leave), results into the staging arrays; the glue kernel that follows picks.
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_WG) void k_nj_best_pairs(Arena<REAL> A, NjEngine<REAL> E, TopHits<REAL> T, int which) {
    extern __shared__ __attribute__((aligned(16))) double njLds[];
    const NjState<REAL> *st = E.st;
    if (st->halt || !st->runRound) return;
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

// Both walks of a hill-climbing round in one launch: workgroups [0, m) are getBestFromTopHits(join.i), [m, 2 m) are
// getBestFromTopHits(join.j) for the candidate as k_nj_glue_scan left it - speculating that the first walk will not change it
// (it does in 0.6 % of the joins).  What the second walk may not do before the first one is over is kept apart: join.j's
// forced out-distance comes from the state block (specOut), and every out-distance the second walk refreshes is logged so that
// k_nj_glue_join can undo the walk.  One writer per refreshed node (claims): the log then holds consistent (value, stamp)
// pairs.  Staging: entries [0, m) and [m, 2 m).
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_WG) void k_nj_best_pairs2(Arena<REAL> A, NjEngine<REAL> E, TopHits<REAL> T, unsigned int tag) {
    extern __shared__ __attribute__((aligned(16))) double njLds[];
    NjState<REAL> *st = E.st;
    if (st->halt || !st->runRound) return;
    const int side = (int) blockIdx.x >= T.m ? 1 : 0, t = (int) blockIdx.x - side * T.m;
    const int64_t node = side ? st->curJ : st->curI;
    const int32_t specValid = st->specValid, specStamp = st->specStamp;
    const REAL specOut = st->specOut;
    const SweepArgs s = vft_nj_args(E, st->nActive, st->totdiam);
    const int lenN = T.len[node];
    const ThHit<REAL> h = T.hits[node * T.m + t];   // (t < m: inside the node's block whatever its length; asked for together with it)
    if (t >= lenN) return;
    const int slot = side * T.m + t;
    // The common case - the partner is still active, its out-distance fresh enough - is one more round of loads by one thread:
    // everything that hangs on the list entry is asked for at once, on that guess (k_nj_merge_pairs).
    __shared__ int thFast;
    if (threadIdx.x == 0) {
        const int32_t jg = h.j;
        const int32_t pj = A.parent[jg];
        int32_t sj = __hip_atomic_load(&A.nOutActive[jg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        REAL oj = __hip_atomic_load(&A.outDist[jg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int32_t si = A.nOutActive[node];
        REAL oi = A.outDist[node];
        if (side && specValid) {
            si = specStamp;
            oi = specOut;
        }
        const bool fast = pj < 0 && jg != (int32_t) node && !((int64_t) sj - s.nActive > s.nDiffAllow);
        if (fast) {
            if ((int64_t) sj == s.nActive) {   // (a stamp of this very step: the value again, now certainly after the stamp - vft_th_pair)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                oj = __hip_atomic_load(&A.outDist[jg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            T.stJ[slot] = jg;
            T.stD[slot] = h.dist;
            T.stC[slot] = vft_criterion<REAL>(h.dist, oi, si, oj, sj, s.nActive);
        }
        thFast = fast;
    }
    __syncthreads();
    if (thFast) return;
    // (the slow kind - some entry of a list usually is - decides how long the launch lasts: vft_th_pair with the rows known by id)
    const int32_t j = vft_active_ancestor(A.parent, h.j);
    if (j < 0 || j == (int32_t) node) {
        if (threadIdx.x == 0) T.stJ[slot] = -1;
        return;
    }
    ThPairSpec<REAL> spec;
    spec.ovOut = side && specValid ? &st->specOut : nullptr;
    spec.ovStamp = side && specValid ? &st->specStamp : nullptr;
    spec.claim = E.refClaim;
    spec.tag = tag;
    spec.logNode = side ? E.logNode : nullptr;
    spec.logOut = side ? E.logOut : nullptr;
    spec.logStamp = side ? E.logStamp : nullptr;
    spec.logCount = side ? &st->logCount : nullptr;
    spec.rowsById = true;
    REAL d = h.dist, cr = (REAL) 1e20;
    vft_th_pair<REAL, NC>(A, s, node, j, j != h.j, njLds, njLds + A.d.nPosPad, d, cr, &spec);
    if (threadIdx.x == 0) {
        T.stJ[slot] = j;
        T.stD[slot] = d;
        T.stC[slot] = cr;
    }
}

// the end of getBestFromTopHits and the comparison of the hill climbing (NJ.tcc:4226-4260): the first strict minimum in list
// order; "if (best.j != join.<other end> && best.criterion < join.criterion) join = best".  Every thread calls; thread 0 updates
// the state.  which == 0 opens a round ("changed = false").
template <typename REAL>
__device__ __forceinline__ void vft_nj_best_pick(const NjEngine<REAL> &E, const TopHits<REAL> &T, int which, double *redC, int *redT, int *out,
                                                 int base = 0, bool dry = false) {
    // base: where this walk's staging entries start; dry: only report whether the walk would change the candidate (out[0])
    NjState<REAL> *st = E.st;
    const int64_t node = which ? st->curJ : st->curI;
    const int n = T.len[node];
    double bc = 1e20;
    int bt = 0x7FFFFFFF;
    for (int b0 = 0; b0 < n; b0 += VFT_NJ_BATCH * (int) blockDim.x) {   // (the loads of a batch are in flight together)
        int32_t jb[VFT_NJ_BATCH];
        REAL cb[VFT_NJ_BATCH];
#pragma unroll
        for (int q = 0; q < VFT_NJ_BATCH; q++) {
            const int u0 = b0 + q * (int) blockDim.x + (int) threadIdx.x;
            jb[q] = u0 < n ? T.stJ[base + u0] : -1;
            cb[q] = u0 < n ? T.stC[base + u0] : (REAL) 0;   // (meaningless where the partner is < 0)
        }
#pragma unroll
        for (int q = 0; q < VFT_NJ_BATCH; q++) {   // ascending list positions: the first strict minimum
            const int u = base + b0 + q * (int) blockDim.x + (int) threadIdx.x;
            if (jb[q] < 0) continue;
            if ((bt == 0x7FFFFFFF && cb[q] < (REAL) 1e20) || (bt != 0x7FFFFFFF && (double) cb[q] < bc)) {
                bc = (double) cb[q];
                bt = u;
            }
        }
    }
    vft_nj_arg_reduce<false>(bc, bt, redC, redT);
    if (threadIdx.x == 0) {
        int changed = which == 0 ? 0 : st->changed;
        int32_t ci = st->curI, cj = st->curJ;
        const int b = bt;
        if (b != 0x7FFFFFFF) {
            const int32_t bj = T.stJ[b];
            const REAL bcr = T.stC[b];
            const int32_t other = which ? ci : cj;
            if (bj != other && bcr < st->curCrit) {
                changed = 1;
                if (!dry) {
                    ci = (int32_t) node;
                    cj = bj;
                    st->curI = ci;
                    st->curJ = cj;
                    st->curDist = T.stD[b];
                    st->curCrit = bcr;
                }
            }
        }
        if (!dry) st->changed = changed;
        out[0] = changed;   // (to the other threads through LDS: they may hold the state's cache line from the kernel's start)
        out[1] = ci;
        out[2] = cj;
    }
    __syncthreads();
}

// k_nj_glue_best: the first half of a hill-climbing round is over (k_nj_best_pairs(0)): its result against the candidate, then
// getBestFromTopHits' setOutDistance for the - possibly new - second end.  One workgroup of VFT_WG threads.
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_WG) void k_nj_glue_best(Arena<REAL> A, NjEngine<REAL> E, TopHits<REAL> T) {
    extern __shared__ __attribute__((aligned(16))) double njLds[];
    NjState<REAL> *st = E.st;
    if (st->halt || !st->runRound) return;
    __shared__ double redC[VFT_WG];
    __shared__ int redT[VFT_WG];
    __shared__ int picked[3];
    const SweepArgs s = vft_nj_args(E, st->nActive, st->totdiam);
    vft_nj_best_pick<REAL>(E, T, 0, redC, redT, picked);
    vft_nj_force_out_distance<REAL, NC>(A, s, picked[2], njLds, njLds + A.d.nPosPad);
}

// k_nj_glue_join: the second half of the round (k_nj_best_pairs(1)) against the candidate; when the round changed it, another
// round follows (its first setOutDistance here; raised as an event when no further round has been enqueued: lastRound).
// Otherwise the join itself (NJ.tcc:2897-3042): criterion on the fresh out-distances of both ends, tree arrays, branch
// lengths, diameter, the new profile, its self distance, the incremental out-profile (vft_join_body), totdiam, the counters;
// and - unless the caller recomputes the out-profile first (updateOut == 0) - the new node's out-distance, which the first
// setCriterion of topHitJoin would compute.  One workgroup of VFT_WG_PROF threads; dynamic LDS: 4 * nPosPad doubles.
template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_WG_PROF) void k_nj_glue_join(Arena<REAL> A, NjEngine<REAL> E, TopHits<REAL> T, long long joinIndex,
                                                              int32_t updateOut, int32_t slot, int32_t lastRound, int32_t speculative) {
    extern __shared__ __attribute__((aligned(16))) double njLds[];
    NjState<REAL> *st = E.st;
    // (the state block in one round of loads)
    const int32_t halt = st->halt, runRound = st->runRound;
    const long long nActive = st->nActive;
    const double totdiam0 = st->totdiam;
    const int64_t newn = st->maxnode;
    const REAL curDist0 = st->curDist;
    int64_t i = st->curI, j = st->curJ;
    if (halt) return;
    __shared__ double redC[VFT_WG_PROF];
    __shared__ int redT[VFT_WG_PROF];
    __shared__ REAL sDiam;
    const SweepArgs s = vft_nj_args(E, nActive, totdiam0);
    __shared__ int picked[3];
    if (!E.fastest && runRound) {
        bool roundChanged;
        if (speculative) {
            // k_nj_best_pairs2 walked both lists at once: both walks' first minima in one pass over the staging arrays (loads of
            // both sides in flight together, one reduction for the two), then thread 0 takes the reference's decisions in the
            // reference's order: did the first walk leave the candidate alone?  If so, does the second?
            __shared__ int pickLen[2], pickBest[2];
            if (threadIdx.x < 2) pickLen[threadIdx.x] = T.len[threadIdx.x ? j : i];
            __syncthreads();
            const int n0 = pickLen[0], n1 = pickLen[1];
            double bc0 = 1e20, bc1 = 1e20;
            int bt0 = 0x7FFFFFFF, bt1 = 0x7FFFFFFF;
            for (int b0 = 0; b0 < (n0 > n1 ? n0 : n1); b0 += (VFT_NJ_BATCH / 2) * (int) blockDim.x) {
                int32_t ja[VFT_NJ_BATCH / 2], jb[VFT_NJ_BATCH / 2];
                REAL ca[VFT_NJ_BATCH / 2], cb[VFT_NJ_BATCH / 2];
#pragma unroll
                for (int q = 0; q < VFT_NJ_BATCH / 2; q++) {   // (unconditional loads from clamped indices: see k_nj_glue_scan)
                    const int u = b0 + q * (int) blockDim.x + (int) threadIdx.x;
                    const int ua = u < n0 ? u : 0, ub = u < n1 ? u : 0;
                    ja[q] = T.stJ[ua];
                    ca[q] = T.stC[ua];
                    jb[q] = T.stJ[T.m + ub];
                    cb[q] = T.stC[T.m + ub];
                }
#pragma unroll
                for (int q = 0; q < VFT_NJ_BATCH / 2; q++) {
                    const int u = b0 + q * (int) blockDim.x + (int) threadIdx.x;
                    if (u >= n0) ja[q] = -1;
                    if (u >= n1) jb[q] = -1;
                }
#pragma unroll
                for (int q = 0; q < VFT_NJ_BATCH / 2; q++) {   // ascending list positions: the first strict minimum of each walk
                    const int u = b0 + q * (int) blockDim.x + (int) threadIdx.x;
                    if (ja[q] >= 0 && ((bt0 == 0x7FFFFFFF && ca[q] < (REAL) 1e20) || (bt0 != 0x7FFFFFFF && (double) ca[q] < bc0))) {
                        bc0 = (double) ca[q];
                        bt0 = u;
                    }
                    if (jb[q] >= 0 && ((bt1 == 0x7FFFFFFF && cb[q] < (REAL) 1e20) || (bt1 != 0x7FFFFFFF && (double) cb[q] < bc1))) {
                        bc1 = (double) cb[q];
                        bt1 = T.m + u;
                    }
                }
            }
            vft_nj_arg_reduce2(bc0, bt0, bc1, bt1, redC, redT);
            if (threadIdx.x == 0) {
                const REAL curCrit = st->curCrit;
                // (everything thread 0 may need, asked for together)
                int32_t bj0 = -1, bj1 = -1;
                REAL cr0 = 0, cr1 = 0, d1 = 0;
                if (bt0 != 0x7FFFFFFF) {
                    bj0 = T.stJ[bt0];
                    cr0 = T.stC[bt0];
                }
                if (bt1 != 0x7FFFFFFF) {
                    bj1 = T.stJ[bt1];
                    cr1 = T.stC[bt1];
                    d1 = T.stD[bt1];
                }
                const int32_t specValid = st->specValid, specStamp = st->specStamp, nLog = st->logCount;
                const REAL specOut = st->specOut;
                int verdict = 0;   // 0 join, 1 the first walk changed the candidate (undo the second), 2 the second walk changed it
                if (bt0 != 0x7FFFFFFF && bj0 != (int32_t) j && cr0 < curCrit) {
                    verdict = 1;
                } else {
                    if (specValid) {   // confirmed: setOutDistance(join.j) as k_nj_glue_scan computed it
                        A.outDist[j] = specOut;
                        A.nOutActive[j] = specStamp;
                        A.mOutDist[j] = specOut;
                        A.mNOut[j] = specStamp;
                    }
                    int changed = 0;
                    if (bt1 != 0x7FFFFFFF && bj1 != (int32_t) i && cr1 < curCrit) {
                        changed = 1;
                        verdict = 2;
                        st->curI = (int32_t) j;
                        st->curJ = bj1;
                        st->curDist = d1;
                        st->curCrit = cr1;
                        picked[1] = (int32_t) j;
                        picked[2] = bj1;
                    }
                    st->changed = changed;
                }
                picked[0] = verdict;
                pickBest[0] = nLog;
            }
            __syncthreads();
            const int verdict = picked[0];
            if (verdict == 1) {
                // The second walk was for the wrong node.  Every out-distance it refreshed goes back to what it was (the first
                // walk's refreshes are the reference's own), the candidate stays as the scan left it, and the host enqueues the
                // round again as two separate walks.
                const int nLog = pickBest[0];
                for (int k = threadIdx.x; k < nLog; k += blockDim.x) {
                    const int32_t v = E.logNode[k];
                    A.outDist[v] = E.logOut[k];
                    A.nOutActive[v] = E.logStamp[k];
                    A.mOutDist[v] = E.logOut[k];
                    A.mNOut[v] = E.logStamp[k];
                }
                if (threadIdx.x == 0) {
                    st->halt = VFT_NJ_HALT_CLIMB;
                    st->haltJoin = (int32_t) joinIndex;
                    vft_nj_publish(E, st);
                }
                return;
            }
            if (verdict == 2) {
                i = picked[1];
                j = picked[2];
            }
            roundChanged = verdict == 2;
        } else {
            vft_nj_best_pick<REAL>(E, T, 1, redC, redT, picked);
            i = picked[1];
            j = picked[2];
            roundChanged = picked[0] != 0;
        }
        if (roundChanged) {
            vft_nj_force_out_distance<REAL, NC>(A, s, i, njLds, njLds + A.d.nPosPad);
            if (threadIdx.x == 0 && lastRound) {
                st->halt = VFT_NJ_HALT_CLIMB;
                st->haltJoin = (int32_t) joinIndex;
                vft_nj_publish(E, st);
            }
            return;   // runRound stays set: the next round's kernels execute
        }
        __syncthreads();
        if (threadIdx.x == 0) st->runRound = 0;
    }
    double totdiamNew = totdiam0;
    // setOutDistance(i), setOutDistance(j) (NJ.tcc:2897-2898) have happened: the search forces the first end (k_nj_glue_scan, or
    // this kernel when a round changed the candidate) and the second (k_nj_glue_best; with -fastest k_nj_glue_scan forces both)
    if (threadIdx.x == 0) {
        // (one round of loads: the ends' out-distances, stamps and diameters, their lists' lengths for the merge)
        const REAL dist = curDist0;
        const REAL outI = A.outDist[i], outJ = A.outDist[j], diaI = A.diameter[i], diaJ = A.diameter[j];
        const int32_t stampI = A.nOutActive[i], stampJ = A.nOutActive[j], lenI = T.len[i], lenJ = T.len[j], ageI = E.age[i], ageJ = E.age[j];
        st->mergeAge = (ageI + ageJ + 1) / 2 + 1;
        st->mergeNew = (int32_t) newn;
        st->mergeC0 = (int32_t) (i < j ? i : j);
        st->mergeC1 = (int32_t) (i < j ? j : i);
        st->mergeN0 = i < j ? lenI : lenJ;
        st->mergeN1 = i < j ? lenJ : lenI;
        const REAL crit = vft_criterion<REAL>(dist, outI, stampI, outJ, stampJ, nActive);   // criterionFresh / setDistCriterion(join)
        // NJ.tcc:2911-2916, 3003-3007 (BIONJ off: weight 1/2)
        const double distIJ = (double) dist;
        const REAL od = outI - outJ;
        const double deltaDist = (double) od / (double) (nActive - 2);
        const REAL blI = (REAL) ((distIJ + deltaDist) / 2), blJ = (REAL) ((distIJ - deltaDist) / 2);
        const double bw = 0.5;
        const REAL bi = blI + diaI, bj = blJ + diaJ;
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
        // The host reads this record as soon as it sees the join counted in the status word, which a LATER kernel of the stream
        // publishes with a plain store: the record must have left the chip before this kernel ends.  One system-scope fence on
        // the record (thread 0 of the one workgroup), not on the status word of every scan.
        __threadfence_system();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (updateOut) {   // (a full out-profile follows otherwise, and the host sets totdiam from the diameters)
            const REAL dd = diam - diaI - diaJ;
            totdiamNew = totdiam0 + (double) dd;
            st->totdiam = totdiamNew;
        }
        A.mOutDist[newn] = 0;
        A.mNOut[newn] = E.staleStamp;
    }
    __syncthreads();
    __shared__ double newOut[2];
    vft_join_body<REAL, NC>(A, i, j, newn, sDiam, E.staleStamp, nActive, updateOut, E.tol, E.stash, E.pendIds, slot, njLds,
                            updateOut ? newOut : (double *) nullptr);
    __syncthreads();
    if (threadIdx.x == 0) {
        st->maxnode = newn + 1;
        st->nActive = nActive - 1;
        if (updateOut) {
            // setOutDistance(new node) (NJ.tcc:1012-1053) from profileDist(new node, updated out-profile), whose column sums the
            // join body has just formed
            const double top = newOut[0], denom = newOut[1];
            const REAL w = (REAL) (denom > 0 ? denom : 0.01), d = (REAL) (denom > 0 ? top / denom : 1.0);
            const REAL od = vft_out_distance<REAL>(d, w, nActive - 1, A.selfweight[newn], A.selfdist[newn], sDiam, totdiamNew);
            A.outDist[newn] = od;
            A.nOutActive[newn] = (int32_t) (nActive - 1);
            A.mOutDist[newn] = od;
            A.mNOut[newn] = (int32_t) (nActive - 1);
        }
    }
}

// the new node's out-distance after the caller has recomputed the out-profile (the first setCriterion of topHitJoin refreshes
// it: its stamp is "unreasonably high")
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
__global__ __launch_bounds__(VFT_WG) void k_nj_merge_pairs(Arena<REAL> A, NjEngine<REAL> E, TopHits<REAL> T, long long joinIndex, unsigned int tag) {
    extern __shared__ __attribute__((aligned(16))) double njLds[];
    const NjState<REAL> *st = E.st;
    if (st->halt) return;
    __shared__ int thOwner;
    // (the join's record as k_nj_glue_join left it in the state block: no chain log -> lengths -> hits)
    const int64_t newnode = st->mergeNew, c0 = st->mergeC0, c1 = st->mergeC1;
    const int n0 = st->mergeN0, n1 = st->mergeN1, t = (int) blockIdx.x;
    if (t >= n0 + n1) return;
    const SweepArgs s = vft_nj_args(E, st->nActive, st->totdiam);
    const ThHit<REAL> h = t < n0 ? T.hits[c0 * T.m + t] : T.hits[c1 * T.m + (t - n0)];
    const int32_t j = vft_active_ancestor(A.parent, h.j);
    if (threadIdx.x == 0)
        thOwner = j >= 0 && j != (int32_t) newnode && __hip_atomic_exchange(&T.mark[j], tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag;
    __syncthreads();
    if (!thOwner) {
        if (threadIdx.x == 0) T.stJ[t] = -1;
        return;
    }
    // (asking for everything that hangs on the list entry at once, on the guess that the partner is still active, was tried: every
    //  workgroup then pays for a distance, owners or not, and the launch - which lasts as long as its slowest workgroup - took
    //  19.3 us instead of 15.4)
    ThPairSpec<REAL> spec{};
    spec.rowsById = true;   // (every internal node of the engine's run has its row: no flag lookup in front of the column loads)
    REAL d = 0, cr = (REAL) 1e20;
    vft_th_pair<REAL, NC>(A, s, newnode, j, true, njLds, njLds + A.d.nPosPad, d, cr, &spec);
    if (threadIdx.x == 0) {
        T.stJ[t] = j;
        T.stD[t] = d;
        T.stC[t] = cr;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// resetTopVisible (NJ.tcc:4728-4784), device part: the criterion of (node, visible[node]) for every active node with a usable
// visible hit, after the lazy refreshes the reference's getVisible calls make, as a sweep-shaped result (dist = the visible
// hit's distance, weight = its partner, criterion) from which the top-k selection of the sweeps (k_select_*) takes the best
// k under the same order the reference sorts by: criterion ascending, ties by descending position = descending node id.
// The host walks those k records (a few thousand) to fill the list.

// the distinct nodes a getVisible would refresh; grid over [0, maxnode)
template <typename REAL>
__global__ __launch_bounds__(VFT_WG) void k_nj_reset_stale(Arena<REAL> A, NjEngine<REAL> E, SweepArgs s, int64_t maxnode, unsigned int *mark,
                                                           unsigned int tag, int64_t *list, unsigned int *counter) {
    const int64_t v = (int64_t) blockIdx.x * VFT_WG + threadIdx.x;
    if (v >= maxnode || A.parent[v] >= 0) return;
    const int32_t vj = E.visJ[v];
    if (vj < 0 || A.parent[vj] >= 0) return;
    if ((long long) A.nOutActive[v] - s.nActive > s.nDiffAllow && atomicExch(&mark[v], tag) != tag) list[atomicAdd(counter, 1u)] = v;
    if ((long long) A.nOutActive[vj] - s.nActive > s.nDiffAllow && atomicExch(&mark[vj], tag) != tag) list[atomicAdd(counter, 1u)] = vj;
}

template <typename REAL>
__global__ __launch_bounds__(VFT_WG) void k_nj_reset_crit(Arena<REAL> A, NjEngine<REAL> E, SweepArgs s, int64_t maxnode, SweepOut<REAL> O,
                                                          unsigned int *nVisible) {
    const int64_t v = (int64_t) blockIdx.x * VFT_WG + threadIdx.x;
    REAL cmin = (REAL) 1e30, cmax = (REAL) -1e30;
    bool ok = false;
    if (v < maxnode) {
        REAL d = (REAL) 1e20, w = 0, cr = (REAL) 1e20;
        if (A.parent[v] < 0) {
            const int32_t vj = E.visJ[v];
            if (vj >= 0 && A.parent[vj] < 0) {
                ok = true;
                d = E.visD[v];
                w = (REAL) vj;
                cr = vft_criterion<REAL>(d, A.outDist[v], A.nOutActive[v], A.outDist[vj], A.nOutActive[vj], s.nActive);
                cmin = cmax = cr;
            }
        }
        O.dist[v] = d;
        O.weight[v] = w;
        O.crit[v] = cr;
    }
    const unsigned long long m = __ballot(ok);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(nVisible, (unsigned int) __popcll(m));
    vft_block_minmax<REAL>(cmin, cmax, O.partMin, O.partMax, (int) blockIdx.x);
}

static __global__ void k_nj_publish_u32(const unsigned int *src, unsigned int *hostDst, int n) {
    for (int t = threadIdx.x; t < n; t += blockDim.x) hostDst[t] = src[t];
    __threadfence_system();
}
