// A whole round of subtree-prune-regraft moves (SPR, NJ.tcc:6185-6404; traverseSPR with findSPRSteps :1805-1859 and
// unwindSPRStep :1861-1879, one thread, fast flavour) as ONE persistent workgroup.
//
// The reference walks the nodes in a fixed post-order; around each it tries up to four chains of at most maxSPRLength forced
// minimum-evolution NNIs, keeps the best prefix and unwinds the rest.  Every step is tiny - two averages and six profile
// distances over a few hundred to a few thousand columns - and depends on the one before through the tree, so the host-driven
// walk (MLLengths::doSPR) was one host<->device round trip per step: ~47 us each, 300 000 steps for two rounds on 10 000 taxa,
// three quarters of the whole pipeline once the NNI rounds ran as lanes.  Here the tree (parent / child arrays, the up-profile
// cache flags) lives in device memory and wave 0 of the workgroup IS the walk: it runs the reference's control flow, and hands the
// column work to the whole workgroup as commands through LDS -
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
#define VFT_SPR_CHAIN 48          // averages queued per command
#define VFT_SPR_MAXLEN 16         // chain length the kernel is built for (the reference's default maxSPRLength is 10)

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
                                  // [4] / [5] / [6] clock ticks (100 MHz) in average commands / distance commands / in all
};

struct SprCmd {
    int32_t type, n;              // 0: stop, 1: averages (n ops), 2: distances
    int32_t out[VFT_SPR_CHAIN], a[VFT_SPR_CHAIN], b[VFT_SPR_CHAIN];
    int32_t q[4];
    double dist[6];
    double sum[12];
    // wave 0's chain bookkeeping (LDS instead of scratch: the arrays are indexed at run time)
    int32_t stepA[VFT_SPR_MAXLEN], stepB[VFT_SPR_MAXLEN];
    double stepDelta[VFT_SPR_MAXLEN];
};

// One column of a node for the SPR walk.  byId: every internal id (>= nSeqs: nodes and up-profile slots) is a plain row - true
// in the refinement phase once vft_set_profile_rows has copied the tree over - so neither the row flag nor the "is this a
// vector?" test stands between the thread and its loads: weight, code and vector are requested together (one memory round
// instead of three; the vector is simply ignored under a code).
template <typename REAL, int NC>
__device__ __forceinline__ void vft_spr_load(const Arena<REAL> &A, int64_t node, int64_t p, bool byId, Col<REAL, NC> &c) {
    if (node >= A.d.nSeqs && (byId || (A.mlIs != nullptr && A.mlIs[node - A.d.nSeqs]))) {
        const int64_t idx = (node - A.d.nSeqs) * A.d.nPos + p;
        const REAL *src = A.mlF + idx * NC;
        REAL f[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) f[k] = src[k];
        c.w = A.mlW[idx];
        c.code = (int) A.mlC[idx];
        c.vec = c.w > 0 && c.code == VFT_NOCODE_;
#pragma unroll
        for (int k = 0; k < NC; k++) c.f[k] = f[k];
        return;
    }
    vft_load_col<REAL, NC>(A, node, p, c);
}

// the data half of a command, executed by every thread of the workgroup (one copy in the binary: the walk calls it from
// many places)
template <typename REAL, int NC>
__device__ __noinline__ void vft_spr_exec(const Arena<REAL> &A, const SprState &S, SprCmd &cmd, double *pwLds) {
    const bool byId = S.rowsById != 0;
    if (cmd.type == 1) {
        const int n = cmd.n;
        for (int64_t p = threadIdx.x; p < A.d.nPos; p += VFT_SPR_WG) {
            int64_t prevOut = -1;
            Col<REAL, NC> prev;
            prev.w = 0;
            prev.code = VFT_NOCODE_;
            prev.vec = false;
#pragma unroll
            for (int q = 0; q < NC; q++) prev.f[q] = 0;
            for (int k = 0; k < n; k++) {
                const int64_t a = cmd.a[k], b = cmd.b[k], o = cmd.out[k];
                Col<REAL, NC> c1, c2;
                if (a == prevOut) c1 = prev;
                else vft_spr_load<REAL, NC>(A, a, p, byId, c1);
                if (b == prevOut) c2 = prev;
                else vft_spr_load<REAL, NC>(A, b, p, byId, c2);
                REAL wo, f[NC];
                int co;
                vft_average_col<REAL, NC>(A, c1, c2, 0.5, S.tol, wo, co, f);
                {   // the row's vector slot is written whatever the column holds: the loads above never test before they read
                    const int64_t idx = (o - A.d.nSeqs) * A.d.nPos + p;
                    A.mlW[idx] = wo;
                    A.mlC[idx] = (uint8_t) co;
                    REAL *dst = A.mlF + idx * NC;
#pragma unroll
                    for (int q = 0; q < NC; q++) dst[q] = f[q];
                }
                prevOut = o;
                prev.w = wo;
                prev.code = co;
                prev.vec = wo > 0 && co == VFT_NOCODE_;
#pragma unroll
                for (int q = 0; q < NC; q++) prev.f[q] = f[q];
            }
        }
        // every output is a row from now on (the flags of internal nodes are set already in this phase; up-profile slots get
        // theirs here); other threads read them only after the barrier that ends the command
        if (threadIdx.x < (unsigned) n && A.mlIs) A.mlIs[cmd.out[threadIdx.x] - A.d.nSeqs] = 1;
    } else if (cmd.type == 2) {
        // The six distances AB AC AD BC BD CD of (A, B, C, D) = q[0..3] (chooseNNI, NJ.tcc:4836-4846).  A wavefront per pair
        // (vft_pair_wave) walks nPos / 64 dependent memory rounds - 16 us of a step at 1 000 columns.  Instead every thread
        // loads ITS columns of the four profiles once and parks the addends of all six pairs in LDS (vft_pair_addends, exactly
        // as the pair kernels form them); then twelve lanes on eight wavefronts add the twelve chains in column order.
        // Leaf x leaf pairs of a matrix-free nucleotide run are seqDist's integer counts (any order): vft_pair_wave's fast path.
        const int64_t nPos = A.d.nPos, nPosPad = A.d.nPosPad;
        int64_t id[4];
        bool leaf[4];
#pragma unroll
        for (int x = 0; x < 4; x++) {
            id[x] = cmd.q[x];
            leaf[x] = id[x] < A.d.nSeqs;
        }
        const bool counts = !A.dmDist && NC == 4;   // leaf pairs by integer counts
        for (int64_t p = threadIdx.x; p < nPos; p += VFT_SPR_WG) {
            if constexpr (NC == 4) {   // the four columns at once: 4 loads for 6 pairs
                Col<REAL, NC> c[4];
#pragma unroll
                for (int x = 0; x < 4; x++) vft_spr_load<REAL, NC>(A, id[x], p, byId, c[x]);
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const int x = k < 3 ? 0 : k < 5 ? 1 : 2, y = k == 0 ? 1 : (k == 1 || k == 3) ? 2 : 3;
                    const bool ll = leaf[x] && leaf[y];
                    if (ll && counts) continue;
                    vft_pair_addends<REAL, NC>(A, ll, false, p, c[x], c[y], pwLds + (int64_t) (2 * k) * nPosPad, pwLds + (int64_t) (2 * k + 1) * nPosPad);
                }
            } else {                   // 20-state columns: two in registers at a time
#pragma unroll 1
                for (int k = 0; k < 6; k++) {
                    const int x = k < 3 ? 0 : k < 5 ? 1 : 2, y = k == 0 ? 1 : (k == 1 || k == 3) ? 2 : 3;
                    const bool ll = id[x] < A.d.nSeqs && id[y] < A.d.nSeqs;
                    Col<REAL, NC> c1, c2;
                    vft_spr_load<REAL, NC>(A, id[x], p, byId, c1);
                    vft_spr_load<REAL, NC>(A, id[y], p, byId, c2);
                    vft_pair_addends<REAL, NC>(A, ll, false, p, c1, c2, pwLds + (int64_t) (2 * k) * nPosPad, pwLds + (int64_t) (2 * k + 1) * nPosPad);
                }
            }
        }
        __syncthreads();
        {   // chain ch = 2 k + which (0: the pair's weights sW -> denom, 1: its terms sT -> top) on wave ch & 7, lane ch >> 3
            const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
            if (l < 2 && w + 8 * l < 12) {
                const double *src = pwLds + (int64_t) (w + 8 * l) * nPosPad;
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
                cmd.sum[w + 8 * l] = acc;
            }
        }
        __syncthreads();
        {
            const int w = threadIdx.x >> 6;
            if (w < 6) {
                const int x = w < 3 ? 0 : w < 5 ? 1 : 2, y = w == 0 ? 1 : (w == 1 || w == 3) ? 2 : 3;
                if (leaf[x] && leaf[y] && counts) {
                    REAL d, wt;
                    vft_pair_wave<REAL, NC>(A, id[x], id[y], false, nullptr, nullptr, d, wt);   // (the integer path touches no LDS)
                    if ((threadIdx.x & 63) == 0) cmd.dist[w] = (double) d;
                } else if ((threadIdx.x & 63) == 0) {
                    const double denom = cmd.sum[2 * w], top = cmd.sum[2 * w + 1];
                    const REAL d = (REAL) (denom > 0 ? top / denom : 1.0);
                    cmd.dist[w] = (double) d;
                }
            }
        }
    }
}

// wave 0's state: plain scalars handed to small helpers by reference (everything is inlined; nothing is indexed at run time)
struct SprWalk {
    uint32_t epoch;
    int nQueued;
    long long nAvg, tAvg, tDist;
};

template <typename REAL, int NC>
__global__ __launch_bounds__(VFT_SPR_WG) void k_spr_walk(Arena<REAL> A_, SprState S_) {
    extern __shared__ __attribute__((aligned(16))) double pwLds[];
    __shared__ SprCmd cmd;
    // The arena and the walk's state as LDS objects: vft_spr_exec is ONE function called from many places (not inlined), and a
    // reference to a kernel argument handed to a real function makes the compiler keep a copy of the argument in scratch memory
    // - every A.mlW, A.d.nPos in the column loops was then a trip to private memory in front of the load it feeds (12 us of a
    // 28 us distance command).  A reference to an LDS object costs an LDS read.
    __shared__ Arena<REAL> sArena;
    __shared__ SprState sState;
    if (threadIdx.x == 0) {
        sArena = A_;
        sState = S_;
    }
    __syncthreads();
    const Arena<REAL> &A = sArena;
    const SprState &S = sState;
    const int lane = threadIdx.x & 63;
    if ((threadIdx.x >> 6) != 0) {   // workers: wait for a command, do its data half, report back
        for (;;) {
            __syncthreads();
            if (cmd.type == 0) return;
            vft_spr_exec<REAL, NC>(A, S, cmd, pwLds);
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
    W.nAvg = W.tAvg = W.tDist = 0;
    int64_t nSPR = 0, nSteps = 0;
    const long long tStart = wall_clock64();
    auto sync0 = [&]() {   // wave 0's own stores (lane 0) before its next loads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto ld = [&](int32_t v) -> SprNode {
        const int4 r = *reinterpret_cast<const int4 *>(&nodes[v]);
        SprNode n;
        n.parent = r.x;
        n.c0 = r.y;
        n.c1 = r.z;
        n.c2 = r.w;
        return n;
    };
    auto issue = [&]() {   // hand the command in LDS to the workgroup and take part in it
        const long long t0 = wall_clock64();
        const int type = cmd.type;
        __syncthreads();
        vft_spr_exec<REAL, NC>(A, S, cmd, pwLds);
        __syncthreads();
        const long long dt = wall_clock64() - t0;
        if (type == 1) W.tAvg += dt;
        else W.tDist += dt;
    };
    auto flushAverages = [&]() {
        if (W.nQueued == 0) return;
        if (lane == 0) {
            cmd.type = 1;
            cmd.n = W.nQueued;
        }
        W.nAvg += W.nQueued;
        W.nQueued = 0;
        issue();
    };
    auto queueAverage = [&](int32_t out, int32_t a, int32_t b) {
        if (lane == 0) {
            cmd.out[W.nQueued] = out;
            cmd.a[W.nQueued] = a;
            cmd.b[W.nQueued] = b;
        }
        if (++W.nQueued == VFT_SPR_CHAIN) flushAverages();
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
        if (lane == 0) upEpoch[v] = e;
    };
    auto replaceChild = [&](int32_t par, int32_t oldChild, int32_t newChild) {   // NJ.tcc:1929-1940
        SprNode P = ld(par);
        if (P.c0 == oldChild) P.c0 = newChild;
        else if (P.c1 == oldChild) P.c1 = newChild;
        else if (P.c2 == oldChild) P.c2 = newChild;
        if (lane == 0) {
            *reinterpret_cast<int4 *>(&nodes[par]) = make_int4(P.parent, P.c0, P.c1, P.c2);
            nodes[newChild].parent = par;
        }
        sync0();
    };
    auto ensureUpProfile = [&](int32_t node) {   // getUpProfile (NJ.tcc:3382-3434): cached; missing ones from the root down
        if (upEpoch[node] == W.epoch) return;
        int len = 0;
        for (int32_t x = node; x != root;) {
            const SprNode X = ld(x);
            if (lane == 0) pathBuf[len] = x;
            len++;
            x = X.parent;
        }
        sync0();
        for (int t = len; t-- > 0;) {
            const int32_t x = pathBuf[t];
            if (upEpoch[x] == W.epoch) continue;
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
        flushAverages();
        if (lane == 0) {
            cmd.type = 2;
            cmd.q[0] = q0;
            cmd.q[1] = q1;
            cmd.q[2] = q2;
            cmd.q[3] = idD;
        }
        issue();
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
    flushAverages();
    if (lane == 0) {
        outBuf[0] = nSPR;
        outBuf[1] = nSteps;
        outBuf[2] = W.nAvg;
        outBuf[3] = (int64_t) W.epoch;
        outBuf[4] = W.tAvg;
        outBuf[5] = W.tDist;
        outBuf[6] = wall_clock64() - tStart;
        cmd.type = 0;
    }
    __syncthreads();   // releases the workers
}

#endif
