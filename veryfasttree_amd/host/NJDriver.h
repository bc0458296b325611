// Host driver of the neighbour-joining phase over the C ABI (include/vft_hip.h): the CALLER of the hot path.
//
// fastNJ's top-hits bookkeeping (src/NeighbourJoining.tcc "NJ.tcc" :2796-3155, :3746-4833) restated so that every
// profile operation goes through a batched device call, while the control flow and the scalar formulas the
// reference evaluates on the host (criterion :1099-1107, branch lengths :2911-2916, diameters :3003) stay here, with
// the reference's float/double mix.  Join order is pinned against the reference's `Join` lines
// (tests/golden/bb_*.npz; tests/test_gpu_nj_driver.py).  The same logic exists as a Python restatement among the
// tests (tests/nj_driver_py.py) that also runs on the CPU oracle for debugging.
//
// Scope: deterministic single-thread semantics, default options, `-fastest` (with its second-level top-hit lists) and
// `-fastest -no2nd`; no constraints, no BIONJ weighting, top-hits on (m >= 4 and 2m < nSeqs).
//
// Device-side lazy state: out-distances are refreshed on the device inside sweeps / pair lists exactly when the
// reference refreshes them (setCriterion, NJ.tcc:1092-1098).  Every refresh is also stored by the kernels into
// host-mapped mirrors (vft_out_distance_mirror), which this driver reads directly; it only has to make sure the
// stream has drained (`pending`) after calls that do not return data.
#ifndef VERYFASTTREE_NJDRIVER_H
#define VERYFASTTREE_NJDRIVER_H

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include <memory>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../../include/vft_host.h"
#include "MLLengths.h"
#include "KnuthRng.h"
#include "AAModels.h"

namespace veryfasttree {

    struct NJOptions {
        bool fastest = false;
        double tophitsMult = 1.0, tophitsClose = -1.0, tophitsRefresh = 0.8, topvisibleMult = 1.5;
        double staleOutLimit = 0.01, fResetOutProfile = 0.02;
        int nResetOutProfile = 200;
        /* second-level top hits: -fastest at one thread (main.cpp:339-343, VeryFastTree.cpp:87-91, Options.h:31-35) */
        bool useTopHits2nd = false;
        double tophits2Mult = 1.0, tophits2Refresh = 0.6;
        int tophits2Safety = 3;
        /* OpenMP threads for the host-only parts of a top-hits refresh (the reference parallelises the same loop,
           NJ.tcc:4476); results do not depend on it */
        int hostThreads = 0;   /* 0: $VFT_HOST_THREADS, else as many as OpenMP offers, at most 16 */
        /* logCorrect of the minimum-evolution lengths: scoredist-like instead of Jukes-Cantor (amino acids / matrix) */
        bool meNNI = false;
        bool scoredist = false;
        /* amino acids: 0 = matrices are the caller's business, else AAModel (JTT / WAG / LG): BLOSUM45-derived distances in
           the NJ / ME phase, the model's transition matrix in the ML phase (VeryFastTreeImpl.tcc:96-108, 253-256) */
        int aaModel = 0;
        /* multi-GPU (include/vft_host.h, vft_comm): sweeps and leaf blocks are split over the ranks */
        const vft_comm *comm = nullptr;
        /* with comm: split the close-neighbour blocks of setAllLeafTopHits by rows and all-gather the results (round 3); off: every
           rank computes them whole - 25 GB of gathers at a million sequences cost more than the ~5 s of integer counts they split */
        bool shardLeafBlocks = false;
        int outProfileParts = 0;   /* vft_nj_options.out_profile_parts: full out-profile recomputations as P blocks of the active list, split over the ranks */
        int seedBatch = 8;         /* setAllLeafTopHits: sweeps of this many unvisited seeds per device call (vft_sweep_batch); 1: a call per seed */
        bool walkServer = true;    /* refinement walks through the resident walk server (vft_walk_server_start); false: a launch per step */
        bool walkDual = true;      /* SPR chains hand both continuations of a step to the walk server (MLLengths::specContinuations); false: every step waits for the host's verdict */
        /* > 1: the refinement stages follow the reference's `-threads T` schedule (MLLengths.h "the subtree schedule") */
        int threads = 1;
        bool gamma = false;          /* `-gamma`: rescale the final lengths to a fitted discrete Gamma (MLLengths::branchlengthScale) */
        /* top-hit lists on the device (vft_tophits_*): the list walks of a join are one launch each; false = the host walks
           of round 2 (kept as the cross-check of VFT_NJ_CHECK and for tools) */
        bool deviceLists = true;
        /* the join loop itself on the device (vft_nj_engine_*): the host enqueues the kernels of many joins ahead, reads the
           join records from a log and handles top-visible resets and top-hits refreshes; needs deviceLists, first-level lists */
        bool deviceJoins = true;
        /* resetTopVisible through the engine's kernels; false = entirely on the host (tools) */
        bool deviceReset = true;
        /* joins the engine is asked to run ahead of the host */
        int engineWindow = 16;
        /* measurements only: ML length rounds as one batch per tree height (MLLengths::optimizeRoundParallel) - not the
           reference's order in any of its modes */
        bool parallelLengths = false;
    };

    template<typename REAL>
    class NJDriver {
    public:
        struct Join {
            int64_t i, j, newnode;
            REAL criterion;
        };

        NJDriver(vft_ctx *ctx, const uint8_t *codes, int64_t nSeqs, int64_t nPos, const NJOptions &opt)
                : ctx(ctx), opt(opt), nSeqs(nSeqs), nPos(nPos), maxnodes(2 * nSeqs), maxnode(nSeqs), totdiam(0.0) {
            if (this->opt.hostThreads <= 0) {
#ifdef _OPENMP
                const char *env = std::getenv("VFT_HOST_THREADS");
                this->opt.hostThreads = env ? std::max(1, atoi(env)) : std::max(1, std::min(omp_get_max_threads(), 16));
#else
                this->opt.hostThreads = 1;
#endif
            }
            parent.assign(maxnodes, -1);
            child0.assign(maxnodes, -1);
            child1.assign(maxnodes, -1);
            diameter.assign(maxnodes, 0);
            branchlength.assign(maxnodes, 0);
            selfweightLeaf.resize(nSeqs);
            for (int k = 0; k < 4; k++) leafCodeCounts[k] = 0;
            gapsPerPos.assign((size_t) nPos, 0);
            for (int64_t i = 0; i < nSeqs; i++) {
                int64_t c = 0;
                for (int64_t p = 0; p < nPos; p++) {
                    const uint8_t code = codes[i * nPos + p];
                    c += code != VFT_NOCODE;
                    gapsPerPos[(size_t) p] += code == VFT_NOCODE;
                    if (code < 4) leafCodeCounts[code]++;
                }
                selfweightLeaf[i] = (REAL) c;
            }
            if (opt.aaModel) installBlosum45();
            if (opt.comm && opt.comm->world > 1) {
                if (!opt.comm->allgather || !opt.comm->d_send || !opt.comm->d_recv || !opt.comm->h_send || !opt.comm->h_recv)
                    throw std::invalid_argument("NJDriver: incomplete vft_comm");
                chkT("vft_set_shard_mode", [&]() { return vft_set_shard_mode(ctx, 1); });
            }
            /* NJ constructor, NJ.tcc:233-260 */
            chkT("vft_upload_leaves", [&]() { return vft_upload_leaves(ctx, codes); });
            std::vector<REAL> z(nSeqs, 0);
            chkT("vft_set_node_scalars", [&]() { return vft_set_node_scalars(ctx, 0, nSeqs, z.data(), selfweightLeaf.data(), z.data()); });
            chkT("vft_set_max_node", [&]() { return vft_set_max_node(ctx, nSeqs); });
            std::vector<int64_t> ids(nSeqs);
            for (int64_t i = 0; i < nSeqs; i++) ids[i] = i;
            chkT("vft_out_profile_full", [&]() { return vft_out_profile_full(ctx, nSeqs, ids.data()); });
            std::vector<int64_t> stale(nSeqs, 10 * nSeqs);
            chkT("vft_set_out_distances", [&]() { return vft_set_out_distances(ctx, 0, nSeqs, z.data(), stale.data()); });
            chkT("vft_out_distances", [&]() { return vft_out_distances(ctx, 0, nullptr, nSeqs, 0.0); });
            chkT("vft_synchronize", [&]() { return vft_synchronize(ctx); });
            const void *od;
            chkT("vft_out_distance_mirror", [&]() { return vft_out_distance_mirror(ctx, &od, &mN); });
            mOut = (const REAL *) od;
            pending = false;
        }

        const std::vector<Join> &run(int64_t maxJoins = -1) {
            if (nSeqs < 3) throw std::invalid_argument("NJDriver: fewer than 3 sequences");
            int64_t m = opt.tophitsMult > 0 ? (int64_t) (0.5 + opt.tophitsMult * std::sqrt((double) nSeqs)) : 0;
            /* NJ.tcc:2827-2834: no top hits on tiny inputs; the visible set then holds every node's best hit */
            const bool noTop = m < 4 || 2 * m >= nSeqs;
            if (noTop) {
                visibleAll.assign((size_t) maxnodes, Besthit());
                for (int64_t v = 0; v < nSeqs; v++) visibleAll[(size_t) v] = bestHitOf(v, nSeqs, nullptr);   /* NJ.tcc:2849-2851 */
            } else {
                initTopHits(m);
                setAllLeafTopHits();
                resetTopVisible(nSeqs);
                if (devLists && !hostLists && opt.deviceJoins && runEngine(maxJoins)) return joins;
            }
            int64_t nActiveReset = nSeqs;
            for (int64_t nActive = nSeqs; nActive > 3; nActive--) {
                if (maxJoins >= 0 && (int64_t) joins.size() >= maxJoins) break;
                if (!noTop && dumpJoin >= 0 && (int64_t) joins.size() == dumpJoin) {
                    drain();
                    dumpVisibleState("host", dumpJoin, nActive);
                }
                Besthit join = noTop ? fastNJSearch(nActive) : topHitNJSearch(nActive);
                /* setOutDistance(i), setOutDistance(j), setDistCriterion(join) (NJ.tcc:2897-2901) as ONE pair list of
                   length 1 with nDiffAllow = 0: the lazy refresh then fires for every stamp != nActive, i.e. it is the
                   unconditional setOutDistance, and the pair kernel follows in the same call */
                if (!noTop && stamp(join.i) == nActive && stamp(join.j) == nActive && !checkJoins) {
                    /* both out-distances are current (the hill climbing of topHitNJSearch forced them) and join.dist is
                       the distance of exactly this pair of unchanged profiles: only the criterion's arithmetic is left */
                    criterionFresh(nActive, join);
                } else {
                    std::vector<Besthit *> one(1, &join);
                    Besthit before = join;
                    setDistCriterionBatch(nActive, one, 0);
                    if (checkJoins && !noTop && before.dist != join.dist)
                        throw std::runtime_error("NJDriver: the stored distance of a join differs from its recomputation");
                    if (checkJoins && !noTop) {
                        criterionFresh(nActive, before);
                        if (before.criterion != join.criterion)
                            throw std::runtime_error("NJDriver: host and device criterion of a join differ");
                    }
                }
                const int64_t newnode = maxnode++;
                const int64_t i = join.i, j = join.j;
                parent[i] = parent[j] = newnode;
                child0[newnode] = std::min(i, j);
                child1[newnode] = std::max(i, j);
                joins.push_back(Join{std::min(i, j), std::max(i, j), newnode, join.criterion});
                const double distIJ = join.dist;
                const REAL od = value(i) - value(j);
                const double deltaDist = od / (double) (nActive - 2);
                branchlength[i] = (REAL) ((distIJ + deltaDist) / 2);
                branchlength[j] = (REAL) ((distIJ - deltaDist) / 2);
                const double bw = 0.5;
                const REAL bi = branchlength[i] + diameter[i], bj = branchlength[j] + diameter[j];
                diameter[newnode] = (REAL) (bw * bi + (1 - bw) * bj);
                /* one launch per join: tree arrays, the average, its self-distance and the incremental out-profile
                   (vft_join_fused; the tile streams are rebuilt lazily, 64 joins at a time, before the next sweep) */
                const int64_t changed = nActiveReset - (nActive - 1);
                const bool fullOut = changed >= opt.nResetOutProfile && changed >= opt.fResetOutProfile * nActiveReset;
                chkT("vft_join_fused", [&]() {
                    return vft_join_fused(ctx, i, j, newnode, (double) diameter[newnode], 10 * nSeqs /* NJ.tcc:254: "unreasonably high" */,
                                          nActive, fullOut ? 0 : 1);
                });
                if (fullOut) {
                    std::vector<int64_t> active;
                    double tot = 0;
                    for (int64_t v = 0; v < maxnode; v++)
                        if (parent[v] < 0) {
                            active.push_back(v);
                            tot += diameter[v];
                        }
                    totdiam = tot;
                    outProfileFull(active);
                    nActiveReset = nActive - 1;
                } else {
                    const REAL dd = diameter[newnode] - diameter[i] - diameter[j];
                    totdiam += dd;
                }
                if (noTop) visibleJoin(newnode, nActive - 1);
                else topHitJoin(newnode, nActive - 1);
            }
            return joins;
        }

        /* ---- the join loop on the device (include/vft_hip.h, vft_nj_engine_*) */
        int64_t engineConsumed = 0;
        int64_t dumpJoin = std::getenv("VFT_NJ_DUMP_JOIN") ? atoll(std::getenv("VFT_NJ_DUMP_JOIN")) : -1;
        bool traceEvents = std::getenv("VFT_NJ_TRACE_EVENTS") != nullptr;   /* debugging: resets / refreshes with their join index */

        /* tree arrays, branch lengths, diameters and the Join records of the joins [engineConsumed, upTo) from the device's log */
        void consumeLog(const vft_nj_join_t *log, int64_t upTo) {
            if (upTo <= engineConsumed) return;
            chkT("vft_nj_engine_adopt", [&]() { return vft_nj_engine_adopt(ctx, engineConsumed, upTo); });
            for (int64_t k = engineConsumed; k < upTo; k++) {
                const vft_nj_join_t &r = log[k];
                const int64_t i = r.i, j = r.j, newnode = r.newnode;
                if (newnode != maxnode) throw std::runtime_error("NJDriver: join log out of order");
                maxnode++;
                parent[i] = parent[j] = newnode;
                child0[newnode] = std::min(i, j);
                child1[newnode] = std::max(i, j);
                joins.push_back(Join{std::min(i, j), std::max(i, j), newnode, (REAL) r.criterion});
                branchlength[i] = (REAL) r.bl_i;
                branchlength[j] = (REAL) r.bl_j;
                diameter[newnode] = (REAL) r.diameter;
            }
            engineConsumed = upTo;
        }

        void engineDownloadVisible() {
            std::vector<int32_t> vj((size_t) maxnode);
            std::vector<REAL> vd((size_t) maxnode);
            chkT("vft_nj_engine_visible_get", [&]() { return vft_nj_engine_visible_get(ctx, 0, maxnode, vj.data(), vd.data()); });
            for (int64_t v = 0; v < maxnode; v++) visible[(size_t) v] = Hit{vj[(size_t) v], vd[(size_t) v]};
        }

        void engineUploadNodes(const std::vector<int64_t> &nodes, int32_t newAge) {
            if (nodes.empty()) return;
            std::vector<int32_t> vj(nodes.size());
            std::vector<REAL> vd(nodes.size());
            for (size_t t = 0; t < nodes.size(); t++) {
                vj[t] = visible[(size_t) nodes[t]].j;
                vd[t] = visible[(size_t) nodes[t]].dist;
            }
            chkT("vft_nj_engine_nodes_set", [&]() { return vft_nj_engine_nodes_set(ctx, (int64_t) nodes.size(), nodes.data(), vj.data(), vd.data(), newAge); });
        }

        void engineUploadTopVisible() {
            std::vector<int32_t> tv(topvisible.size());
            for (size_t t = 0; t < tv.size(); t++) tv[t] = (int32_t) topvisible[t];
            chkT("vft_nj_engine_topvisible_set", [&]() { return vft_nj_engine_topvisible_set(ctx, tv.data()); });
            chkT("vft_nj_engine_set_state", [&]() { return vft_nj_engine_set_state(ctx, -1, -1, std::nan(""), (int32_t) topvisibleAge); });
        }

        /* debugging (VFT_NJ_DUMP_JOIN=k): the visible set as join k's search finds it */
        void dumpVisibleState(const char *who, int64_t k, int64_t nActive) const {
            fprintf(stderr, "[dump %s] join %lld nActive %lld topvisibleAge %lld\n", who, (long long) k, (long long) nActive, (long long) topvisibleAge);
            for (size_t t = 0; t < topvisible.size(); t++) {
                const int64_t node = topvisible[t];
                if (node < 0) {
                    fprintf(stderr, "[dump %s] slot %zu: -1\n", who, t);
                    continue;
                }
                const int64_t j = visible[(size_t) node].j;
                fprintf(stderr, "[dump %s] slot %zu: node %lld parent %lld vis (%lld, %.9g) out %.9g stamp %d | partner parent %lld out %.9g stamp %d\n", who, t,
                        (long long) node, (long long) parent[(size_t) node], (long long) j, (double) visible[(size_t) node].dist, (double) mOut[node], (int) mN[node],
                        (long long) (j >= 0 ? parent[(size_t) j] : -9), (double) (j >= 0 ? mOut[j] : 0), (int) (j >= 0 ? mN[j] : 0));
            }
        }

        /* false: the engine cannot be used on this context (the caller runs the host-driven loop) */
        bool runEngine(int64_t maxJoins = -1) {
            vft_nj_engine_config cfg;
            memset(&cfg, 0, sizeof(cfg));
            cfg.m = (int32_t) m;
            cfg.n_top = (int32_t) topvisible.size();
            cfg.need = (int32_t) (int64_t) (0.5 + m * opt.tophitsRefresh);
            cfg.age_limit = (int32_t) std::max<int64_t>(1, (int64_t) (0.5 + std::log((double) m) / std::log(2.0)));
            cfg.fastest = opt.fastest ? 1 : 0;
            cfg.stale_stamp = 10 * nSeqs;
            cfg.stale_out_limit = opt.tophitsMult > 0 ? opt.staleOutLimit : 0.0;
            if (vft_nj_engine_create(ctx, &cfg) != VFT_OK) return false;
            engineActive = true;
            Section sec(this, "[host] join engine (incl. device)");
            drain();
            const vft_nj_join_t *log = nullptr;
            chk(vft_nj_engine_log(ctx, &log));
            {
                std::vector<int32_t> vj((size_t) nSeqs);
                std::vector<REAL> vd((size_t) nSeqs);
                for (int64_t v = 0; v < nSeqs; v++) {
                    vj[(size_t) v] = visible[(size_t) v].j;
                    vd[(size_t) v] = visible[(size_t) v].dist;
                }
                chkT("vft_nj_engine_visible_set", [&]() { return vft_nj_engine_visible_set(ctx, 0, nSeqs, vj.data(), vd.data()); });
            }
            chkT("vft_nj_engine_set_state", [&]() { return vft_nj_engine_set_state(ctx, nSeqs, nSeqs, totdiam, 0); });
            engineUploadTopVisible();
            const int64_t nTotal = maxJoins >= 0 ? std::min(maxJoins, nSeqs - 3) : nSeqs - 3;   /* (a truncated run: tests) */
            const int64_t window = std::max(1, opt.engineWindow);
            int64_t enq = 0, nActiveReset = nSeqs;
            bool climbPending = false, needSearch = true;   /* needSearch: the search of join `enq` has not been enqueued behind the previous merge */
            auto isFullOut = [&](int64_t k) {
                const int64_t nActive = nSeqs - k, changed = nActiveReset - (nActive - 1);
                return changed >= opt.nResetOutProfile && changed >= opt.fResetOutProfile * nActiveReset;
            };
            /* the events the kernels raise; afterwards the loop goes on enqueueing from `enq` */
            auto handleHalt = [&](int32_t reason, int64_t h) {
                int64_t devActive = 0, devMax = 0, done = 0;
                int32_t tvAge = 0;
                double devTot = 0;
                chkT("vft_nj_engine_get_state", [&]() { return vft_nj_engine_get_state(ctx, &devActive, &devMax, &devTot, &tvAge, &done, nullptr, nullptr, nullptr); });
                totdiam = devTot;
                if (traceEvents) {
                    int32_t nU = 0;
                    chk(vft_nj_engine_get_state(ctx, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &nU));
                    fprintf(stderr, "[event engine] %s at join %lld (nActive %lld) tvAge %d nUnique %d\n", reason == VFT_NJ_HALT_RESET ? "reset" : reason == VFT_NJ_HALT_REFRESH ? "refresh" : "climb",
                            (long long) h, (long long) devActive, (int) tvAge, (int) nU);
                }
                if (reason == VFT_NJ_HALT_CLIMB) {
                    consumeLog(log, h);
                    chkT("vft_nj_engine_resume", [&]() { return vft_nj_engine_resume(ctx, h); });
                    climbPending = true;
                    needSearch = false;
                    enq = h;
                    if (profiling) acc["[count]  engine: extra hill-climbing rounds"].calls++;
                    return;
                }
                if (reason == VFT_NJ_HALT_RESET) {
                    Section s2(this, "[host]   engine: top-visible reset (incl. device)");
                    consumeLog(log, h);
                    chkT("vft_nj_engine_resume", [&]() { return vft_nj_engine_resume(ctx, h); });
                    const int64_t nActive = nSeqs - h;
                    if (devActive != nActive || devMax != maxnode) throw std::runtime_error("NJDriver: engine state out of step at a reset");
                    topvisibleAge = tvAge;
                    if (topvisibleAge <= 2) {   /* NJ.tcc:4166-4195: visible hits whose partner was joined are re-targeted first */
                        engineDownloadVisible();
                        std::vector<int64_t> changedNodes;
                        for (int64_t node = 0; node < maxnode; node++) {
                            if (parent[node] >= 0) continue;
                            Hit &v = visible[node];
                            int64_t newj = activeAncestor(v.j);
                            if (newj >= 0 && newj != v.j) {
                                if (newj == node) {
                                    newj = 0;
                                    while (parent[newj] >= 0 || newj == node) newj++;
                                }
                                Besthit bh;
                                bh.i = node;
                                bh.j = newj;
                                std::vector<Besthit *> one(1, &bh);
                                setDistCriterionBatch(nActive, one);
                                v.j = (int32_t) newj;
                                v.dist = bh.dist;
                                changedNodes.push_back(node);
                            }
                        }
                        engineUploadNodes(changedNodes, -1);
                    }
                    resetTopVisible(nActive);
                    engineUploadTopVisible();
                    enq = h;
                    needSearch = true;
                    climbPending = false;
                    return;
                }
                if (reason == VFT_NJ_HALT_REFRESH) {
                    Section s2(this, "[host]   engine: top-hits refresh (incl. device)");
                    consumeLog(log, h + 1);
                    chkT("vft_nj_engine_resume", [&]() { return vft_nj_engine_resume(ctx, h + 1); });
                    const int64_t nActive = nSeqs - h - 1, newnode = log[h].newnode;
                    if (devActive != nActive || devMax != maxnode) throw std::runtime_error("NJDriver: engine state out of step at a refresh");
                    refreshTopHits(newnode, nActive);   /* (device merge; uploads the new first hits; ends with resetTopVisible) */
                    engineUploadTopVisible();
                    enq = h + 1;
                    needSearch = true;
                    climbPending = false;
                    return;
                }
                throw std::runtime_error("NJDriver: the join engine stopped with an error");
            };
            int64_t lastDone = -1;
            auto lastProgress = std::chrono::steady_clock::now();
            while (engineConsumed < nTotal) {
                int64_t done = 0;
                int32_t halt = 0, haltJoin = 0;
                chk(vft_nj_engine_poll(ctx, &done, &halt, &haltJoin));
                if (halt) {
                    handleHalt(halt, haltJoin);
                    lastProgress = std::chrono::steady_clock::now();
                    continue;
                }
                consumeLog(log, std::min(done, nTotal));
                if (dumpJoin >= 0 && done == dumpJoin && enq == done) {   /* (window 1: nothing of join dumpJoin has been enqueued) */
                    chkT("vft_synchronize", [&]() { return vft_synchronize(ctx); });
                    engineDownloadVisible();
                    std::vector<int32_t> tv(topvisible.size());
                    chk(vft_nj_engine_topvisible_get(ctx, tv.data()));
                    for (size_t t = 0; t < tv.size(); t++) topvisible[t] = tv[t];
                    int32_t tvAge = 0;
                    chk(vft_nj_engine_get_state(ctx, nullptr, nullptr, nullptr, &tvAge, nullptr, nullptr, nullptr, nullptr));
                    topvisibleAge = tvAge;
                    dumpVisibleState("engine", dumpJoin, nSeqs - dumpJoin);
                    dumpJoin = -1;
                }
                if (done != lastDone) {
                    lastDone = done;
                    lastProgress = std::chrono::steady_clock::now();
                }
                if (enq >= nTotal || enq >= done + window) {
                    for (int spin = 0; spin < 64; spin++) __builtin_ia32_pause();
                    /* a stream that makes no progress for 30 s is broken: surface its error instead of spinning on */
                    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - lastProgress).count() > 30.0) {
                        chkT("vft_synchronize", [&]() { return vft_synchronize(ctx); });
                        chk(vft_nj_engine_poll(ctx, &done, &halt, &haltJoin));
                        if (!halt && done == lastDone)
                            throw std::runtime_error("NJDriver: the join engine makes no progress (join " + std::to_string((long long) done) + ")");
                        lastProgress = std::chrono::steady_clock::now();
                    }
                    continue;
                }
                const int32_t first = (climbPending ? VFT_NJ_PHASE_CLIMB : 0) | (needSearch ? VFT_NJ_PHASE_SEARCH : 0);
                const int32_t next = enq + 1 < nTotal ? VFT_NJ_PHASE_NEXT : 0;   /* the merge kernels go on with the next join's search */
                int rc;
                if (!isFullOut(enq)) {
                    rc = vft_nj_engine_enqueue(ctx, enq, first | VFT_NJ_PHASE_JOIN | VFT_NJ_PHASE_MERGE | next, 1);
                    if (rc == VFT_ERR_HALTED) continue;   /* halted meanwhile: the next poll sees it (any other error is thrown) */
                    chk(rc);
                    climbPending = false;
                    needSearch = next == 0;
                    enq++;
                    continue;
                }
                /* a join after which the out-profile is recomputed from scratch (NJ.tcc:3012-3033): the merge needs the new
                   out-profile and totdiam, so the host waits for the join in between */
                rc = vft_nj_engine_enqueue(ctx, enq, first | VFT_NJ_PHASE_JOIN, 0);
                if (rc == VFT_ERR_HALTED) continue;
                chk(rc);
                climbPending = false;
                needSearch = false;   /* (of THIS join: a halt below sets it again) */
                int64_t devActive = 0;
                chkT("vft_nj_engine_get_state", [&]() { return vft_nj_engine_get_state(ctx, &devActive, nullptr, nullptr, nullptr, nullptr, &halt, &haltJoin, nullptr); });
                if (halt) {
                    handleHalt(halt, haltJoin);
                    continue;
                }
                if (devActive != nSeqs - enq - 1) throw std::runtime_error("NJDriver: engine state out of step at an out-profile reset");
                consumeLog(log, enq + 1);
                {
                    std::vector<int64_t> active;
                    double tot = 0;
                    for (int64_t v = 0; v < maxnode; v++)
                        if (parent[v] < 0) {
                            active.push_back(v);
                            tot += diameter[v];
                        }
                    totdiam = tot;
                    outProfileFull(active);
                    chkT("vft_nj_engine_set_state", [&]() { return vft_nj_engine_set_state(ctx, -1, -1, totdiam, -1); });
                    nActiveReset = nSeqs - enq - 1;
                }
                chk(vft_nj_engine_enqueue(ctx, enq, VFT_NJ_PHASE_MERGE | next, 0));
                needSearch = next == 0;
                enq++;
            }
            /* the loop state of the host-driven code, for what follows (finishRoot, branch lengths) */
            chkT("vft_nj_engine_get_state", [&]() { return vft_nj_engine_get_state(ctx, nullptr, nullptr, &totdiam, nullptr, nullptr, nullptr, nullptr, nullptr); });
            pending = false;
            engineActive = false;
            return true;
        }

        std::vector<Join> joins;

        /* The end of fastNJ (NJ.tcc:3098-3120): a root for the 3 remaining nodes; their branch lengths from the three
           raw profile distances.  Call after run() has gone all the way (nActive == 3). */
        void finishRoot() {
            int64_t top[3], nTop = 0;
            for (int64_t v = 0; v < maxnode; v++)
                if (parent[v] < 0) {
                    if (nTop == 3) throw std::invalid_argument("NJDriver::finishRoot: more than 3 active nodes");
                    top[nTop++] = v;
                }
            if (nTop != 3) throw std::invalid_argument("NJDriver::finishRoot: fewer than 3 active nodes");
            root = maxnode++;
            for (int k = 0; k < 3; k++) parent[top[k]] = root;
            rootChild[0] = top[0];
            rootChild[1] = top[1];
            rootChild[2] = top[2];
            const int64_t pi[3] = {top[0], top[0], top[1]}, pj[3] = {top[1], top[2], top[2]};
            REAL d[3], w[3];
            chkT("vft_profile_distances", [&]() { return vft_profile_distances(ctx, 3, pi, pj, d, w); });
            const double d01 = d[0] - diameter[top[0]] - diameter[top[1]];
            const double d02 = d[1] - diameter[top[0]] - diameter[top[2]];
            const double d12 = d[2] - diameter[top[1]] - diameter[top[2]];
            branchlength[top[0]] = (REAL) ((d01 + d02 - d12) / 2);
            branchlength[top[1]] = (REAL) ((d01 + d12 - d02) / 2);
            branchlength[top[2]] = (REAL) ((d02 + d12 - d01) / 2);
            /* the NJ phase is over: from here on single profiles are rewritten (up-profiles, NNIs, SPRs) and nothing
               sweeps them - plain rows instead of tile re-packs */
            chkT("vft_set_profile_rows", [&]() { return vft_set_profile_rows(ctx, 1); });
        }

        /* up-profiles of all internal nodes on the device (node X -> id X + nSeqs), breadth first from the root:
           one vft_average_profiles per depth (getUpProfile, NJ.tcc:3382-3434, useML = false, -nj weighting) */
        void ensureUpProfiles() {
            if (upReady) return;
            if (root < 0) throw std::invalid_argument("NJDriver: up-profiles before finishRoot");
            auto isInternal = [&](int64_t v) { return v >= nSeqs && v != root; };
            chkT("vft_set_max_node", [&]() { return vft_set_max_node(ctx, maxnode + nSeqs); });
            std::vector<int64_t> level;
            for (int k = 0; k < 3; k++)
                if (isInternal(rootChild[k])) level.push_back(rootChild[k]);
            while (!level.empty()) {
                std::vector<int64_t> out, a, b, next;
                for (int64_t x: level) {
                    int64_t cd[2];
                    quartetCD(x, cd);
                    out.push_back(x + nSeqs);
                    a.push_back(cd[0]);
                    b.push_back(cd[1]);
                    if (isInternal(child0[x])) next.push_back(child0[x]);
                    if (isInternal(child1[x])) next.push_back(child1[x]);
                }
                chkT("vft_average_profiles", [&]() { return vft_average_profiles(ctx, (int64_t) out.size(), out.data(), a.data(), b.data(), nullptr); });
                level.swap(next);
            }
            upReady = true;
        }

        /* C and D of setupABCD (NJ.tcc:1942-1975) for node x: the two other children of the root, or the sibling and
           the up-profile of the parent */
        void quartetCD(int64_t x, int64_t cd[2]) const {
            const int64_t p = parent[x];
            if (p == root) {
                int n = 0;
                for (int k = 0; k < 3; k++)
                    if (rootChild[k] != x) cd[n++] = rootChild[k];
            } else {
                cd[0] = child0[p] == x ? child1[p] : child0[p];
                cd[1] = p + nSeqs;
            }
        }

        /* reliabilityNJ (NJ.tcc:3191-3238): local-bootstrap support of every internal split from nBootstrap column
           resamples drawn from Knuth's generator (resampleColumns, NJ.tcc:705-727); all nodes in one device call */
        void computeSupports(int32_t nBootstrap) {
            if (nSeqs <= 3 || nBootstrap <= 0) return;
            ensureUpProfiles();
            KnuthRng rng;
            std::vector<int32_t> col((size_t) nBootstrap * nPos);
            for (size_t t = 0; t < col.size(); t++) {
                int64_t pos = (int64_t) (rng.rand() * nPos);
                if (pos < 0) pos = 0;
                else if (pos == nPos) pos = nPos - 1;
                col[t] = (int32_t) pos;
            }
            std::vector<int64_t> nodes, a, b, c, d;
            for (int64_t v = nSeqs; v < maxnode; v++) {
                if (v == root) continue;
                int64_t cd[2];
                quartetCD(v, cd);
                nodes.push_back(v);
                a.push_back(child0[v]);
                b.push_back(child1[v]);
                c.push_back(cd[0]);
                d.push_back(cd[1]);
            }
            std::vector<double> sup(nodes.size());
            chkT("vft_split_supports", [&]() { return vft_split_supports(ctx, (int64_t) nodes.size(), a.data(), b.data(), c.data(), d.data(), nBootstrap, col.data(), sup.data()); });
            support.assign((size_t) maxnodes, -1.0);
            for (size_t t = 0; t < nodes.size(); t++) support[(size_t) nodes[t]] = sup[t];
        }

        /* updateBranchLengths (NJ.tcc:6514-6595) on the finished NJ topology, as the pipeline does before it prints a
           minimum-evolution tree (VeryFastTreeImpl.tcc:205-213): every branch length from log-corrected profile
           distances between the node's children / sibling / up-profile (correctedPairDistances NJ.tcc:1460-1488,
           logCorrect :322-330; -nj weighting, no pseudocounts: the defaults).  The up-profile of an internal node X
           (getUpProfile :3382-3434, useML = false) is average(sibling(X), up(parent X)), or the average of the two
           other children of the root; it lives on the device as node X + nSeqs, so the context must have been created
           with max_nodes >= 3 * nSeqs.  Up-profiles go level by level (one vft_average_profiles per depth), all
           distances are ONE vft_profile_distances list. */
        void updateBranchLengths() {
            if (root < 0) throw std::invalid_argument("NJDriver::updateBranchLengths before finishRoot");
            const int64_t upOff = nSeqs;
            auto up = [&](int64_t v) { return v + upOff; };
            auto siblingOf = [&](int64_t v) { const int64_t p = parent[v]; return child0[p] == v ? child1[p] : child0[p]; };
            auto rootSibs = [&](int64_t v, int64_t sibs[2]) {
                int n = 0;
                for (int k = 0; k < 3; k++)
                    if (rootChild[k] != v) sibs[n++] = rootChild[k];
            };
            ensureUpProfiles();
            /* 2. every distance of every branch, in correctedPairDistances' order (i < j over A, B, C[, D]) */
            std::vector<int64_t> pi, pj, firstPair((size_t) maxnode + 1, 0);
            for (int64_t v = 0; v < maxnode; v++) {
                firstPair[(size_t) v] = (int64_t) pi.size();
                if (v == root) continue;
                int64_t q[4], nq;
                if (v < nSeqs) {
                    q[0] = v;
                    if (parent[v] == root) {
                        int64_t sibs[2];
                        rootSibs(v, sibs);
                        q[1] = sibs[0];
                        q[2] = sibs[1];
                    } else {
                        q[1] = siblingOf(v);
                        q[2] = up(parent[v]);
                    }
                    nq = 3;
                } else {
                    q[0] = child0[v];
                    q[1] = child1[v];
                    if (parent[v] == root) {
                        int64_t sibs[2];
                        rootSibs(v, sibs);
                        q[2] = sibs[0];
                        q[3] = sibs[1];
                    } else {
                        q[2] = siblingOf(v);
                        q[3] = up(parent[v]);
                    }
                    nq = 4;
                }
                for (int64_t x = 0; x < nq; x++)
                    for (int64_t y = x + 1; y < nq; y++) {
                        pi.push_back(q[x]);
                        pj.push_back(q[y]);
                    }
            }
            firstPair[(size_t) maxnode] = (int64_t) pi.size();
            const int64_t nPairs = (int64_t) pi.size();
            std::vector<REAL> pd((size_t) nPairs), pw((size_t) nPairs);
            const int64_t maxCall = 1 << 22;
            for (int64_t p0 = 0; p0 < nPairs; p0 += maxCall) {
                const int64_t cnt = std::min<int64_t>(maxCall, nPairs - p0);
                chkT("vft_profile_distances", [&]() { return vft_profile_distances(ctx, cnt, pi.data() + p0, pj.data() + p0, pd.data() + p0, pw.data() + p0); });
            }
            /* 3. branch lengths (double arithmetic, stored as numeric_t) */
            auto logCorrect = [&](double dist) {   /* NJ.tcc:322-330 */
                const double maxscore = 3.0;
                if (!opt.scoredist) dist = dist < 0.74 ? -0.75 * std::log(1.0 - dist * 4.0 / 3.0) : maxscore;   /* Jukes-Cantor */
                else dist = dist < 0.99 ? -1.3 * std::log(1.0 - dist) : maxscore;                              /* scoredist-like */
                return dist < maxscore ? dist : maxscore;
            };
            for (int64_t v = 0; v < maxnode; v++) {
                if (v == root) continue;
                const int64_t f = firstPair[(size_t) v];
                if (v < nSeqs) {
                    const double dAB = logCorrect((double) pd[(size_t) f]), dAC = logCorrect((double) pd[(size_t) f + 1]),
                                 dBC = logCorrect((double) pd[(size_t) f + 2]);
                    branchlength[(size_t) v] = (REAL) ((dAB + dAC - dBC) / 2.0);
                } else {
                    double d[6];
                    for (int k = 0; k < 6; k++) d[k] = logCorrect((double) pd[(size_t) f + k]);
                    /* qAB 0, qAC 1, qAD 2, qBC 3, qBD 4, qCD 5 */
                    branchlength[(size_t) v] = (REAL) ((d[1] + d[2] + d[3] + d[4]) / 4.0 - (d[0] + d[5]) / 2.0);
                }
            }
        }

        /* parent[] / child[][3] of the finished tree in the layout MLLengths.h works on */
        void treeArrays(std::vector<int64_t> &par, std::vector<int64_t> &ch) const {
            par.assign((size_t) maxnode, -1);
            ch.assign((size_t) (3 * maxnode), -1);
            for (int64_t v = 0; v < maxnode; v++) {
                par[(size_t) v] = parent[(size_t) v];
                if (v == root) {
                    for (int k = 0; k < 3; k++) ch[(size_t) (3 * v + k)] = rootChild[k];
                } else if (v >= nSeqs) {
                    ch[(size_t) (3 * v)] = child0[(size_t) v];
                    ch[(size_t) (3 * v + 1)] = child1[(size_t) v];
                }
            }
            par[(size_t) root] = -1;
        }

        /* the topology back from MLLengths.h after rearrangements (child order as the NNIs left it: printNJ walks it) */
        void adoptTree(const std::vector<int64_t> &par, const std::vector<int64_t> &ch) {
            for (int64_t v = 0; v < maxnode; v++) {
                parent[(size_t) v] = v == root ? -1 : par[(size_t) v];
                if (v == root) {
                    for (int k = 0; k < 3; k++) rootChild[k] = ch[(size_t) (3 * v + k)];
                } else if (v >= nSeqs) {
                    child0[(size_t) v] = ch[(size_t) (3 * v)];
                    child1[(size_t) v] = ch[(size_t) (3 * v + 1)];
                }
            }
            upReady = false;
        }

        /* Minimum-evolution NNIs and SPRs (VeryFastTreeImpl.tcc:160-205): up to round(4 log2 N) rounds of
           DoNNI(useML = false), skipped once a round changes nothing, with `spr` rounds of subtree-prune-regraft moves
           in between (the reference's default is 2; 0 = `-spr 0`).  Call after
           finishRoot; the context needs max_nodes >= 3 * nSeqs (up-profiles).  Returns the number of NNIs made. */
        int64_t meNNIRounds(int32_t spr = 0) {
            if (root < 0) throw std::invalid_argument("NJDriver::meNNIRounds before finishRoot");
            if (nSeqs <= 3) return 0;
            std::vector<int64_t> par, ch;
            treeArrays(par, ch);
            MLLengths<REAL> tree(ctx, nSeqs, maxnode, par.data(), ch.data(), root);
            tree.walkServer = opt.walkServer;
            tree.walkDual = opt.walkDual;
            tree.comm = opt.comm;
            typename MLLengths<REAL>::NNIParams prm;
            prm.useML = false;
            prm.scoredist = opt.scoredist;
            std::vector<typename MLLengths<REAL>::NNIStats> stats;
            tree.initNNIStats(stats);
            const int64_t nniToDo = (int64_t) (0.5 + 4.0 * std::log((double) nSeqs) / std::log(2.0));
            int64_t total = 0, sprRemaining = spr;
            meNNIRoundsDone = 0;
            meSPRs = 0;
            meSPRSeconds = 0;
            bool bConverged = false;
            for (int64_t i = 0; i < nniToDo; i++) {
                if (!bConverged) {
                    double maxDelta;
                    const int64_t nChange = opt.threads > 1 ? tree.doNNIThreaded(prm, stats, maxDelta, opt.threads) : tree.doNNI(prm, stats, maxDelta);
                    meNNIRoundsDone++;
                    total += nChange;
                    if (nChange == 0) bConverged = true;
                }
                /* SPR rounds sit between thirds of the NNI rounds (VeryFastTreeImpl.tcc:187-198) */
                if (sprRemaining > 0 && nniToDo / (spr + 1) > 0 && ((i + 1) % (nniToDo / (spr + 1))) == 0) {
                    {
                        const std::chrono::steady_clock::time_point ts = std::chrono::steady_clock::now();
                        if (std::getenv("VFT_STAGE_TRACE")) fprintf(stderr, "[stage] SPR round begins after %lld NNI rounds\n", (long long) (i + 1));
                        meSPRs += tree.doSPR(opt.scoredist);
                        meSPRSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - ts).count();
                        if (std::getenv("VFT_STAGE_TRACE")) fprintf(stderr, "[stage] SPR round done: %.1f s of SPR so far, %lld moves\n", meSPRSeconds, (long long) meSPRs);
                    }
                    sprRemaining--;
                    bConverged = false;
                    tree.initNNIStats(stats);
                }
                if (bConverged && sprRemaining == 0) break;
            }
            while (sprRemaining > 0) {
                const std::chrono::steady_clock::time_point ts = std::chrono::steady_clock::now();
                meSPRs += tree.doSPR(opt.scoredist);
                meSPRSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - ts).count();
                sprRemaining--;
            }
            meSPRSteps = tree.sprSteps;
            meSPRDualSent = tree.walkDualSent;
            meSPRDualTaken = tree.walkDualTaken;
            laneGathers += tree.laneGathers;
            laneGatherBytes += tree.laneGatherBytes;
            adoptTree(tree.parents(), tree.children());
            return total;
        }

        int64_t meSPRs = 0, meSPRSteps = 0;
        int64_t meSPRDualSent = 0, meSPRDualTaken = 0;   /* SPR chains: dual commands sent / continuations the walk server ran without waiting for the host */
        int64_t laneGathers = 0, laneGatherBytes = 0;   /* lanes across ranks: all-gathers of verdicts + lengths, bytes received */
        double meSPRSeconds = 0, mlNNISeconds = 0, mlSupportSeconds = 0, mlModelSeconds = 0;   /* stage timers (vft_nj_last_stage_seconds) */
        int64_t meNNIRoundsDone = 0;

        /* `-mllen -nocat` under Jukes-Cantor (VeryFastTreeImpl.tcc:249-311): rounds of optimizeAllBranchLengths +
           treeLogLk on the finished topology until the largest change of a length drops below 0.001, at most
           round(log2 N) rounds; after the first round setMLRates (NJ.tcc:5429-5488): with one category (-nocat) just
           recomputeMLProfiles, with nRateCats > 1 the CAT approximation (MLLengths::setMLRates).  recomputeProfiles(tmatAsDist) at the start (VeryFastTreeImpl.tcc:253-256) is the identity
           here: without a transition matrix the unweighted averages are the profiles the joins already made.
           Call after updateBranchLengths; the context needs max_nodes >= 3 * nSeqs.  Returns the tree log-likelihood
           after each round (the reference's "TreeLogLk Length<k>" log lines). */
        std::vector<double> mlLengths(int32_t nRateCats = 1, int32_t nBootstrap = 0, bool mllen = true, bool mlNNI = false,
                                      bool reaverage = false, bool gtr = false) {
            if (root < 0) throw std::invalid_argument("NJDriver::mlLengths before finishRoot");
            const bool f32 = sizeof(REAL) == 4;
            const REAL one = 1;
            std::vector<int64_t> cat((size_t) nPos, 0);
            chkT("vft_set_rates", [&]() { return vft_set_rates(ctx, &one, 1, cat.data()); });
            chkT("vft_set_ml_limits", [&]() { return vft_set_ml_limits(ctx, f32 ? 5.0e-4 : 5.0e-9, f32 ? 2.5e-4 : 2.5e-9, f32 ? 1.0e-10 : 1.0e-20); });   /* Constants.h:26-39 */
            const double ftol = 0.001, atol = f32 ? 1.0e-4 : 1.0e-9;
            std::vector<int64_t> par, ch;
            treeArrays(par, ch);
            double nonGap = 0;
            for (int64_t i = 0; i < nSeqs; i++) nonGap += (double) selfweightLeaf[(size_t) i];
            int64_t nLeafGaps = nSeqs * nPos - (int64_t) nonGap;   /* -1 once a transition matrix is installed */
            if (opt.aaModel) {
                if (gtr) throw std::invalid_argument("NJDriver::mlLengths: -gtr is a nucleotide model");
                nLeafGaps = -1;   /* treeLogLk's gap correction is Jukes-Cantor only (NJ.tcc:5236) */
            } else {
                chkT("vft_set_transition_matrix", [&]() { return vft_set_transition_matrix(ctx, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr); });
            }
            mlNNISeconds = mlSupportSeconds = mlModelSeconds = 0;
            MLLengths<REAL> ml(ctx, nSeqs, maxnode, par.data(), ch.data(), root);
            ml.comm = opt.comm;
            upReady = false;   /* the up-profile slots now hold ML up-profiles */
            ml.setLengths(branchlength.data());
            /* recomputeProfiles(tmatAsDist) (VeryFastTreeImpl.tcc:253-256): plain re-averaging under Jukes-Cantor - the
               identity straight after fastNJ, needed after minimum-evolution NNIs (they leave some profiles stale).  With
               an amino-acid model the averages move from the BLOSUM45 eigen-basis into the model's (transMatToDistanceMat,
               VeryFastTreeImpl.tcc:517-542), so every internal profile is rebuilt from the leaves up. */
            if (opt.aaModel) {
                installAAModel();
                ml.recomputeAverageProfiles();
            } else if (reaverage) ml.recomputeAverageProfiles();
            std::vector<double> loglk;
            const bool parallelLengths = opt.parallelLengths;
            const int64_t maxRound = mllen ? (int64_t) (0.5 + std::log((double) nSeqs) / std::log(2.0)) : 0;
            std::vector<REAL> old((size_t) maxnode);
            bool ratesSet = false;
            for (int64_t iRound = 1; iRound <= maxRound; iRound++) {
                for (int64_t v = 0; v < maxnode; v++) old[(size_t) v] = branchlength[(size_t) v];
                /* opt.parallelLengths (measurements only): the level-parallel rounds of MLLengths.h, which do not
                   follow the one-thread order of the reference */
                if (parallelLengths) ml.optimizeRoundParallel(ftol, atol);
                else if (opt.threads > 1) ml.optimizeRoundThreaded(ftol, atol, opt.threads);
                else ml.optimizeRound(ftol, atol);
                ml.getLengths(branchlength.data());
                double dMaxChange = 0;
                for (int64_t v = 0; v < maxnode; v++) dMaxChange = std::max(dMaxChange, std::fabs((double) old[(size_t) v] - (double) branchlength[(size_t) v]));
                loglk.push_back(ml.treeLogLk(nPos, nLeafGaps));
                /* (the reference never updates its dLastLogLk, so the likelihood clause of its test cannot fire) */
                const bool converged = iRound > 1 && dMaxChange < 0.001;
                if (iRound == 1) {
                    if (gtr) fitGtr(ml, nLeafGaps, ftol, atol);           /* `-gtr`: Jukes-Cantor up to here (VeryFastTreeImpl.tcc:299-302) */
                    ml.setMLRates(nRateCats, nPos, mlRates, mlRateCat);   /* VeryFastTreeImpl.tcc:299-305 */
                    ratesSet = nRateCats > 1;
                }
                if (converged) break;
            }
            if (mlNNI && nSeqs > 3) {
                /* maximum-likelihood NNIs (VeryFastTreeImpl.tcc:311-393): lengths first, then up to round(2 log2 N)
                   rounds of DoNNI(useML) with the reference's convergence rule (one more round after the likelihood or
                   the best NNI stops improving by 0.1, NNI statistics reset for the last one), the rate categories
                   fitted after the first round, and a final pass over all lengths */
                const int64_t MLnniToDo = (int64_t) (0.5 + 2.0 * std::log((double) nSeqs) / std::log(2.0));
                if (opt.threads > 1) ml.optimizeRoundThreaded(ftol, atol, opt.threads);
                else ml.optimizeRound(ftol, atol);
                typename MLLengths<REAL>::NNIParams prm;
                prm.useML = true;
                prm.ftol = ftol;
                prm.atol = atol;
                std::vector<typename MLLengths<REAL>::NNIStats> stats;
                ml.initNNIStats(stats);
                double lastloglk = -1e20;
                bool bConverged = false;
                mlNNIs = 0;
                for (int64_t iMLnni = 0; iMLnni < MLnniToDo; iMLnni++) {
                    double maxDelta;
                    {
                        const std::chrono::steady_clock::time_point ts = std::chrono::steady_clock::now();
                        mlNNIs += opt.threads > 1 ? ml.doNNIThreaded(prm, stats, maxDelta, opt.threads) : ml.doNNI(prm, stats, maxDelta);
                        mlNNISeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - ts).count();
                    }
                    const double ll = ml.treeLogLk(nPos, nLeafGaps);
                    loglk.push_back(ll);
                    const bool bConvergedHere = iMLnni > 0 && (ll < lastloglk + 0.1 || maxDelta < 0.1);
                    if (bConverged) break;
                    if (bConvergedHere) bConverged = true;
                    if (bConverged || iMLnni == MLnniToDo - 2) ml.initNNIStats(stats);
                    lastloglk = ll;
                    if (iMLnni == 0 && !ratesSet) {
                        const std::chrono::steady_clock::time_point ts = std::chrono::steady_clock::now();
                        if (gtr && !gtrFitted) fitGtr(ml, nLeafGaps, ftol, atol);
                        ml.setMLRates(nRateCats, nPos, mlRates, mlRateCat);
                        ratesSet = nRateCats > 1;
                        mlModelSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - ts).count();
                    }
                }
                if (opt.threads > 1) ml.optimizeRoundThreaded(ftol, atol, opt.threads);
                else ml.optimizeRound(ftol, atol);
                loglk.push_back(ml.treeLogLk(nPos, nLeafGaps));   /* "TreeLogLk ML_Lengths2" */
                ml.getLengths(branchlength.data());
                adoptTree(ml.parents(), ml.children());
            }
            if (nBootstrap > 0 && nSeqs > 3) {
                /* testSplitsML (VeryFastTreeImpl.tcc:396-398): SH-like supports from the same column resamples the
                   minimum-evolution supports would use (resampleColumns, NJ.tcc:705-727: Knuth's generator from its
                   default state) */
                KnuthRng rng;
                std::vector<int32_t> col((size_t) nBootstrap * nPos);
                for (size_t t = 0; t < col.size(); t++) {
                    int64_t pos = (int64_t) (rng.rand() * nPos);
                    if (pos < 0) pos = 0;
                    else if (pos == nPos) pos = nPos - 1;
                    col[t] = (int32_t) pos;
                }
                const std::chrono::steady_clock::time_point ts = std::chrono::steady_clock::now();
                const typename MLLengths<REAL>::SplitTests st = ml.testSplits(ftol, atol, nBootstrap, col.data());
                mlSupportSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - ts).count();
                support = st.support;
                mlBadSplits = st.nBadSplits;
                mlSplits = st.nSplits;
                mlWorstDelta = st.worstDelta;
            }
            if (opt.gamma && nRateCats > 1) {   /* `-gamma` (VeryFastTreeImpl.tcc:391-394): after the supports */
                /* (Jukes-Cantor only: treeLogLk's per-site gap terms, NJ.tcc:5236-5252) */
                const typename MLLengths<REAL>::GammaFit g = ml.branchlengthScale(nRateCats, nPos, mlRates, mlRateCat, nLeafGaps >= 0 ? gapsPerPos.data() : nullptr);
                gammaFit[0] = g.loglk;
                gammaFit[1] = g.alpha;
                gammaFit[2] = g.rescale;
                ml.getLengths(branchlength.data());
            }
            mlEvaluations = ml.evaluations();
            mlLaneSteps = ml.laneSteps;
            mlLaneWork = ml.laneWork;
            laneGathers += ml.laneGathers;
            laneGatherBytes += ml.laneGatherBytes;
            return loglk;
        }

        void fitGtr(MLLengths<REAL> &ml, int64_t &nLeafGaps, double ftol, double atol) {
            const typename MLLengths<REAL>::GtrFit g = ml.setMLGtr(leafCodeCounts, nPos, /*mlAccuracy*/1, ftol, atol, opt.threads);
            for (int i = 0; i < 6; i++) gtrRates[i] = g.rates[i];
            for (int i = 0; i < 4; i++) gtrFreq[i] = g.freq[i];
            gtrFitted = true;
            nLeafGaps = -1;   /* treeLogLk's Jukes-Cantor correction is gone with the transition matrix */
        }

        int64_t mlNNIs = 0;
        int64_t mlLaneSteps = 0, mlLaneWork = 0;   /* lockstep steps of the subtree schedule and the quartets / splits judged in them */
        std::vector<int64_t> gapsPerPos;          /* gaps of the unique sequences, by column */
        double gammaFit[3] = {0, 0, 0};           /* `-gamma`: Gamma(nCat) log-likelihood, alpha, the factor the lengths were multiplied by */
        int64_t leafCodeCounts[4];                /* occurrences of codes 0..3 in the unique sequences (setMLGtr) */
        bool gtrFitted = false;
        double gtrRates[6] = {1, 1, 1, 1, 1, 1}, gtrFreq[4] = {0.25, 0.25, 0.25, 0.25};   /* "GTR rates" / "GTR Frequencies" */
        int64_t mlEvaluations = 0, mlBadSplits = 0, mlSplits = 0;   /* "Bad splits: b/n" of the reference's summary line */
        double mlWorstDelta = 0;
        std::vector<REAL> mlRates;           /* what the reference logs as "Rates" ... */
        std::vector<int64_t> mlRateCat;      /* ... and "SiteCategories" (0-based here) */

        /* printNJ (NJ.tcc:2706-2794, no supports, no quoting): depth-first, children in stored order, leaf names with
           their duplicates expanded as (a:0.0,b:0.0), lengths as %.5f (float) / %.9f (double).
           names[k] = name of alignment row k; uniqueFirst[u] = row of unique sequence u; alnNext[k] = next row with the
           same sequence or -1 (Alignment.cpp:494-526). */
        std::string newick(const std::vector<std::string> &names, const std::vector<int64_t> &uniqueFirst,
                           const std::vector<int64_t> &alnNext) const {
            if (root < 0) throw std::invalid_argument("NJDriver::newick before finishRoot");
            const char *fmt = sizeof(REAL) == 4 ? "%.5f" : "%.9f";
            std::string out;
            char buf[64];
            auto firstChildOf = [&](int64_t p) { return p == root ? rootChild[0] : child0[p]; };
            std::vector<std::pair<int64_t, int> > stack;
            stack.push_back(std::make_pair(root, 0));
            while (!stack.empty()) {
                const int64_t node = stack.back().first;
                const int end = stack.back().second;
                stack.pop_back();
                if (node < nSeqs) {
                    if (firstChildOf(parent[node]) != node) out += ",";
                    const int64_t first = uniqueFirst[(size_t) node];
                    if (alnNext[(size_t) first] == -1) out += names[(size_t) first];
                    else {
                        out += "(" + names[(size_t) first] + ":0.0";
                        for (int64_t k = alnNext[(size_t) first]; k >= 0; k = alnNext[(size_t) k]) out += "," + names[(size_t) k] + ":0.0";
                        out += ")";
                    }
                    snprintf(buf, sizeof buf, fmt, (double) branchlength[(size_t) node]);
                    out += ":";
                    out += buf;
                } else if (end) {
                    if (node == root) out += ")";
                    else {
                        if (!support.empty()) {   /* bShowSupport, NJ.tcc:2776-2777 */
                            snprintf(buf, sizeof buf, ")%.3f:", support[(size_t) node]);
                            out += buf;
                        } else out += "):";
                        snprintf(buf, sizeof buf, fmt, (double) branchlength[(size_t) node]);
                        out += buf;
                    }
                } else {
                    if (node != root && firstChildOf(parent[node]) != node) out += ",";
                    out += "(";
                    stack.push_back(std::make_pair(node, 1));
                    if (node == root) {
                        for (int k = 2; k >= 0; k--) stack.push_back(std::make_pair(rootChild[k], 0));
                    } else {
                        stack.push_back(std::make_pair(child1[(size_t) node], 0));
                        stack.push_back(std::make_pair(child0[(size_t) node], 0));
                    }
                }
            }
            out += ";";
            return out;
        }

    private:
        int64_t root = -1, rootChild[3] = {-1, -1, -1};
        bool upReady = false;
        std::vector<double> support;   /* per node, filled by computeSupports */

        struct Besthit {
            int64_t i = -1, j = -1;
            REAL weight = 0, dist = (REAL) 1e20, criterion = (REAL) 1e20;
            int32_t src = -1;   /* top-hits refresh: column of the distance block this record's distance comes from */
        };
        struct Hit {        /* one entry of a top-hits list: 8 bytes in float precision (node ids are below 2^31) - the lists of a
                               million-sequence run are 10^9 entries and several host loops stream them */
            int32_t j;
            REAL dist;
        };
        typedef typename std::conditional<sizeof(REAL) == 4, vft_hit_f32, vft_hit_f64>::type DevHit;

        vft_ctx *ctx;
        NJOptions opt;
        int64_t nSeqs, nPos, maxnodes, maxnode;
        double totdiam;
        std::vector<int64_t> parent, child0, child1;
        std::vector<REAL> diameter, branchlength, selfweightLeaf;
        const REAL *mOut = nullptr;     /* host-mapped mirrors written by the device */
        const int32_t *mN = nullptr;
        bool pending = false;           /* a call that may refresh out-distances has not been waited for yet */
        /* top hits (NJ.h:206-248) */
        int64_t m = 0, q = 0, topvisibleAge = 0;
        std::vector<std::vector<Hit> > hits;
        std::vector<int64_t> age, topvisible, hitSource;
        std::vector<Hit> visible;
        std::vector<int64_t> inTopScratch;   /* resetTopVisible: all -1 between calls */
        std::vector<uint8_t> seenScratch;    /* resetTopVisible: all 0 between calls */
        std::vector<int64_t> gbForced, gbJ;   /* getBestFromTopHits: scratch that keeps its capacity */
        std::vector<int32_t> ujScratch, ujTmp;
        std::vector<Besthit> gbTodo;
        std::vector<Besthit *> gbTodoPtr;
        bool leafBlocks = true;              /* setAllLeafTopHits: vft_leaf_block_distances applies (nucleotides, no matrix) */
        bool devLists = false;               /* the lists have a device copy (vft_tophits_create succeeded) */
        bool refreshOnDevice = true;         /* the per-node merges of a top-hits refresh run on the device (k_th_refresh) */
        bool hostLists = true;               /* the host holds the lists as well: no device lists, second-level lists (their
                                                transfers and the 2nd -> 1st level switch are host code), or VFT_NJ_CHECK */
        std::vector<int32_t> listLen;        /* devLists: length of every node's list */
        std::vector<int32_t> thJ;            /* vft_tophits_join results: scratch that keeps its capacity */
        std::vector<REAL> thD, thC;

        void chk(int rc) {
            if (rc != VFT_OK) throw std::invalid_argument(std::string("NJDriver: ") + vft_last_error(ctx));
        }

        /* the reference's default distance matrix for proteins (VeryFastTree.cpp:96-98 keeps useMatrix on for 20 codes;
           matrixBLOSUM45 + setupDistanceMatrix) */
        void installBlosum45() {
            DistanceTables20 d;
            blosum45Tables<REAL>(d);
            installDistanceTables(d);
        }

        void installDistanceTables(const DistanceTables20 &d) {
            REAL dist[400], cf[400], ev[20], et[20];
            for (int i = 0; i < 20; i++) {
                ev[i] = (REAL) d.eigenval[i];
                et[i] = (REAL) d.eigentot[i];
                for (int j = 0; j < 20; j++) {
                    dist[20 * i + j] = (REAL) d.distances[i][j];
                    cf[20 * i + j] = (REAL) d.codeFreq[i][j];
                }
            }
            chkT("vft_set_distance_matrix", [&]() { return vft_set_distance_matrix(ctx, dist, cf, ev, et); });
        }

        /* before the ML stage: the transition matrix as the averaging basis, then as the likelihood model */
        void installAAModel() {
            TransitionTables20 t;
            createAAModel<REAL>(opt.aaModel, t);
            DistanceTables20 d;
            transitionAsDistanceTables<REAL>(t, d);
            installDistanceTables(d);
            REAL stat[20], statinv[20], eval[20], cf[21 * 20], ei[400], eiT[400];
            for (int i = 0; i < 20; i++) {
                stat[i] = (REAL) t.stat[i];
                statinv[i] = (REAL) t.statinv[i];
                eval[i] = (REAL) t.eigenval[i];
                for (int j = 0; j < 20; j++) {
                    ei[20 * i + j] = (REAL) t.eigeninv[i][j];
                    eiT[20 * i + j] = (REAL) t.eigeninvT[i][j];
                }
            }
            for (int i = 0; i < 21; i++)
                for (int j = 0; j < 20; j++) cf[20 * i + j] = (REAL) t.codeFreq[i][j];
            chkT("vft_set_transition_matrix", [&]() { return vft_set_transition_matrix(ctx, stat, statinv, eval, cf, ei, eiT); });
        }

        /* optional per-entry-point wall-clock accounting (VFT_NJ_PROFILE=1) */
        struct Acc {
            double seconds = 0;
            int64_t calls = 0;
        };
        std::map<std::string, Acc> acc;
        bool profiling = std::getenv("VFT_NJ_PROFILE") != nullptr;
        bool checkJoins = std::getenv("VFT_NJ_CHECK") != nullptr;   /* tools: cross-check the shortcuts against the device */
        int64_t checkFailures = 0;

        /* inclusive wall-clock of a host section (VFT_NJ_PROFILE=1); device calls made inside are also listed on
           their own lines */
        struct Section {
            NJDriver *d;
            const char *name;
            std::chrono::steady_clock::time_point t0;
            Section(NJDriver *d_, const char *n) : d(d_->profiling ? d_ : nullptr), name(n) {
                if (d) t0 = std::chrono::steady_clock::now();
            }
            ~Section() {
                if (!d) return;
                Acc &a = d->acc[name];
                a.seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                a.calls++;
            }
        };

        template<typename F>
        void chkT(const char *name, F f) {
            if (!profiling) {
                chk(f());
                return;
            }
            const auto t0 = std::chrono::steady_clock::now();
            const int rc = f();
            Acc &a = acc[name];
            a.seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            a.calls++;
            chk(rc);
        }

    public:
        void report() const {
            if (!profiling) return;
            for (const auto &kv: acc)
                fprintf(stderr, "  %-28s %9lld calls %9.3f s  %8.1f us/call\n", kv.first.c_str(), (long long) kv.second.calls,
                        kv.second.seconds, 1e6 * kv.second.seconds / (double) (kv.second.calls ? kv.second.calls : 1));
        }

    private:

        /* ---- out-distance mirror */
        int64_t nDiffAllow(int64_t nActive) const {
            return opt.tophitsMult > 0 ? (int64_t) (nActive * opt.staleOutLimit) : 0;
        }

        void drain() {
            if (pending) {
                chkT("vft_synchronize", [&]() { return vft_synchronize(ctx); });
                pending = false;
            }
        }

        REAL value(int64_t v) {
            drain();
            return mOut[v];
        }

        int64_t stamp(int64_t v) {
            drain();
            return mN[v];
        }

        /* Batched form of the lazy refresh inside setCriterion: `pairs` lists the (i, j) a loop is ABOUT to pass to
           setCriterion, all of them, unconditionally (loops with early exits must not use this).  The refresh of a
           node depends only on the node, nActive, the out-profile and totdiam, none of which change inside such a
           loop, so refreshing the stale ones up front in one device call is the same computation. */
        void prefetchStale(int64_t nActive, const std::vector<std::pair<int64_t, int64_t> > &pairs) {
            staleBegin(nActive);
            for (const auto &pr: pairs) staleAdd(pr.first, pr.second);
            staleFlush();
        }

        /* the same without the intermediate pair list: staleBegin, staleAdd for every pair, staleFlush */
        std::vector<int64_t> staleIds;
        int64_t staleActive = 0, staleAllow = 0;

        void staleBegin(int64_t nActive) {
            drain();
            staleActive = nActive;
            staleAllow = nDiffAllow(nActive);
            staleIds.clear();
            if (seenScratch.size() != (size_t) maxnodes) seenScratch.assign((size_t) maxnodes, 0);
        }

        void staleAdd(int64_t i, int64_t j) {
            if (i < 0 || j < 0 || parent[i] >= 0 || parent[j] >= 0) return;
            if (mN[i] - staleActive > staleAllow && !seenScratch[(size_t) i]) {
                seenScratch[(size_t) i] = 1;
                staleIds.push_back(i);
            }
            if (mN[j] - staleActive > staleAllow && !seenScratch[(size_t) j]) {
                seenScratch[(size_t) j] = 1;
                staleIds.push_back(j);
            }
        }

        /* the same from inside an OpenMP region: collect candidates per thread (duplicates allowed), then staleMerge */
        void staleCandidates(int64_t nActive, int64_t allow, int64_t i, int64_t j, std::vector<int64_t> &out) const {
            if (i < 0 || j < 0 || parent[i] >= 0 || parent[j] >= 0) return;
            if (mN[i] - nActive > allow) out.push_back(i);
            if (mN[j] - nActive > allow) out.push_back(j);
        }

        void staleMerge(int64_t nActive, std::vector<int64_t> &ids) {
            if (ids.empty()) return;
            std::sort(ids.begin(), ids.end());
            ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
            const int64_t n = (int64_t) ids.size();
            chkT("vft_out_distances", [&]() { return vft_out_distances(ctx, n, ids.data(), nActive, totdiam); });
            pending = true;
        }

        void staleFlush() {
            if (staleIds.empty()) return;
            for (int64_t v: staleIds) seenScratch[(size_t) v] = 0;
            std::sort(staleIds.begin(), staleIds.end());
            const int64_t n = (int64_t) staleIds.size(), nActive = staleActive;
            chkT("vft_out_distances", [&]() { return vft_out_distances(ctx, n, staleIds.data(), nActive, totdiam); });
            pending = true;
        }

        void prefetchVisible(int64_t nActive, const std::vector<int64_t> &nodes) {
            std::vector<std::pair<int64_t, int64_t> > pairs;
            for (int64_t node: nodes) {
                if (node < 0 || parent[node] >= 0) continue;
                pairs.push_back(std::make_pair(node, visible[node].j));
            }
            prefetchStale(nActive, pairs);
        }

        void setOutDistance(int64_t node, int64_t nActive) {
            if (stamp(node) == nActive) return;
            chkT("vft_out_distances", [&]() { return vft_out_distances(ctx, 1, &node, nActive, totdiam); });
            pending = true;
        }

        void setCriterion(int64_t nActive, Besthit &hit) { /* NJ.tcc:1085-1113 */
            if (hit.i < 0 || hit.j < 0 || parent[hit.i] >= 0 || parent[hit.j] >= 0) return;
            const int64_t allow = nDiffAllow(nActive);
            if (stamp(hit.i) - nActive > allow) setOutDistance(hit.i, nActive);
            if (stamp(hit.j) - nActive > allow) setOutDistance(hit.j, nActive);
            double outI = value(hit.i);
            if (stamp(hit.i) != nActive) outI *= (nActive - 1) / (double) (stamp(hit.i) - 1);
            double outJ = value(hit.j);
            if (stamp(hit.j) != nActive) outJ *= (nActive - 1) / (double) (stamp(hit.j) - 1);
            hit.criterion = (REAL) (hit.dist - (outI + outJ) / (double) (nActive - 2));
        }

        /* setCriterion for pairs whose out-distances are known to be fresh enough (prefetchStale + drain done): pure host
           arithmetic, safe inside OpenMP regions */
        void criterionFresh(int64_t nActive, Besthit &hit) const {
            if (hit.i < 0 || hit.j < 0 || parent[hit.i] >= 0 || parent[hit.j] >= 0) return;
            double outI = mOut[hit.i];
            if (mN[hit.i] != nActive) outI *= (nActive - 1) / (double) (mN[hit.i] - 1);
            double outJ = mOut[hit.j];
            if (mN[hit.j] != nActive) outJ *= (nActive - 1) / (double) (mN[hit.j] - 1);
            hit.criterion = (REAL) (hit.dist - (outI + outJ) / (double) (nActive - 2));
        }

        /* forced: nodes whose out-distance is recomputed in the same call unless it carries the stamp nActive
           (vft_pair_distances_refresh) - one wait instead of two when a list comes with refreshes of its own */
        void setDistCriterionBatch(int64_t nActive, std::vector<Besthit *> &list, int64_t allow = -1,
                                   const std::vector<int64_t> *forced = nullptr) { /* NJ.tcc:1115-1124 */
            const int64_t nF = forced ? (int64_t) forced->size() : 0;
            if (list.empty()) {
                if (nF > 0) {
                    chkT("vft_out_distances", [&]() { return vft_out_distances(ctx, nF, forced->data(), nActive, totdiam); });
                    pending = true;
                }
                return;
            }
            if (allow < 0) allow = nDiffAllow(nActive);
            const int64_t n = (int64_t) list.size();
            std::vector<int64_t> pi(n), pj(n);
            std::vector<REAL> d(n), w(n), c(n);
            for (int64_t t = 0; t < n; t++) {
                pi[t] = list[t]->i;
                pj[t] = list[t]->j;
            }
            chkT("vft_pair_distances", [&]() {
                return vft_pair_distances_refresh(ctx, n, pi.data(), pj.data(), nF, nF ? forced->data() : nullptr, nActive, allow, totdiam,
                                                  d.data(), w.data(), c.data());
            });
            pending = false;   /* the call returned data: the stream has drained */
            for (int64_t t = 0; t < n; t++) {
                list[t]->dist = d[t];
                list[t]->weight = w[t];
                list[t]->criterion = c[t];
            }
        }

        int64_t activeAncestor(int64_t node) const {
            if (node < 0) return node;
            while (parent[node] >= 0) node = parent[node];
            return node;
        }

        /* ---- sorting with the reference's tie rule: ascending key, ties by DESCENDING position.
           The lists a join sorts have ~2m entries (2000 at a million sequences) and a comparison sort through 40-byte
           records was most of a join's host time; keys are packed into integers and sorted by a stable LSD radix sort
           that starts from the positions in descending order - stability then IS the tie rule - and skips the bytes all
           keys share. */
        static uint64_t orderedKey(float x) {
            if (x == 0) x = 0;   /* -0.0 and +0.0 compare equal in the reference's comparator */
            uint32_t u;
            memcpy(&u, &x, 4);
            return (uint64_t) ((u & 0x80000000u) ? ~u : (u | 0x80000000u));
        }

        static uint64_t orderedKey(double x) {
            if (x == 0) x = 0;
            uint64_t u;
            memcpy(&u, &x, 8);
            return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
        }

        /* order = the permutation that sorts keys ascending, equal keys by descending position */
        static void radixOrder(const std::vector<uint64_t> &keys, std::vector<uint32_t> &order) {
            const size_t n = keys.size();
            order.resize(n);
            for (size_t t = 0; t < n; t++) order[t] = (uint32_t) (n - 1 - t);
            if (n < 2) return;
            uint64_t orAll = 0, andAll = ~0ull;
            for (uint64_t k: keys) {
                orAll |= k;
                andAll &= k;
            }
            const uint64_t differ = orAll ^ andAll;
            std::vector<uint32_t> tmp(n);
            for (int b = 0; b < 8; b++) {
                if (((differ >> (8 * b)) & 0xFFu) == 0) continue;
                uint32_t count[257] = {0};
                for (size_t t = 0; t < n; t++) count[((keys[order[t]] >> (8 * b)) & 0xFFu) + 1]++;
                for (int c = 0; c < 256; c++) count[c + 1] += count[c];
                for (size_t t = 0; t < n; t++) tmp[count[(keys[order[t]] >> (8 * b)) & 0xFFu]++] = order[t];
                order.swap(tmp);
            }
        }

        static void permute(std::vector<Besthit> &v, const std::vector<uint32_t> &order) {
            std::vector<Besthit> out(v.size());
            for (size_t t = 0; t < v.size(); t++) out[t] = v[order[t]];
            v.swap(out);
        }

        static void sortByCriterion(std::vector<Besthit> &v) {
            std::vector<uint64_t> keys(v.size());
            for (size_t t = 0; t < v.size(); t++) keys[t] = orderedKey(v[t].criterion);
            std::vector<uint32_t> order;
            radixOrder(keys, order);
            permute(v, order);
        }

        static void sortByIJ(std::vector<Besthit> &v) {
            /* node ids are below 2^31 and -1 marks an empty record: (i + 1, j + 1) packs into 64 bits in order */
            std::vector<uint64_t> keys(v.size());
            for (size_t t = 0; t < v.size(); t++) keys[t] = ((uint64_t) (uint32_t) (v[t].i + 1) << 32) | (uint64_t) (uint32_t) (v[t].j + 1);
            std::vector<uint32_t> order;
            radixOrder(keys, order);
            permute(v, order);
        }

        /* ---- top-hits structures */
        void initTopHits(int64_t m_) {
            m = m_;
            q = (int64_t) (0.5 + opt.tophits2Mult * std::sqrt((double) m));   /* NJ.tcc:199-204 */
            if (!opt.useTopHits2nd || q >= m) q = 0;
            hitSource.assign(maxnodes, -1);
            hits.assign(maxnodes, std::vector<Hit>());
            age.assign(maxnodes, 0);
            visible.assign(maxnodes, Hit{(int32_t) -1, (REAL) 1e20});
            topvisible.assign((size_t) (0.5 + opt.topvisibleMult * m), -1);
            topvisibleAge = 0;
            devLists = false;
            /* (second-level lists keep the host walks of round 2: their transfers and the 2nd -> 1st level switch are host code,
                and the device walks only pay off when the host holds no lists at all; VFT_NJ_HOST_LISTS: tools) */
            if (opt.deviceLists && q == 0)
                devLists = vft_tophits_create(ctx, (int32_t) m, maxnodes) == VFT_OK;
            hostLists = !devLists || q > 0 || checkJoins;
            if (devLists) listLen.assign((size_t) maxnodes, 0);
        }

        int32_t lenOf(int64_t node) const { return devLists ? listLen[(size_t) node] : (int32_t) hits[(size_t) node].size(); }

        /* device copies of host-side lists (stage 1: the refresh and setAllLeafTopHits still build lists on the host) */
        void uploadLists(const std::vector<int64_t> &nodes) {
            if (!devLists || nodes.empty()) return;
            Section sec(this, "[host] list upload (incl. device)");
            const size_t chunk = 2048;
            std::vector<Hit> packed;
            std::vector<int32_t> lens;
            for (size_t a0 = 0; a0 < nodes.size(); a0 += chunk) {
                const size_t cnt = std::min(chunk, nodes.size() - a0);
                packed.resize(cnt * (size_t) m);
                lens.resize(cnt);
#pragma omp parallel for schedule(static) num_threads(opt.hostThreads)
                for (int64_t t = 0; t < (int64_t) cnt; t++) {
                    const std::vector<Hit> &l = hits[(size_t) nodes[a0 + (size_t) t]];
                    lens[(size_t) t] = (int32_t) l.size();
                    listLen[(size_t) nodes[a0 + (size_t) t]] = (int32_t) l.size();
                    if (!l.empty()) memcpy(&packed[(size_t) t * (size_t) m], l.data(), l.size() * sizeof(Hit));
                }
                chkT("vft_tophits_upload", [&]() { return vft_tophits_upload(ctx, (int64_t) cnt, nodes.data() + a0, lens.data(), packed.data()); });
            }
        }

        std::vector<Besthit> hitsToBestHits(const std::vector<Hit> &l, int64_t node) const {
            std::vector<Besthit> out(l.size());
            for (size_t t = 0; t < l.size(); t++) {
                out[t].i = node;
                out[t].j = l[t].j;
                out[t].dist = l[t].dist;
                out[t].criterion = (REAL) 1e20;
                out[t].weight = -1;
            }
            return out;
        }

        void sortSaveBestHits(int64_t node, std::vector<Besthit> &bh, int64_t nIn, int64_t nOutWanted, bool sort = true) {
            /* NJ.tcc:4535-4578 */
            if (sort) sortByCriterion(bh);
            nIn = std::min<int64_t>(nIn, (int64_t) bh.size());
            int64_t nSave = 0, jLast = -1;
            for (int64_t t = 0; t < nIn && nSave < nOutWanted; t++) {
                if (bh[t].i < 0) continue;
                const int64_t j = bh[t].j;
                if (j != node && j != jLast && j >= 0) {
                    nSave++;
                    jLast = j;
                }
            }
            std::vector<Hit> &l = hits[node];
            l.clear();
            jLast = -1;
            for (int64_t t = 0; t < nIn && (int64_t) l.size() < nSave; t++) {
                const int64_t j = bh[t].j;
                if (j != node && j != jLast && j >= 0) {
                    l.push_back(Hit{(int32_t) j, bh[t].dist});
                    jLast = j;
                }
            }
        }

        std::vector<Besthit> transferBestHits(int64_t nActive, int64_t node, const std::vector<Besthit> &old, int64_t nOld,
                                              bool updateDistances) { /* NJ.tcc:4580-4613 */
            std::vector<Besthit> out((size_t) nOld);
            std::vector<Besthit *> todoDist, todoCrit;
            for (int64_t t = 0; t < nOld; t++) {
                const Besthit &o = old[t];
                Besthit &h = out[t];
                h.i = node;
                h.j = activeAncestor(o.j);
                h.dist = o.dist;
                h.weight = o.weight;
                h.criterion = o.criterion;
                if (h.j < 0 || h.j == node) {
                    h.weight = 0;
                    h.dist = (REAL) -1e20;
                    h.criterion = (REAL) 1e20;
                } else if (h.i != o.i || h.j != o.j) {
                    if (updateDistances) todoDist.push_back(&h);
                    else {
                        h.dist = (REAL) -1e20;
                        h.criterion = (REAL) 1e20;
                    }
                } else {
                    if (updateDistances) todoCrit.push_back(&h);
                    else h.criterion = (REAL) 1e20;
                }
            }
            setDistCriterionBatch(nActive, todoDist);
            if (!todoCrit.empty()) {
                std::vector<std::pair<int64_t, int64_t> > pairs;
                for (Besthit *h: todoCrit) pairs.push_back(std::make_pair(h->i, h->j));
                prefetchStale(nActive, pairs);
            }
            for (Besthit *h: todoCrit) setCriterion(nActive, *h);
            return out;
        }

        bool updateBestHit(Besthit &hit, bool updateDist, std::vector<Besthit *> *todo) { /* NJ.tcc:1626-1648 */
            const int64_t i = activeAncestor(hit.i), j = activeAncestor(hit.j);
            if (i < 0 || j < 0 || i == j) {
                hit.i = hit.j = -1;
                hit.weight = 0;
                hit.dist = hit.criterion = (REAL) 1e20;
                return false;
            }
            if (i != hit.i || j != hit.j) {
                hit.i = i;
                hit.j = j;
                if (updateDist) todo->push_back(&hit);
                else {
                    hit.dist = (REAL) -1e20;
                    hit.criterion = (REAL) 1e20;
                }
            }
            return true;
        }

        std::vector<Besthit> uniqueBestHits(int64_t nActive, std::vector<Besthit> &combined) { /* NJ.tcc:4786-4833 */
            for (Besthit &h: combined) updateBestHit(h, false, nullptr);
            sortByIJ(combined);
            std::vector<Besthit> out;
            out.reserve(combined.size());
            int64_t last = -1;
            for (size_t t = 0; t < combined.size(); t++) {
                const Besthit &h = combined[t];
                if (h.i < 0 || h.j < 0) continue;
                if (last >= 0 && combined[last].i == h.i && combined[last].j == h.j) continue;
                out.push_back(h);
                last = (int64_t) t;
            }
            std::vector<Besthit *> todo;
            std::vector<uint8_t> isTodo(out.size(), 0);
            for (size_t t = 0; t < out.size(); t++)
                if (out[t].dist < 0.0) {
                    todo.push_back(&out[t]);
                    isTodo[t] = 1;
                }
            {   /* the lazy refreshes of the hits whose distance is known travel with the pair list */
                drain();
                const int64_t allow = nDiffAllow(nActive);
                std::vector<int64_t> forced;
                for (size_t t = 0; t < out.size(); t++)
                    if (!isTodo[t]) staleCandidates(nActive, allow, out[t].i, out[t].j, forced);
                std::sort(forced.begin(), forced.end());
                forced.erase(std::unique(forced.begin(), forced.end()), forced.end());
                setDistCriterionBatch(nActive, todo, -1, &forced);
            }
            drain();   /* (as in getBestFromTopHits: nothing these hits name is stale any more) */
            for (size_t t = 0; t < out.size(); t++)
                if (!isTodo[t]) criterionFresh(nActive, out[t]);
            return out;
        }

        /* uniqueBestHits(combined lists of the two children of a fresh join), NJ.tcc:4325-4330 + 4786-4833.  Every record
           of the children's lists changes its first node (c0 / c1 -> newnode), so updateBestHit invalidates every distance:
           what is left after the sort by (i, j) and the removal of duplicates is the ascending list of the distinct active
           ancestors of the partners, all of them to be measured against the new node - built here from the ids alone. */
        std::vector<Besthit> uniqueOfJoin(int64_t nActive, int64_t newnode, int64_t c0, int64_t c1) {
            std::vector<int32_t> &js = ujScratch;
            js.clear();
            for (int side = 0; side < 2; side++)
                for (const Hit &h: hits[(size_t) (side ? c1 : c0)]) {
                    const int64_t j = activeAncestor(h.j);
                    if (j >= 0 && j != newnode) js.push_back((int32_t) j);
                }
            {   /* ascending ids: LSD radix sort over the bytes that differ (std::sort costs more than the rest of this function) */
                uint32_t orAll = 0, andAll = ~0u;
                for (int32_t v: js) {
                    orAll |= (uint32_t) v;
                    andAll &= (uint32_t) v;
                }
                std::vector<int32_t> &tmp = ujTmp;
                tmp.resize(js.size());
                for (int b = 0; b < 4; b++) {
                    if ((((orAll ^ andAll) >> (8 * b)) & 0xFFu) == 0) continue;
                    uint32_t count[257] = {0};
                    for (int32_t v: js) count[(((uint32_t) v >> (8 * b)) & 0xFFu) + 1]++;
                    for (int c = 0; c < 256; c++) count[c + 1] += count[c];
                    for (int32_t v: js) tmp[count[((uint32_t) v >> (8 * b)) & 0xFFu]++] = v;
                    js.swap(tmp);
                }
            }
            js.erase(std::unique(js.begin(), js.end()), js.end());
            std::vector<Besthit> out(js.size());
            std::vector<Besthit *> todo(js.size());
            for (size_t t = 0; t < js.size(); t++) {
                out[t].i = newnode;
                out[t].j = js[t];
                out[t].dist = (REAL) -1e20;
                out[t].weight = -1;
                out[t].criterion = (REAL) 1e20;
                todo[t] = &out[t];
            }
            setDistCriterionBatch(nActive, todo);
            return out;
        }

        bool getVisible(int64_t nActive, int64_t node, Besthit &out) { /* NJ.tcc:546-557 */
            if (node < 0 || parent[node] >= 0) return false;
            const Hit &v = visible[node];
            if (v.j < 0 || parent[v.j] >= 0) return false;
            out.i = node;
            out.j = v.j;
            out.dist = v.dist;
            out.criterion = (REAL) 1e20;
            out.weight = -1;
            setCriterion(nActive, out);
            return true;
        }

        /* getVisible when the out-distances of the node and its visible partner are known to be fresh enough
           (prefetchVisible + drain done): host arithmetic only */
        bool getVisibleFresh(int64_t nActive, int64_t node, Besthit &out) const {
            if (node < 0 || parent[node] >= 0) return false;
            const Hit &v = visible[node];
            if (v.j < 0 || parent[v.j] >= 0) return false;
            out.i = node;
            out.j = v.j;
            out.dist = v.dist;
            out.criterion = (REAL) 1e20;
            out.weight = -1;
            criterionFresh(nActive, out);
            return true;
        }

        void updateTopVisible(int64_t nActive, int64_t iIn, const Hit &hit) { /* NJ.tcc:4660-4726 */
            bool placed = false;
            for (size_t t = 0; t < topvisible.size() && !placed; t++) {
                const int64_t node = topvisible[t];
                if (node == iIn) placed = true;
                else if (node < 0 || parent[node] >= 0) {
                    placed = true;
                    topvisible[t] = iIn;
                }
            }
            int64_t posWorst = -1;
            double critWorst = -1e20;
            if (!placed) {
                const int64_t allow = nDiffAllow(nActive);
                for (size_t t = 0; t < topvisible.size() && !placed; t++) {
                    const int64_t node = topvisible[t];
                    Besthit vis;
                    /* getVisible = setCriterion on (node, its visible partner): a lazy refresh if one of them is stale,
                       then arithmetic.  Almost always nothing is stale: test that on the mirrors and do the arithmetic
                       in place (this loop runs over ~1.5 m nodes about twice per join) */
                    bool ok;
                    if (pending) drain();
                    if (node >= 0 && parent[node] < 0 && visible[(size_t) node].j >= 0 && parent[visible[(size_t) node].j] < 0 &&
                        !(mN[node] - nActive > allow) && !(mN[visible[(size_t) node].j] - nActive > allow))
                        ok = getVisibleFresh(nActive, node, vis);
                    else
                        ok = getVisible(nActive, node, vis);
                    if (!ok) {
                        topvisible[t] = iIn;
                        placed = true;
                    } else if (vis.i == hit.j && vis.j == iIn) {
                        placed = true;
                    } else if (vis.criterion >= critWorst) {
                        posWorst = (int64_t) t;
                        critWorst = vis.criterion;
                    }
                }
            }
            if (!placed && posWorst >= 0) {
                Besthit b;
                b.i = iIn;
                b.j = hit.j;
                b.dist = hit.dist;
                b.weight = -1;
                setCriterion(nActive, b);
                if (b.criterion < critWorst) topvisible[posWorst] = iIn;
            }
        }

        void updateVisible(int64_t nActive, const std::vector<Besthit> &list, int64_t count) { /* NJ.tcc:4633-4657 */
            {
                std::vector<int64_t> nodes;
                for (int64_t t = 0; t < count; t++)
                    if (list[t].i >= 0) nodes.push_back(list[t].j);
                prefetchVisible(nActive, nodes);
                drain();
            }
            for (int64_t t = 0; t < count; t++) {
                const Besthit &hit = list[t];
                if (hit.i < 0) continue;
                Besthit vis;
                const bool ok = getVisibleFresh(nActive, hit.j, vis);
                if (!ok || hit.criterion < vis.criterion) {
                    Hit &v = visible[hit.j];
                    v.j = hit.i;
                    v.dist = hit.dist;
                    Section s3(this, "[host]     updateVisible: updateTopVisible");
                    updateTopVisible(nActive, hit.j, v);
                }
            }
        }

        /* resetTopVisible with the join engine: criteria, lazy refreshes and the selection of the best candidates on the device
           (vft_nj_engine_reset_candidates); the host walks the few thousand sorted records.  false: not enough candidates (the
           caller takes the host path). */
        bool resetTopVisibleEngine(int64_t nActive) {
            Section sec(this, "[host] resetTopVisible, device candidates (incl. device)");
            const int64_t nTop = (int64_t) topvisible.size();
            const int64_t K = std::min<int64_t>(nActive, 4 * nTop + 64);
            if (K > 8192 || K < 1) return false;
            std::vector<DevHit> recs((size_t) K);
            int64_t nVisible = 0;
            chkT("vft_nj_engine_reset_candidates", [&]() { return vft_nj_engine_reset_candidates(ctx, nActive, totdiam, (int32_t) K, recs.data(), &nVisible); });
            pending = false;
            int64_t nReal = 0;   /* records the selection returned (all of them when K >= nVisible) */
            while (nReal < K && recs[(size_t) nReal].j >= 0) nReal++;
            if (nReal < std::min(K, nVisible)) return false;
            /* The reference sorts nActive records of which only the first nVisible are real (NJ.tcc:4729-4744): the others are
               zeros - criterion 0, positions above every real record, i.e. in front of the real records with criterion 0 - and
               every one of them reads as the pair (0, 0). */
            if (inTopScratch.size() != (size_t) maxnodes) inTopScratch.assign((size_t) maxnodes, -1);
            std::vector<int64_t> &inTop = inTopScratch;
            std::vector<int64_t> touched, out;
            int64_t zerosLeft = nActive - nVisible, r = 0, t = 0;
            bool complete = true;
            while (t < nVisible && (int64_t) out.size() < nTop) {
                int64_t vI, vJ;
                if (zerosLeft > 0 && (r == nVisible || (r < nReal && !(recs[(size_t) r].criterion < 0)))) {
                    vI = vJ = 0;
                    zerosLeft--;
                } else if (r < nReal) {
                    vI = recs[(size_t) r].j;
                    vJ = (int64_t) recs[(size_t) r].weight;
                    r++;
                } else {
                    complete = false;   /* more candidates needed than were selected */
                    break;
                }
                t++;
                if (inTop[(size_t) vI] != vJ) {
                    out.push_back(vI);
                    inTop[(size_t) vI] = vJ;
                    inTop[(size_t) vJ] = vI;
                    touched.push_back(vI);
                    touched.push_back(vJ);
                }
            }
            for (int64_t v: touched) inTop[(size_t) v] = -1;
            if (!complete) return false;
            for (size_t k = 0; k < topvisible.size(); k++) topvisible[k] = k < out.size() ? out[k] : -1;
            topvisibleAge = 0;
            return true;
        }

        bool engineActive = false;   /* the visible set lives on the device (runEngine) */

        void resetTopVisible(int64_t nActive) { /* NJ.tcc:4728-4784 */
            if (engineActive) {
                if (opt.deviceReset && resetTopVisibleEngine(nActive)) return;
                engineDownloadVisible();
            }
            Section sec(this, "[host] resetTopVisible (incl. device)");
            /* the reference sorts a value-initialised array of nActive records of which only nVisible are filled:
               the zero records take part in the sort and only the first nVisible sorted positions are considered */
            drain();
            /* 1. getVisible of every active node touches setCriterion: refresh the out-distances that are staler than
                  allowed, all in one device call (what prefetchVisible does, without the intermediate pair lists) */
            const int64_t allow = nDiffAllow(nActive);
            std::vector<int64_t> stale, vi;
            {   /* node ranges in parallel; concatenated in range order, vi stays ascending */
                const int nT = opt.hostThreads;
                std::vector<std::vector<int64_t> > viPart((size_t) nT), stalePart((size_t) nT);
#pragma omp parallel for schedule(static, 1) num_threads(opt.hostThreads)
                for (int part = 0; part < nT; part++) {
                    const int64_t lo = maxnode * part / nT, hi = maxnode * (part + 1) / nT;
                    std::vector<int64_t> &v = viPart[(size_t) part], &st = stalePart[(size_t) part];
                    for (int64_t node = lo; node < hi; node++) {
                        if (parent[node] >= 0) continue;
                        const int64_t j = visible[node].j;
                        if (j < 0 || parent[j] >= 0) continue;   /* getVisible fails, NJ.tcc:550-552 */
                        v.push_back(node);
                        if (mN[node] - nActive > allow) st.push_back(node);
                        if (mN[j] - nActive > allow) st.push_back(j);
                    }
                }
                size_t total = 0;
                for (auto &v: viPart) total += v.size();
                vi.reserve(total);
                for (auto &v: viPart) vi.insert(vi.end(), v.begin(), v.end());
                for (auto &v: stalePart) stale.insert(stale.end(), v.begin(), v.end());
            }
            if (!stale.empty()) {
                std::sort(stale.begin(), stale.end());
                stale.erase(std::unique(stale.begin(), stale.end()), stale.end());
                const int64_t n = (int64_t) stale.size();
                chkT("vft_out_distances", [&]() { return vft_out_distances(ctx, n, stale.data(), nActive, totdiam); });
                pending = true;
                drain();
            }
            /* 2. criteria of the nVisible real records; the remaining nActive - nVisible records are zeros */
            const int64_t nVisible = (int64_t) vi.size();
            struct Key {
                uint64_t key;
                uint32_t negPos;   /* ~position: ascending (key, negPos) = criterion ascending, ties by descending position */
            };
            std::unique_ptr<Key[]> keysBuf(new Key[(size_t) nActive]);   /* (no value-initialisation: 16 MB at a million nodes) */
            Key *keys = keysBuf.get();
#pragma omp parallel for schedule(static) num_threads(opt.hostThreads)
            for (int64_t t = 0; t < nActive; t++) {
                REAL crit = 0;
                if (t < nVisible) {
                    Besthit b;
                    b.i = vi[(size_t) t];
                    b.j = visible[(size_t) b.i].j;
                    b.dist = visible[(size_t) b.i].dist;
                    b.criterion = (REAL) 1e20;
                    criterionFresh(nActive, b);
                    crit = b.criterion;
                }
                keys[(size_t) t].key = orderedKey(crit);
                keys[(size_t) t].negPos = ~(uint32_t) t;
            }
            /* Only a prefix of the sorted array is ever looked at (until the top-visible list is full), and the order
               is total: select + sort that prefix, doubling it in the rare case it does not suffice, instead of sorting
               nActive records every m/2 joins. */
            auto before = [](const Key &a, const Key &b) { return a.key != b.key ? a.key < b.key : a.negPos < b.negPos; };
            if (inTopScratch.size() != (size_t) maxnodes) inTopScratch.assign((size_t) maxnodes, -1);
            std::vector<int64_t> &inTop = inTopScratch;
            std::vector<int64_t> touched;
            size_t save = 0;
            int64_t sorted = 0, t = 0;
            std::vector<Key> cand;
            const Key *order = keys;
            {   /* the first `upto` records of the order: every part of the array contributes its own `upto` smallest
                   (selected in parallel); the answer is among those */
                const int64_t upto = std::min<int64_t>(nActive, 4 * (int64_t) topvisible.size() + 64);
                const int nT = opt.hostThreads;
                if ((int64_t) nT * upto * 2 < nActive) {
                    cand.resize((size_t) nT * (size_t) upto);
#pragma omp parallel for schedule(static, 1) num_threads(opt.hostThreads)
                    for (int part = 0; part < nT; part++) {
                        const int64_t lo = nActive * part / nT, hi = nActive * (part + 1) / nT;   /* hi - lo > upto */
                        std::nth_element(keys + lo, keys + lo + upto, keys + hi, before);
                        std::copy(keys + lo, keys + lo + upto, cand.begin() + (size_t) part * (size_t) upto);
                    }
                    std::nth_element(cand.begin(), cand.begin() + upto, cand.end(), before);
                    std::sort(cand.begin(), cand.begin() + upto, before);
                    order = cand.data();
                    sorted = upto;
                }
            }
            while (t < nVisible && save < topvisible.size()) {
                if (t == sorted) {   /* (after the parallel selection: rare; the order is total, so starting over on the whole
                                        array reproduces the prefix already consumed) */
                    const int64_t upto = std::min<int64_t>(nActive, std::max<int64_t>(2 * sorted, 4 * (int64_t) topvisible.size() + 64));
                    const int64_t from = order == keys ? sorted : 0;
                    if (upto < nActive) std::nth_element(keys + from, keys + upto, (keys + nActive), before);
                    std::sort(keys + from, keys + upto, before);
                    order = keys;
                    sorted = upto;
                }
                const int64_t pos = (int64_t) (uint32_t) ~order[(size_t) t++].negPos;
                const int64_t vI = pos < nVisible ? vi[(size_t) pos] : 0;
                const int64_t vJ = pos < nVisible ? visible[(size_t) vI].j : 0;
                if (inTop[(size_t) vI] != vJ) {
                    topvisible[save++] = vI;
                    inTop[(size_t) vI] = vJ;
                    inTop[(size_t) vJ] = vI;
                    touched.push_back(vI);
                    touched.push_back(vJ);
                }
            }
            for (int64_t v: touched) inTop[(size_t) v] = -1;
            while (save < topvisible.size()) topvisible[save++] = -1;
            topvisibleAge = 0;
        }

        /* one-vs-all sweep -> the first k records of the reference's sorted besthits array.  With several ranks: this
           rank's share of the target ids (whole tiles of 64), its local top-k into the exchange buffer, one all-gather, the
           merge under the reference's order on every rank - the same k records a single sweep selects. */
        std::vector<Besthit> sweep(int64_t node, int64_t nActive, int32_t k) {
            std::vector<DevHit> dev((size_t) k);
            const vft_comm *cm = opt.comm;
            if (cm && cm->world > 1) {
                const int64_t bytes = (int64_t) k * (int64_t) sizeof(DevHit);
                if (bytes > cm->d_cap) throw std::invalid_argument("NJDriver: vft_comm device buffers too small for a sweep");
                const int64_t tiles = (maxnode + 63) / 64, per = (tiles + cm->world - 1) / cm->world;
                const int64_t tlo = std::min<int64_t>(tiles, cm->rank * per), thi = std::min<int64_t>(tiles, (cm->rank + 1) * per);
                if (thi > tlo) {
                    const int64_t lo = tlo * 64, hi = std::min<int64_t>(maxnode, thi * 64);
                    chkT("vft_set_shard", [&]() { return vft_set_shard(ctx, lo, hi); });
                    chkT("vft_sweep", [&]() { return vft_sweep(ctx, node, nActive, nDiffAllow(nActive), totdiam, k, nullptr, cm->d_send, nullptr); });
                    chkT("vft_synchronize", [&]() { return vft_synchronize(ctx); });
                } else {   /* more ranks than tiles: this rank contributes an empty list */
                    for (int32_t t = 0; t < k; t++) {
                        dev[t].j = -1;
                        dev[t].dist = dev[t].criterion = (decltype(dev[t].dist)) 1e20;
                        dev[t].weight = 0;
                    }
                    chkT("vft_device_upload", [&]() { return vft_device_upload(ctx, cm->d_send, dev.data(), bytes); });
                }
                if (cm->allgather(cm->user, bytes, 1) != 0) throw std::runtime_error("NJDriver: all-gather of the sweep lists failed");
                chkT("vft_merge_hits", [&]() { return vft_merge_hits(ctx, cm->d_recv, cm->world, k, dev.data(), nullptr); });
                chkT("vft_set_shard", [&]() { return vft_set_shard(ctx, 0, maxnodes); });
            } else {
                chkT("vft_sweep", [&]() { return vft_sweep(ctx, node, nActive, nDiffAllow(nActive), totdiam, k, dev.data(), nullptr, nullptr); });
            }
            pending = false;
            std::vector<Besthit> out((size_t) k);
            for (int32_t t = 0; t < k; t++) {
                out[t].i = dev[t].j >= 0 ? node : -1;
                out[t].j = dev[t].j;
                out[t].weight = (REAL) dev[t].weight;
                out[t].dist = (REAL) dev[t].dist;
                out[t].criterion = (REAL) dev[t].criterion;
            }
            return out;
        }

        /* outProfile over the active list (NJ.tcc:729-815).  opt.outProfileParts = 0: one pass in id order on this GPU (the one-thread
           order).  P >= 2: the list in P blocks of ceil(n / P) consecutive entries; block b is summed by rank b % world
           (vft_out_profile_partial, raw sums to the host), the ranks' blocks are all-gathered through vft_comm's host buffers
           (rank r sends its blocks b = r, r + world, ... as slots 0, 1, ...), put back in block order and added up on every rank
           (vft_out_profile_finish): the result depends on P, not on the number of ranks. */
        int64_t outProfilePartCalls = 0;
        /* the layout of that exchange: block b of P is summed by rank b % world and travels as slot b / world of that rank's share of
           ceil(P / world) slots; entries [i0, i1) of a list of n */
    public:
        static void outProfileBlock(int64_t P, int64_t world, int64_t b, int64_t n, int64_t &owner, int64_t &slot, int64_t &slots, int64_t &i0, int64_t &i1) {
            const int64_t per = (n + P - 1) / P;
            owner = b % world;
            slot = b / world;
            slots = (P + world - 1) / world;
            i0 = std::min(n, b * per);
            i1 = std::min(n, (b + 1) * per);
        }
    private:
        void outProfileFull(const std::vector<int64_t> &active) {
            const int P = opt.outProfileParts;
            const int64_t n = (int64_t) active.size();
            if (P < 2) {
                chkT("vft_out_profile_full", [&]() { return vft_out_profile_full(ctx, n, active.data()); });
                return;
            }
            const vft_comm *cm = opt.comm && opt.comm->world > 1 ? opt.comm : nullptr;
            const int64_t W = cm ? cm->world : 1, rank = cm ? cm->rank : 0;
            int32_t nCodes = 0;
            chk(vft_get_n_codes(ctx, &nCodes));
            const size_t pb = (size_t) nPos * (size_t) (1 + nCodes) * sizeof(REAL);
            int64_t owner, slot, slots, i0, i1;
            outProfileBlock(P, W, 0, n, owner, slot, slots, i0, i1);
            if (cm && (int64_t) (slots * pb) > cm->h_cap) throw std::invalid_argument("NJDriver: the out-profile blocks of a rank do not fit vft_comm's host buffer");
            std::vector<char> mine((size_t) slots * pb, 0), all((size_t) P * pb);
            for (int64_t b = 0; b < P; b++) {
                outProfileBlock(P, W, b, n, owner, slot, slots, i0, i1);
                if (owner != rank) continue;
                chkT("vft_out_profile_partial", [&]() { return vft_out_profile_partial(ctx, n, i1 - i0, active.data() + i0, mine.data() + (size_t) slot * pb); });
                outProfilePartCalls++;
            }
            if (cm) {
                memcpy(cm->h_send, mine.data(), mine.size());
                if (cm->allgather(cm->user, (int64_t) mine.size(), 0) != 0) throw std::runtime_error("NJDriver: all-gather of the out-profile blocks failed");
                for (int64_t b = 0; b < P; b++) {
                    outProfileBlock(P, W, b, n, owner, slot, slots, i0, i1);
                    memcpy(all.data() + (size_t) b * pb, (const char *) cm->h_recv + (size_t) owner * mine.size() + (size_t) slot * pb, pb);
                }
            } else {
                memcpy(all.data(), mine.data(), all.size());   /* (one rank: slot b is block b) */
            }
            chkT("vft_out_profile_finish", [&]() { return vft_out_profile_finish(ctx, P, all.data()); });
        }

        /* setAllLeafTopHits' seed sweeps, several seeds per device call (vft_sweep_batch: its leaf seeds share passes over the targets,
           four per launch, and one batched selection).  Nothing on the device changes between the seeds of that loop - every
           out-distance carries the stamp n, the lists live on the host - so a seed's sweep is the same whenever it runs: the sweeps
           of the next unvisited seeds are taken ahead, and one is wasted only when its seed turns out to be a close neighbour of
           an earlier seed of the batch (a seed covers ~m of n leaves).  One rank only: sharded runs keep a sweep + exchange per seed. */
        std::vector<std::pair<int64_t, std::vector<Besthit>>> seedAhead;
        int64_t seedSweepsWasted = 0;
        std::vector<Besthit> seedSweep(const std::vector<int64_t> &seeds, int64_t s, const std::vector<uint8_t> &visited, int64_t n, int32_t k) {
            const int64_t seed = seeds[(size_t) s];
            for (size_t t = 0; t < seedAhead.size(); t++)
                if (seedAhead[t].first == seed) {
                    std::vector<Besthit> out;
                    out.swap(seedAhead[t].second);
                    seedAhead.erase(seedAhead.begin() + (long) t);
                    return out;
                }
            if ((opt.comm && opt.comm->world > 1) || opt.seedBatch < 2) return sweep(seed, n, k);
            seedSweepsWasted += (int64_t) seedAhead.size();   /* (taken ahead, their seeds visited since) */
            seedAhead.clear();
            std::vector<int64_t> batch(1, seed);
            for (int64_t t = s + 1; t < (int64_t) seeds.size() && (int) batch.size() < opt.seedBatch; t++)
                if (!visited[(size_t) seeds[(size_t) t]]) batch.push_back(seeds[(size_t) t]);
            /* the records are read where the selection left them - the host-mapped result block of each slot (vft_sweep_batch_view) -
               and converted record by record: no copy in between (what bench.py's step times) */
            chkT("vft_sweep_batch", [&]() { return vft_sweep_batch(ctx, (int32_t) batch.size(), batch.data(), n, nDiffAllow(n), totdiam, k, nullptr, nullptr, nullptr); });
            pending = false;
            std::vector<Besthit> first;
            for (size_t b = 0; b < batch.size(); b++) {
                const void *view = nullptr;
                chk(vft_sweep_batch_view(ctx, (int32_t) b, &view, nullptr));
                const DevHit *dev = (const DevHit *) view;
                std::vector<Besthit> out((size_t) k);
                for (int32_t t = 0; t < k; t++) {
                    const DevHit &h = dev[(size_t) t];
                    out[t].i = h.j >= 0 ? batch[b] : -1;
                    out[t].j = h.j;
                    out[t].weight = (REAL) h.weight;
                    out[t].dist = (REAL) h.dist;
                    out[t].criterion = (REAL) h.criterion;
                }
                if (b == 0) first.swap(out);
                else seedAhead.emplace_back(batch[b], std::move(out));
            }
            return first;
        }

        /* vft_leaf_block_distances for nA x nB pairs; with several ranks the rows are split and the three result arrays
           all-gathered through the host buffers (whole rows, padded to equal shares).  Returns the device's code. */
        int leafBlock(int64_t nA, const int64_t *a, int64_t nB, const int64_t *b, int64_t n, REAL *pd, REAL *pw, REAL *pc) {
            const vft_comm *cm = opt.comm;
            if (!cm || cm->world <= 1 || !opt.shardLeafBlocks) return vft_leaf_block_distances(ctx, nA, a, nB, b, n, nDiffAllow(n), totdiam, pd, pw, pc);
            const int64_t W = cm->world, per = (nA + W - 1) / W;
            const int64_t share = per * nB * (int64_t) sizeof(REAL);   /* bytes of ONE array of one rank */
            if (3 * share > cm->h_cap) return -1;                       /* the caller takes fewer rows at a time */
            /* Out-distances are replicated state: a lazy refresh inside a rank's share would happen on that rank only.
               The one caller (setAllLeafTopHits) runs before the first join, when every stamp equals n. */
            for (int64_t t = 0; t < nA; t++)
                if (mN[a[t]] - n > nDiffAllow(n)) throw std::logic_error("NJDriver::leafBlock: stale out-distance in a sharded block");
            const int64_t r0 = std::min(nA, cm->rank * per), r1 = std::min(nA, (cm->rank + 1) * per);
            REAL *mine = (REAL *) cm->h_send;
            std::fill(mine, mine + 3 * per * nB, (REAL) 0);
            {   /* a rank without rows still asks the device for one (discarded) row: every rank must learn the same way
                   whether the block kernel applies to this context at all, or some would skip the collective below */
                const bool has = r1 > r0;
                const int rc = vft_leaf_block_distances(ctx, has ? r1 - r0 : 1, a + (has ? r0 : 0), nB, b, n, nDiffAllow(n), totdiam, mine,
                                                        mine + per * nB, mine + 2 * per * nB);
                if (rc != VFT_OK) return rc;
            }
            if (cm->allgather(cm->user, 3 * share, 0) != 0) throw std::runtime_error("NJDriver: all-gather of a leaf block failed");
            const REAL *all = (const REAL *) cm->h_recv;
            for (int64_t r = 0; r < W; r++) {
                const int64_t q0 = std::min(nA, r * per), q1 = std::min(nA, (r + 1) * per);
                if (q1 <= q0) continue;
                const REAL *src = all + r * 3 * per * nB;
                std::copy(src, src + (q1 - q0) * nB, pd + q0 * nB);
                std::copy(src + per * nB, src + per * nB + (q1 - q0) * nB, pw + q0 * nB);
                std::copy(src + 2 * per * nB, src + 2 * per * nB + (q1 - q0) * nB, pc + q0 * nB);
            }
            return VFT_OK;
        }

        /* criteria of the hits of one top-hits list when every out-distance involved is current (host arithmetic only) */
        void fillCriteria(int64_t nActive, int64_t x, std::vector<REAL> &out) const {
            const std::vector<Hit> &l = hits[(size_t) x];
            out.resize(l.size());
            for (size_t t = 0; t < l.size(); t++) {
                Besthit b2;
                b2.i = x;
                b2.j = l[t].j;
                b2.dist = l[t].dist;
                criterionFresh(nActive, b2);
                out[t] = b2.criterion;
            }
        }

        void setAllLeafTopHits() { /* NJ.tcc:3746-4119, threads == 1 branch, first-level lists */
            Section sec(this, "[host] setAllLeafTopHits (incl. device)");
            const int64_t n = nSeqs;
            double close = opt.tophitsClose;
            if (close < 0) {
                if (opt.fastest && n >= 50000) close = 0.99;
                else {
                    const double logN = std::log((double) n) / std::log(2.0);
                    close = logN / (logN + 2.0);
                }
            }
            std::vector<int64_t> nGaps(n), seeds(n);
            for (int64_t i = 0; i < n; i++) {
                nGaps[i] = (int64_t) (0.5 + nPos - selfweightLeaf[i]);
                seeds[i] = i;
            }
            /* CompareSeeds (NJ.tcc:7285-7299) with the psort tie rule */
            std::sort(seeds.begin(), seeds.end(), [&](int64_t a, int64_t b) {
                if (nGaps[a] != nGaps[b]) return nGaps[a] < nGaps[b];
                if (mOut[a] != mOut[b]) return mOut[a] < mOut[b];
                return a > b;
            });
            std::vector<uint8_t> visited(n, 0);
            for (int64_t s = 0; s < n; s++) {
                const int64_t seed = seeds[s];
                if (visited[seed]) continue;
                visited[seed] = 1;
                std::vector<Besthit> best;
                {
                    Section s2(this, "[host]   setAllLeafTopHits: seed sweep (incl. device)");
                    best = seedSweep(seeds, s, visited, n, (int32_t) (2 * m));
                }
                std::vector<Besthit> copy(best);
                sortSaveBestHits(seed, copy, (int64_t) copy.size(), m, false);
                const double neardist = best[2 * m - 1].dist * close;
                double nearweight = 0;
                for (int64_t t = 0; t < 2 * m; t++) nearweight += best[t].weight;
                nearweight = nearweight / (2.0 * m);
                nearweight *= (1.0 - 2.0 * neardist / 3.0);
                const double nearcover = 1.0 - neardist / 2.0;
                if (q == 0) {
                    /* Without second-level lists the close neighbours of a seed do not influence each other
                       (NJ.tcc:3957-3992): their transferBestHits are ONE pair list on the device, and the per-neighbour
                       sort + save runs in parallel on the host.  Same pairs, same values, same lists. */
                    std::vector<int64_t> cns;
                    for (int64_t iClose = 0; iClose < m; iClose++) {
                        const Besthit &ch = best[iClose];
                        const int64_t cn = ch.j;
                        if (cn < 0 || visited[cn]) continue;
                        const bool isClose = ch.dist <= neardist &&
                                             (ch.weight >= nearweight || ch.weight >= (nPos - nGaps[cn]) * nearcover);
                        const bool identical = ch.dist < 1e-6 && std::fabs(ch.weight - (nPos - nGaps[seed])) < 1e-5 &&
                                               std::fabs(ch.weight - (nPos - nGaps[cn])) < 1e-5;
                        if (isClose || identical) {
                            cns.push_back(cn);
                            visited[cn] = 1;
                        }
                    }
                    const int64_t K = 2 * m, nNb = (int64_t) cns.size();
                    if (nNb == 0) continue;
                    /* nucleotides with %-different distances: every pair is an integer seqDist - one block call
                       (k_leaf_block); the device refuses it for other alphabets / a distance matrix, once */
                    bool block = leafBlocks;
                    std::unique_ptr<Section> sDev(new Section(this, "[host]   setAllLeafTopHits: neighbour distances (incl. device)"));
                    std::vector<int64_t> pi, pj, first((size_t) nNb + 1, 0);
                    std::vector<REAL> pd, pw, pc;
                    if (block) {
                        std::vector<int64_t> cand((size_t) K);
                        for (int64_t t = 0; t < K; t++) cand[(size_t) t] = best[t].j;
                        pd.resize((size_t) (nNb * K));
                        pw.resize((size_t) (nNb * K));
                        pc.resize((size_t) (nNb * K));
                        /* rows of at most ~4M pairs per call keep the result block (3 arrays) within a few tens of MB */
                        int64_t rows = std::max<int64_t>(1, (int64_t) (4000000 / K));
                        if (opt.comm && opt.comm->world > 1 && opt.shardLeafBlocks)   /* the shares of all ranks must fit the host exchange buffer */
                            rows = std::max<int64_t>(opt.comm->world, std::min<int64_t>(rows, opt.comm->h_cap / (3 * K * (int64_t) sizeof(REAL)) - opt.comm->world));
                        for (int64_t a0 = 0; a0 < nNb && block; a0 += rows) {
                            const int64_t cnt = std::min<int64_t>(rows, nNb - a0);
                            int rc = VFT_OK;
                            chkT("vft_leaf_block_distances", [&]() {
                                rc = leafBlock(cnt, cns.data() + a0, K, cand.data(), n, pd.data() + a0 * K, pw.data() + a0 * K, pc.data() + a0 * K);
                                return rc == VFT_ERR_STATE ? VFT_OK : rc;
                            });
                            if (rc == VFT_ERR_STATE) block = leafBlocks = false;
                            else if (rc == -1) throw std::invalid_argument("NJDriver: vft_comm host buffers too small for a leaf block");
                        }
                    }
                    if (!block) {
                        pi.reserve((size_t) (nNb * K));
                        pj.reserve((size_t) (nNb * K));
                        for (int64_t a = 0; a < nNb; a++) {
                            for (int64_t t = 0; t < K; t++) {
                                const int64_t j = best[t].j;
                                if (j < 0 || j == cns[a]) continue;
                                pi.push_back(cns[a]);
                                pj.push_back(j);
                            }
                            first[(size_t) a + 1] = (int64_t) pi.size();
                        }
                        const int64_t nPairs = (int64_t) pi.size();
                        pd.resize((size_t) nPairs);
                        pw.resize((size_t) nPairs);
                        pc.resize((size_t) nPairs);
                        const int64_t maxCall = 1 << 22;
                        for (int64_t p0 = 0; p0 < nPairs; p0 += maxCall) {
                            const int64_t cnt = std::min<int64_t>(maxCall, nPairs - p0);
                            chkT("vft_pair_distances", [&]() {
                                return vft_pair_distances(ctx, cnt, pi.data() + p0, pj.data() + p0, n, nDiffAllow(n), totdiam,
                                                          pd.data() + p0, pw.data() + p0, pc.data() + p0);
                            });
                        }
                    }
                    pending = false;
                    sDev.reset();
                    Section sSort(this, "[host]   setAllLeafTopHits: neighbour sort + save (host threads)");
#pragma omp parallel for schedule(dynamic, 4) num_threads(opt.hostThreads)
                    for (int64_t a = 0; a < nNb; a++) {
                        const int64_t cn = cns[(size_t) a];
                        std::vector<Besthit> nb((size_t) K);
                        int64_t u = block ? a * K : first[(size_t) a];
                        for (int64_t t = 0; t < K; t++) {
                            Besthit &h = nb[(size_t) t];
                            h.i = cn;
                            h.j = best[t].j;
                            if (h.j < 0 || h.j == cn) {   /* transferBestHits, NJ.tcc:4593-4597 */
                                h.weight = 0;
                                h.dist = (REAL) -1e20;
                                h.criterion = (REAL) 1e20;
                                if (block) u++;
                            } else {
                                h.dist = pd[(size_t) u];
                                h.weight = pw[(size_t) u];
                                h.criterion = pc[(size_t) u];
                                u++;
                            }
                        }
                        sortSaveBestHits(cn, nb, K, m);
                    }
                    continue;
                }
                for (int64_t iClose = 0; iClose < m; iClose++) {
                    const Besthit &ch = best[iClose];
                    const int64_t cn = ch.j;
                    if (cn < 0 || visited[cn]) continue;
                    const bool isClose = ch.dist <= neardist &&
                                         (ch.weight >= nearweight || ch.weight >= (nPos - nGaps[cn]) * nearcover);
                    const bool identical = ch.dist < 1e-6 && std::fabs(ch.weight - (nPos - nGaps[seed])) < 1e-5 &&
                                           std::fabs(ch.weight - (nPos - nGaps[cn])) < 1e-5;
                    if (opt.useTopHits2nd && iClose < q && (isClose || identical)) {
                        const int64_t nUse = std::min<int64_t>(q * opt.tophits2Safety, 2 * m);
                        std::vector<Besthit> bc = transferBestHits(n, cn, best, nUse, true);
                        visited[cn] = 1;
                        sortSaveBestHits(cn, bc, nUse, q);
                        hitSource[cn] = seed;
                    } else if (isClose || identical || (opt.fastest && iClose < (q + 1) / 2)) {
                        std::vector<Besthit> nb = transferBestHits(n, cn, best, 2 * m, true);
                        visited[cn] = 1;
                        sortSaveBestHits(cn, nb, 2 * m, m);   /* sorts nb in place, like the reference */
                        /* second level of transfer, NJ.tcc:3993-4012 */
                        for (int64_t iClose2 = 0; iClose2 < q && iClose2 < 2 * m; iClose2++) {
                            const int64_t cn2 = nb[iClose2].j;
                            if (cn2 < 0 || visited[cn2]) continue;
                            const int64_t nUse = std::min<int64_t>(q * opt.tophits2Safety, 2 * m);
                            std::vector<Besthit> bc2 = transferBestHits(n, cn2, nb, nUse, true);
                            visited[cn2] = 1;
                            sortSaveBestHits(cn2, bc2, nUse, q);
                            hitSource[cn2] = cn;
                        }
                    }
                }
            }
            seedSweepsWasted += (int64_t) seedAhead.size();
            seedAhead.clear();
            if (profiling) acc["[count]  setAllLeafTopHits: seed sweeps taken ahead and dropped (their seed became a close neighbour first)"].calls += seedSweepsWasted;
            for (int64_t node = 0; node < n; node++) visible[node] = hits[node][0];
            /* checking phase, NJ.tcc:4052-4119 */
            const int64_t nCheck = q > 0 ? q : (int64_t) (0.5 + 2.0 * std::sqrt((double) m));
            /* The loop below is sequential in the reference (an iteration may rewrite one entry of ANOTHER node's list),
               but almost every iteration is a no-op that only reads: (a) the other list's nCheck-th hit already beats
               this one, (b) the other list already holds this node, (c) its worst hit is not worse than this one.  The
               verdicts are computed for all (node, hit) in parallel first.  A later rewrite replaces the WORST entry of a
               list by a better one, so (a) and (c) can only become "more true" (criteria are constants here: every
               out-distance carries the stamp n); a verdict is withdrawn only where the rewrite touches what it read -
               the rewritten entry itself, and (b)-verdicts of the node that was pushed out.  The sequential pass then does
               the full work for the withdrawn and the undecided ones.  n * nCheck = 63 million iterations of cache misses
               at a million sequences. */
            Section sCheck(this, "[host]   setAllLeafTopHits: checking phase");
            drain();
            bool allFresh = true;
            for (int64_t v = 0; v < n && allFresh; v++) allFresh = mN[v] == n;
            std::vector<uint8_t> verdict;
            if (allFresh && n * nCheck < ((int64_t) 1 << 32)) {
                verdict.assign((size_t) (n * nCheck), 0);
                /* (c) asks for the worst criterion of the OTHER list: a property of that list, computed once per list
                   instead of once per (node, hit) - the lists are 16 GB at a million sequences */
                std::vector<double> worstOf((size_t) n, -1e20);
                std::unique_ptr<Section> sW(new Section(this, "[host]     checking: worst criterion per list (host threads)"));
#pragma omp parallel for schedule(dynamic, 256) num_threads(opt.hostThreads)
                for (int64_t x = 0; x < n; x++) {
                    double dWorst = -1e20;
                    for (const Hit &h: hits[x]) {
                        Besthit b2;
                        b2.i = x;
                        b2.j = h.j;
                        b2.dist = h.dist;
                        criterionFresh(n, b2);
                        if (b2.criterion > dWorst) dWorst = b2.criterion;
                    }
                    worstOf[(size_t) x] = dWorst;
                }
                sW.reset(new Section(this, "[host]     checking: verdicts (host threads)"));
#pragma omp parallel for schedule(dynamic, 256) num_threads(opt.hostThreads)
                for (int64_t node = 0; node < n; node++) {
                    for (int64_t iHit = 0; iHit < nCheck && iHit < (int64_t) hits[node].size(); iHit++) {
                        Besthit bh;
                        bh.i = node;
                        bh.j = hits[node][iHit].j;
                        bh.dist = hits[node][iHit].dist;
                        criterionFresh(n, bh);
                        const std::vector<Hit> &lT = hits[bh.j];
                        if ((int64_t) lT.size() < nCheck) continue;   /* (the sequential pass handles it like the reference) */
                        Besthit chk2;
                        chk2.i = bh.j;
                        chk2.j = lT[nCheck - 1].j;
                        chk2.dist = lT[nCheck - 1].dist;
                        criterionFresh(n, chk2);
                        uint8_t why = chk2.criterion < bh.criterion ? 1 : 0;
                        for (size_t t = 0; t < lT.size() && !why; t++)
                            if (lT[t].j == node) why = 2;
                        if (!why && !(worstOf[(size_t) bh.j] > bh.criterion)) why = 3;
                        verdict[(size_t) (node * nCheck + iHit)] = why;
                    }
                }
            }
            std::vector<std::vector<REAL> > critOf;
            std::unique_ptr<Section> sSeq;
            if (!verdict.empty()) {
                Section sC(this, "[host]     checking: criteria of the lists that will be scanned (host threads)");
                critOf.resize((size_t) n);
                std::vector<uint8_t> need((size_t) n, 0);
                for (int64_t node = 0; node < n; node++)
                    for (int64_t iHit = 0; iHit < nCheck && iHit < (int64_t) hits[node].size(); iHit++)
                        if (!verdict[(size_t) (node * nCheck + iHit)]) need[(size_t) hits[node][iHit].j] = 1;
#pragma omp parallel for schedule(dynamic, 64) num_threads(opt.hostThreads)
                for (int64_t x = 0; x < n; x++)
                    if (need[(size_t) x]) fillCriteria(n, x, critOf[(size_t) x]);
            }
            int64_t nFullChecks = 0;
            bool checkedInParallel = false;
            if (!verdict.empty() && opt.hostThreads > 1) {
                /* The undecided iterations, grouped by the list they may rewrite.  An iteration (node, iHit) reads its own
                   entry hits[node][iHit] - one of the first nCheck of node's list - and reads / rewrites the list of its
                   target x = that entry's partner: the worst entry of x's list, critOf[x], visible[x], and the verdicts of
                   entries that point at x.  Lists are sorted when this phase starts, a rewrite puts a BETTER hit in the place of
                   the worst one, so the worst entry stays behind the first nCheck entries unless a list is rewritten almost
                   completely or ties reach into its head: iterations with different targets then touch disjoint data (apart
                   from the immutable heads), and the reference's order only matters among iterations with the same target.  One
                   thread takes a target and runs its iterations in the reference's order, including those a rewrite re-opens
                   (the verdicts of the pushed-out partner).  Should a rewrite ever reach into a head (checked), everything is
                   rolled back from a log and the sequential pass below does the work. */
                Section sPar(this, "[host]     checking: undecided iterations by target (host threads)");
                const int nT = opt.hostThreads;
                const int64_t total = n * nCheck;
                std::vector<std::vector<std::pair<int32_t, int64_t> > > part((size_t) nT);   /* (target, linear index), index ascending */
#pragma omp parallel for schedule(static, 1) num_threads(opt.hostThreads)
                for (int pt = 0; pt < nT; pt++) {
                    const int64_t lo = n * pt / nT, hi = n * (pt + 1) / nT;
                    std::vector<std::pair<int32_t, int64_t> > &mine = part[(size_t) pt];
                    for (int64_t node = lo; node < hi; node++)
                        for (int64_t iHit = 0; iHit < nCheck && iHit < (int64_t) hits[(size_t) node].size(); iHit++)
                            if (!verdict[(size_t) (node * nCheck + iHit)]) mine.push_back(std::make_pair(hits[(size_t) node][(size_t) iHit].j, node * nCheck + iHit));
                }
                std::vector<int64_t> first((size_t) n + 1, 0);
                for (auto &v: part)
                    for (auto &pr: v) first[(size_t) pr.first + 1]++;
                for (int64_t x = 0; x < n; x++) first[(size_t) x + 1] += first[(size_t) x];
                std::vector<int64_t> byTarget((size_t) first[(size_t) n]), fill(first.begin(), first.end() - 1);
                for (auto &v: part)   /* (parts in node order: every target's indices end up ascending) */
                    for (auto &pr: v) byTarget[(size_t) fill[(size_t) pr.first]++] = pr.second;
                std::vector<int32_t> targets;
                for (int64_t x = 0; x < n; x++)
                    if (first[(size_t) x + 1] > first[(size_t) x]) targets.push_back((int32_t) x);
                struct Undo {
                    int32_t x, pos;
                    Hit oldHit, oldVisible;
                    REAL oldCrit;
                    bool visibleChanged;
                };
                std::vector<std::vector<Undo> > undo((size_t) nT);
                std::vector<std::vector<int64_t> > reopened((size_t) nT);   /* verdicts set to 0 on the way (restored on a rollback) */
                std::vector<std::vector<uint8_t> > reopenedOld((size_t) nT);
                bool hazard = false;
                int64_t done = 0;
#pragma omp parallel for schedule(dynamic, 64) num_threads(opt.hostThreads) reduction(+: done)
                for (int64_t tI = 0; tI < (int64_t) targets.size(); tI++) {
#ifdef _OPENMP
                    const int me = omp_get_thread_num();
#else
                    const int me = 0;
#endif
                    const int64_t x = targets[(size_t) tI];
                    std::vector<Hit> &lT = hits[(size_t) x];
                    std::vector<REAL> &cr = critOf[(size_t) x];
                    if (cr.size() != lT.size()) fillCriteria(n, x, cr);
                    int64_t pos = first[(size_t) x];
                    const int64_t end = first[(size_t) x + 1];
                    std::vector<int64_t> extra;   /* re-opened iterations, kept as a min-heap */
                    int64_t lastIdx = -1;
                    for (;;) {
                        int64_t idx;
                        const bool haveBase = pos < end, haveExtra = !extra.empty();
                        if (!haveBase && !haveExtra) break;
                        if (haveExtra && (!haveBase || extra.front() < byTarget[(size_t) pos])) {
                            std::pop_heap(extra.begin(), extra.end(), std::greater<int64_t>());
                            idx = extra.back();
                            extra.pop_back();
                        } else {
                            idx = byTarget[(size_t) pos++];
                        }
                        if (idx == lastIdx) continue;
                        lastIdx = idx;
                        done++;
                        const int64_t node = idx / nCheck, iHit = idx % nCheck;
                        Besthit bh;
                        bh.i = node;
                        bh.j = x;
                        bh.dist = hits[(size_t) node][(size_t) iHit].dist;
                        criterionFresh(n, bh);
                        Besthit chk2;
                        chk2.i = x;
                        chk2.j = lT[(size_t) (nCheck - 1)].j;
                        chk2.dist = lT[(size_t) (nCheck - 1)].dist;
                        criterionFresh(n, chk2);
                        if (chk2.criterion < bh.criterion) continue;
                        bool found = false;
                        for (size_t t = 0; t < lT.size() && !found; t++) found = lT[t].j == node;
                        if (found) continue;
                        int64_t iWorst = -1;
                        double dWorst = -1e20;
                        for (size_t t = 0; t < cr.size(); t++)
                            if (cr[t] > dWorst) {
                                iWorst = (int64_t) t;
                                dWorst = cr[t];
                            }
                        if (!(dWorst > bh.criterion)) continue;
                        if (iWorst < nCheck) {   /* would change the head of a list: not covered by the grouping */
#pragma omp atomic write
                            hazard = true;
                            break;
                        }
                        Undo u;
                        u.x = (int32_t) x;
                        u.pos = (int32_t) iWorst;
                        u.oldHit = lT[(size_t) iWorst];
                        u.oldCrit = cr[(size_t) iWorst];
                        u.oldVisible = visible[(size_t) x];
                        u.visibleChanged = false;
                        cr[(size_t) iWorst] = bh.criterion;
                        const int64_t gone = lT[(size_t) iWorst].j;
                        if (gone >= 0 && gone < n)
                            for (int64_t ih = 0; ih < nCheck && ih < (int64_t) hits[(size_t) gone].size(); ih++)
                                if (hits[(size_t) gone][(size_t) ih].j == x) {
                                    const int64_t g = gone * nCheck + ih;
                                    reopened[(size_t) me].push_back(g);
                                    reopenedOld[(size_t) me].push_back(verdict[(size_t) g]);
                                    verdict[(size_t) g] = 0;
                                    if (g > idx) {   /* the reference's loop has not been there yet: it will look at it */
                                        extra.push_back(g);
                                        std::push_heap(extra.begin(), extra.end(), std::greater<int64_t>());
                                    }
                                }
                        lT[(size_t) iWorst].j = (int32_t) node;
                        lT[(size_t) iWorst].dist = bh.dist;
                        Besthit v;
                        getVisibleFresh(n, x, v);
                        if (bh.criterion < v.criterion) {
                            visible[(size_t) x] = lT[(size_t) iWorst];
                            u.visibleChanged = true;
                        }
                        undo[(size_t) me].push_back(u);
                    }
                }
                if (!hazard) {
                    checkedInParallel = true;
                    nFullChecks = done;
                } else {
                    for (int pt = 0; pt < nT; pt++) {
                        for (size_t k = undo[(size_t) pt].size(); k-- > 0;) {
                            const Undo &u = undo[(size_t) pt][k];
                            hits[(size_t) u.x][(size_t) u.pos] = u.oldHit;
                            critOf[(size_t) u.x][(size_t) u.pos] = u.oldCrit;
                            if (u.visibleChanged) visible[(size_t) u.x] = u.oldVisible;
                        }
                        for (size_t k = reopened[(size_t) pt].size(); k-- > 0;) verdict[(size_t) reopened[(size_t) pt][k]] = reopenedOld[(size_t) pt][k];
                    }
                    if (profiling) acc["[count]  checking phase: parallel pass rolled back"].calls++;
                }
                (void) total;
            }
            sSeq.reset(new Section(this, "[host]     checking: sequential pass"));
            if (checkedInParallel) goto checkingDone;
            {
            /* the lists are 8 GB at a million sequences and every full iteration walks one of them from cold memory: the
               target list of the NEXT undecided iteration is requested while this one is worked on (a hint only) */
            int64_t ahead = 0;
            auto prefetchNext = [&](int64_t from) {
                if (verdict.empty()) return;
                if (ahead < from) ahead = from;
                const int64_t end = n * nCheck;
                while (ahead < end && verdict[(size_t) ahead]) ahead++;
                if (ahead >= end) return;
                const int64_t nd = ahead / nCheck, ih = ahead % nCheck;
                ahead++;
                if (ih >= (int64_t) hits[(size_t) nd].size()) return;
                const int64_t x = hits[(size_t) nd][(size_t) ih].j;
                if (x < 0 || x >= n) return;
                const char *p0 = (const char *) hits[(size_t) x].data();
                const size_t b0 = hits[(size_t) x].size() * sizeof(Hit);
                for (size_t o = 0; o < b0; o += 64) __builtin_prefetch(p0 + o, 0, 1);
                if ((size_t) x < critOf.size() && !critOf[(size_t) x].empty()) {
                    const char *p1 = (const char *) critOf[(size_t) x].data();
                    const size_t b1 = critOf[(size_t) x].size() * sizeof(REAL);
                    for (size_t o = 0; o < b1; o += 64) __builtin_prefetch(p1 + o, 0, 1);
                }
            };
            for (int64_t node = 0; node < n; node++) {
                for (int64_t iHit = 0; iHit < nCheck && iHit < (int64_t) hits[node].size(); iHit++) {
                    if (!verdict.empty() && verdict[(size_t) (node * nCheck + iHit)]) continue;
                    nFullChecks++;
                    prefetchNext(node * nCheck + iHit + 1);
                    Besthit bh;
                    bh.i = node;
                    bh.j = hits[node][iHit].j;
                    bh.dist = hits[node][iHit].dist;
                    bh.weight = -1;
                    setCriterion(n, bh);
                    std::vector<Hit> &lT = hits[bh.j];
                    Besthit chk2;
                    chk2.i = bh.j;
                    chk2.j = lT[nCheck - 1].j;
                    chk2.dist = lT[nCheck - 1].dist;
                    chk2.weight = -1;
                    setCriterion(n, chk2);
                    if (chk2.criterion < bh.criterion) continue;
                    bool found = false;
                    for (size_t t = 0; t < lT.size() && !found; t++) found = lT[t].j == node;
                    if (found) continue;
                    int64_t iWorst = -1;
                    double dWorst = -1e20;
                    if (!verdict.empty()) {
                        /* every out-distance is current, so the criteria of a list's hits are constants: computed once per
                           list that is ever scanned (in parallel above for the lists known to need it), updated when an
                           entry is replaced - a scan is then m contiguous numbers instead of m x (two random reads) */
                        std::vector<REAL> &cr = critOf[(size_t) bh.j];
                        if (cr.size() != lT.size()) fillCriteria(n, bh.j, cr);
                        for (size_t t = 0; t < cr.size(); t++)
                            if (cr[t] > dWorst) {
                                iWorst = (int64_t) t;
                                dWorst = cr[t];
                            }
                    } else {
                        for (size_t t = 0; t < lT.size(); t++) {
                            Besthit b2;
                            b2.i = bh.j;
                            b2.j = lT[t].j;
                            b2.dist = lT[t].dist;
                            b2.weight = -1;
                            setCriterion(n, b2);
                            if (b2.criterion > dWorst) {
                                iWorst = (int64_t) t;
                                dWorst = b2.criterion;
                            }
                        }
                    }
                    if (dWorst > bh.criterion) {
                        if (!verdict.empty()) critOf[(size_t) bh.j][(size_t) iWorst] = bh.criterion;   /* (x, node) = (node, x) */
                        if (!verdict.empty()) {   /* withdraw what read the entry that is about to change */
                            if (iWorst < nCheck) verdict[(size_t) (bh.j * nCheck + iWorst)] = 0;
                            const int64_t gone = lT[iWorst].j;
                            if (gone >= 0 && gone < n)
                                for (int64_t ih = 0; ih < nCheck && ih < (int64_t) hits[gone].size(); ih++)
                                    if (hits[gone][ih].j == bh.j) verdict[(size_t) (gone * nCheck + ih)] = 0;
                        }
                        lT[iWorst].j = node;
                        lT[iWorst].dist = bh.dist;
                        Besthit v;
                        getVisible(n, bh.j, v);
                        if (bh.criterion < v.criterion) visible[bh.j] = lT[iWorst];
                    }
                }
            }
            }
            checkingDone:
            sSeq.reset();
            if (devLists) {
                std::vector<int64_t> all((size_t) n);
                for (int64_t v = 0; v < n; v++) all[(size_t) v] = v;
                uploadLists(all);
                if (!hostLists)   /* from here on the lists live on the device only (8 GB at a million sequences) */
                    for (auto &l: hits) std::vector<Hit>().swap(l);
            }
            if (profiling) {
                acc["[count]  checking phase: iterations done in full"].calls += nFullChecks;
                acc["[count]  checking phase: verdicts precomputed"].calls += (int64_t) verdict.size();
            }
        }

        Besthit getBestFromTopHits(int64_t node, int64_t nActive) { /* NJ.tcc:4267-4298 */
            if (!devLists) return getBestFromTopHitsHost(node, nActive);
            /* one launch over the device copy of the list; only the best hit comes back */
            drain();
            vft_tophits_best_t r;
            chkT("vft_tophits_best", [&]() {
                return vft_tophits_best(ctx, node, lenOf(node), nActive, nDiffAllow(nActive), totdiam, opt.fastest ? 0 : 1, &r);
            });
            pending = false;
            Besthit best;
            best.i = best.j = -1;
            if (r.j >= 0) {
                best.i = node;
                best.j = r.j;
                best.dist = (REAL) r.dist;
                best.criterion = (REAL) r.criterion;
                best.weight = -1;
            }
            if (checkJoins) {
                const Besthit h = getBestFromTopHitsHost(node, nActive);
                if (h.i != best.i || h.j != best.j || h.dist != best.dist || h.criterion != best.criterion) {
                    fprintf(stderr, "[check] getBestFromTopHits(%lld) at nActive %lld: device (%lld, %lld, %.9g, %.9g) host (%lld, %lld, %.9g, %.9g)\n",
                            (long long) node, (long long) nActive, (long long) best.i, (long long) best.j, (double) best.dist, (double) best.criterion,
                            (long long) h.i, (long long) h.j, (double) h.dist, (double) h.criterion);
                    throw std::runtime_error("NJDriver: device and host getBestFromTopHits differ");
                }
            }
            return best;
        }

        Besthit getBestFromTopHitsHost(int64_t node, int64_t nActive) {
            /* setOutDistance(node) (unless -fastest), the recomputed distances of re-targeted hits and the lazy refreshes
               of setCriterion on every hit all belong to the same window: one device call, one wait.  The list (1 000 hits
               at a million sequences, twice per join) is walked in place: records are built only for the hits whose
               partner was joined since and needs a new distance. */
            drain();
            std::vector<int64_t> &forced = gbForced;
            forced.clear();
            if (!opt.fastest && mN[node] != nActive) forced.push_back(node);
            const std::vector<Hit> &l = hits[(size_t) node];
            const size_t n = l.size();
            const int64_t iA = activeAncestor(node), allow = nDiffAllow(nActive);
            gbJ.resize(n);
            gbTodo.clear();
            const size_t first = forced.size();
            for (size_t t = 0; t < n; t++) {   /* updateBestHit(hit, true, todo), NJ.tcc:1626-1648 */
                const int64_t j = activeAncestor(l[t].j);
                if (iA < 0 || j < 0 || iA == j) {
                    gbJ[t] = -1;
                    continue;
                }
                gbJ[t] = j;
                if (iA != node || j != l[t].j) {
                    Besthit h;
                    h.i = iA;
                    h.j = j;
                    h.dist = l[t].dist;
                    h.weight = -1;
                    h.src = (int32_t) t;
                    gbTodo.push_back(h);
                }
                staleCandidates(nActive, allow, iA, j, forced);
            }
            if (forced.size() > first + 1) {
                std::sort(forced.begin() + first, forced.end());
                forced.erase(std::unique(forced.begin() + first, forced.end()), forced.end());
            }
            gbTodoPtr.clear();
            for (Besthit &h: gbTodo) gbTodoPtr.push_back(&h);
            setDistCriterionBatch(nActive, gbTodoPtr, -1, &forced);
            drain();   /* every out-distance these hits name is fresh enough now: setCriterion is its arithmetic */
            Besthit best;
            best.i = best.j = -1;
            size_t u = 0;
            for (size_t t = 0; t < n; t++) {
                if (gbJ[t] < 0) continue;
                Besthit h;
                if (u < gbTodo.size() && gbTodo[u].src == (int32_t) t) {
                    h = gbTodo[u++];
                    h.src = -1;
                } else {
                    h.i = iA;
                    h.j = gbJ[t];
                    h.dist = l[t].dist;
                    h.weight = -1;
                }
                h.criterion = (REAL) 1e20;
                criterionFresh(nActive, h);
                if (h.criterion < best.criterion) best = h;
            }
            return best;
        }

        /* ---- tiny inputs: no top hits, the visible set of every node (NJ.tcc:2846-2852, 3049-3090, 3686-3744) */
        std::vector<Besthit> visibleAll;

        /* setBestHit (NJ.tcc:3571-3646): the best join partner of `node` (lowest criterion, lowest id on ties, never
           itself); all: every active node's hit against `node`, indexed by id (besthitNew) */
        Besthit bestHitOf(int64_t node, int64_t nActive, std::vector<Besthit> *all) {
            Besthit best;
            best.i = node;
            best.j = -1;
            if (all) all->assign((size_t) maxnodes, Besthit());
            if (maxnode > 8192) {
                /* more nodes than the device's sorted-hit buffer holds (only reachable with top hits switched off on a
                   large input): the sweep without a selection, then the full per-target arrays - allhits[] itself */
                chkT("vft_sweep", [&]() { return vft_sweep(ctx, node, nActive, nDiffAllow(nActive), totdiam, 0, nullptr, nullptr, nullptr); });
                std::vector<REAL> d((size_t) maxnode), w((size_t) maxnode), c((size_t) maxnode);
                chkT("vft_sweep_results", [&]() { return vft_sweep_results(ctx, 0, maxnode, d.data(), w.data(), c.data()); });
                pending = false;
                for (int64_t j = 0; j < maxnode; j++) {
                    if (parent[(size_t) j] >= 0) continue;
                    Besthit h;
                    h.i = node;
                    h.j = j;
                    h.dist = d[(size_t) j];
                    h.weight = w[(size_t) j];
                    h.criterion = c[(size_t) j];
                    if (all) (*all)[(size_t) j] = h;
                    if (j == node) continue;
                    if (best.j < 0 || h.criterion < best.criterion) {   /* ascending j: the lowest id wins ties */
                        best.j = j;
                        best.dist = h.dist;
                        best.weight = h.weight;
                        best.criterion = h.criterion;
                    }
                }
                return best;
            }
            const int32_t k = (int32_t) maxnode;
            std::vector<Besthit> hits = sweep(node, nActive, k);
            for (const Besthit &h: hits) {
                if (h.j < 0) continue;
                if (all) (*all)[(size_t) h.j] = h;
                if (h.j == node || parent[h.j] >= 0) continue;
                if (best.j < 0 || h.criterion < best.criterion || (h.criterion == best.criterion && h.j < best.j)) {
                    best.j = h.j;
                    best.dist = h.dist;
                    best.weight = h.weight;
                    best.criterion = h.criterion;
                }
            }
            return best;
        }

        Besthit fastNJSearch(int64_t nActive) { /* NJ.tcc:3686-3744 */
            Besthit join;
            for (int64_t v = 0; v < maxnode; v++) {
                Besthit &b = visibleAll[(size_t) v];
                if (b.j < 0 || parent[v] >= 0 || parent[b.j] >= 0) continue;
                setCriterion(nActive, b);
                if (b.criterion < join.criterion) join = b;
            }
            if (!opt.fastest) {
                bool changed;
                do {
                    changed = false;
                    visibleAll[(size_t) join.i] = bestHitOf(join.i, nActive, nullptr);
                    if (visibleAll[(size_t) join.i].j != join.j) changed = true;
                    join.j = visibleAll[(size_t) join.i].j;
                    join.weight = visibleAll[(size_t) join.i].weight;
                    join.dist = visibleAll[(size_t) join.i].dist;
                    join.criterion = visibleAll[(size_t) join.i].criterion;
                    visibleAll[(size_t) join.j] = bestHitOf(join.j, nActive, nullptr);
                    if (visibleAll[(size_t) join.j].j != join.i) {
                        changed = true;
                        join.i = visibleAll[(size_t) join.j].j;
                        join.weight = visibleAll[(size_t) join.j].weight;
                        join.dist = visibleAll[(size_t) join.j].dist;
                        join.criterion = visibleAll[(size_t) join.j].criterion;
                    }
                } while (changed);
            }
            return join;
        }

        void visibleJoin(int64_t newnode, int64_t nActive) { /* NJ.tcc:3049-3090; nActive = the count after the join */
            chkT("vft_out_distances", [&]() { return vft_out_distances(ctx, 0, nullptr, nActive, totdiam); });
            pending = true;
            std::vector<Besthit> fromNew;
            visibleAll[(size_t) newnode] = bestHitOf(newnode, nActive, &fromNew);
            for (int64_t v = 0; v < maxnode; v++) {
                if (parent[v] >= 0 || v == newnode) continue;
                Besthit &b = visibleAll[(size_t) v];
                const int64_t old = b.j;
                if (parent[old] < 0) setCriterion(nActive, b);
                if (parent[old] >= 0 || fromNew[(size_t) v].criterion < b.criterion) {
                    b.j = newnode;
                    b.dist = fromNew[(size_t) v].dist;
                    b.criterion = fromNew[(size_t) v].criterion;
                }
            }
        }

        Besthit topHitNJSearch(int64_t nActive) { /* NJ.tcc:4137-4262 */
            Section sec(this, "[host] topHitNJSearch (incl. device)");
            int64_t bestNode;
            for (;;) {
                int64_t nCand = 0;
                bestNode = -1;
                double bestCrit = 1e20;
                Section sScan(this, "[host]   topHitNJSearch: top-visible scan (incl. device)");
                prefetchVisible(nActive, topvisible);
                drain();
                for (int64_t node: topvisible) {
                    Besthit v;
                    if (getVisibleFresh(nActive, node, v)) {
                        nCand++;
                        if (bestNode < 0 || v.criterion < bestCrit) {
                            bestNode = node;
                            bestCrit = v.criterion;
                        }
                    }
                }
                topvisibleAge++;
                if (2 * topvisibleAge > m || (3 * nCand < (int64_t) topvisible.size() && 3 * nCand < nActive)) {
                    if (topvisibleAge <= 2) {
                        for (int64_t node = 0; node < maxnode; node++) {
                            if (parent[node] >= 0) continue;
                            Hit &v = visible[node];
                            int64_t newj = activeAncestor(v.j);
                            if (newj >= 0 && newj != v.j) {
                                if (newj == node) {
                                    newj = 0;
                                    while (parent[newj] >= 0 || newj == node) newj++;
                                }
                                Besthit bh;
                                bh.i = node;
                                bh.j = newj;
                                std::vector<Besthit *> one(1, &bh);
                                setDistCriterionBatch(nActive, one);
                                v.j = newj;
                                v.dist = bh.dist;
                            }
                        }
                    }
                    if (traceEvents) fprintf(stderr, "[event host] reset at join %zu (nActive %lld) tvAge %lld nCand %lld\n", joins.size(), (long long) nActive, (long long) topvisibleAge, (long long) nCand);
                    resetTopVisible(nActive);
                    continue;
                }
                break;
            }
            Besthit join;
            getVisible(nActive, bestNode, join);
            if (opt.fastest) return join;
            Besthit join2 = join;
            bool changed;
            Section sClimb(this, "[host]   topHitNJSearch: hill climbing (incl. device)");
            do {
                changed = false;
                Besthit best = getBestFromTopHits(join2.i, nActive);
                if (best.j != join2.j && best.criterion < join2.criterion) {
                    changed = true;
                    join2 = best;
                }
                best = getBestFromTopHits(join2.j, nActive);
                if (best.j != join2.i && best.criterion < join2.criterion) {
                    changed = true;
                    join2 = best;
                }
                join = join2;
            } while (changed);
            return join;
        }

        void topHitJoin(int64_t newnode, int64_t nActive) { /* NJ.tcc:4306-4533, first-level lists */
            Section sec(this, "[host] topHitJoin (incl. device)");
            const int64_t c0 = child0[newnode], c1 = child1[newnode];
            std::vector<Besthit> unique;
            age[newnode] = (age[c0] + age[c1] + 1) / 2 + 1;
            const int64_t ageLimit = std::max<int64_t>(1, (int64_t) (0.5 + std::log((double) m) / std::log(2.0)));
            bool second = hitSource[c0] >= 0 && hitSource[c1] >= 0;
            const int64_t need = second ? (int64_t) (0.5 + opt.tophits2Refresh * q) : (int64_t) (0.5 + m * opt.tophitsRefresh);
            bool devSorted = false;   /* `unique` is the device's sorted candidate list and hits[newnode] is saved there */
            if (devLists) {
                /* one launch: candidates, distances, criteria, the sort, the decision and the new list (k_th_join) */
                Section s2(this, "[host]   topHitJoin: uniqueBestHits (incl. device)");
                drain();
                const int32_t n0 = lenOf(c0), n1 = lenOf(c1);
                thJ.resize((size_t) (n0 + n1) + 1);
                thD.resize((size_t) (n0 + n1) + 1);
                thC.resize((size_t) (n0 + n1) + 1);
                vft_tophits_join_t info;
                chkT("vft_tophits_join", [&]() {
                    return vft_tophits_join(ctx, newnode, c0, n0, c1, n1, nActive, nDiffAllow(nActive), totdiam, (int32_t) (second ? q : m),
                                            (int32_t) need, age[newnode] <= ageLimit ? 1 : 0, &info, thJ.data(), thD.data(), thC.data());
                });
                pending = false;
                unique.resize((size_t) info.n_unique);
                for (size_t t = 0; t < unique.size(); t++) {
                    Besthit &h = unique[t];
                    h.i = newnode;
                    h.j = thJ[t];
                    h.dist = thD[t];
                    h.criterion = thC[t];
                    h.weight = -1;
                }
                devSorted = info.use_unique != 0;
                if (checkJoins) {
                    std::vector<Besthit> ref = uniqueOfJoin(nActive, newnode, c0, c1);
                    bool same = ref.size() == unique.size();
                    if (same) {
                        sortByCriterion(ref);
                        for (size_t t = 0; same && t < ref.size(); t++)
                            same = ref[t].j == unique[t].j && ref[t].dist == unique[t].dist && ref[t].criterion == unique[t].criterion;
                    }
                    if (!same) {
                        fprintf(stderr, "[check] merge for node %lld at nActive %lld: device %zu candidates, host %zu\n", (long long) newnode,
                                (long long) nActive, unique.size(), ref.size());
                        for (size_t t = 0; t < std::min(ref.size(), unique.size()); t++)
                            if (ref[t].j != unique[t].j || ref[t].dist != unique[t].dist || ref[t].criterion != unique[t].criterion) {
                                fprintf(stderr, "[check]   first difference at %zu: device (%lld, %.9g, %.9g) host (%lld, %.9g, %.9g)\n", t,
                                        (long long) unique[t].j, (double) unique[t].dist, (double) unique[t].criterion, (long long) ref[t].j,
                                        (double) ref[t].dist, (double) ref[t].criterion);
                                break;
                            }
                        throw std::runtime_error("NJDriver: device and host merge of a join differ");
                    }
                }
                if (!devSorted) {   /* the paths below expect the candidates in ascending id order, as uniqueBestHits leaves them */
                    std::sort(unique.begin(), unique.end(), [](const Besthit &a, const Besthit &b) { return a.j < b.j; });
                }
            } else {
                Section s2(this, "[host]   topHitJoin: uniqueBestHits (incl. device)");
                unique = uniqueOfJoin(nActive, newnode, c0, c1);
            }
            const int64_t nUnique = (int64_t) unique.size();
            hits[c0].clear();
            hits[c1].clear();
            bool useUnique = nUnique == nActive - 1 || (age[newnode] <= ageLimit && nUnique >= need);
            if (devLists && useUnique != devSorted) throw std::runtime_error("NJDriver: host and device disagree on the merged list");
            if (!useUnique && second && age[newnode] <= ageLimit) {
                /* switch from 2nd-level to 1st-level top hits, NJ.tcc:4364-4410 */
                int64_t source = activeAncestor(hitSource[c0]);
                if (source == newnode) source = activeAncestor(hitSource[c1]);
                if (source != newnode && source >= 0 && hitSource[source] < 0) {
                    std::vector<Besthit> merge(unique);
                    Besthit first;
                    first.i = newnode;
                    first.j = source;
                    merge.push_back(first);
                    std::vector<Besthit> more = hitsToBestHits(hits[source], newnode);
                    merge.insert(merge.end(), more.begin(), more.end());
                    std::vector<Besthit *> todo;
                    for (size_t t = (size_t) nUnique; t < merge.size(); t++) todo.push_back(&merge[t]);
                    setDistCriterionBatch(nActive, todo);
                    unique = uniqueBestHits(nActive, merge);
                    /* the reference tests the OLD nUnique here (NJ.tcc:4402) */
                    useUnique = nUnique >= (int64_t) (0.5 + m * opt.tophitsRefresh);
                    second = false;
                }
            }
            if (useUnique) {
                if (second) hitSource[newnode] = hitSource[c0];
                const int64_t nSave = std::min(nUnique, second ? q : m);
                if (devSorted) {
                    /* sorted and saved by k_th_join */
                    listLen[(size_t) newnode] = (int32_t) nSave;
                    if (hostLists) {
                        std::vector<Hit> &l = hits[(size_t) newnode];
                        l.resize((size_t) nSave);
                        for (int64_t t = 0; t < nSave; t++) l[(size_t) t] = Hit{(int32_t) unique[(size_t) t].j, unique[(size_t) t].dist};
                    }
                } else {
                    Section s2(this, "[host]   topHitJoin: sortSaveBestHits");
                    sortSaveBestHits(newnode, unique, nUnique, nSave);
                    uploadLists(std::vector<int64_t>(1, newnode));
                }
                visible[newnode] = devSorted ? Hit{(int32_t) unique[0].j, unique[0].dist} : hits[newnode][0];
                {
                    Section s2(this, "[host]   topHitJoin: updateTopVisible");
                    updateTopVisible(nActive, newnode, visible[newnode]);
                }
                {
                    Section s2(this, "[host]   topHitJoin: updateVisible (incl. device)");
                    updateVisible(nActive, unique, nSave);
                }
                return;
            }
            refreshTopHits(newnode, nActive);
        }

        /* the else-branch of topHitJoin (NJ.tcc:4440-4517): new top hits for the new node from a sweep, and for its m closest
           nodes from theirs + its own; then resetTopVisible */
        std::vector<int64_t> lastRefreshWork;   /* the nodes whose lists (and visible hits) the last refresh rewrote */
        void refreshTopHits(int64_t newnode, int64_t nActive) {
            if (traceEvents) fprintf(stderr, "[event %s] refresh for node %lld (nActive %lld)\n", engineConsumed > 0 ? "engine-host" : "host", (long long) newnode, (long long) nActive);
            age[newnode] = 0;
            lastRefreshWork.clear();
            if (opt.fastest) {
                /* NJ.tcc:4454-4459 touches every active node with setCriterion, i.e. refreshes exactly the out-distances
                   that are staler than allowed: that is the lazy pre-pass vft_sweep runs before the sweep below */
            } else {
                chkT("vft_out_distances", [&]() { return vft_out_distances(ctx, 0, nullptr, nActive, totdiam); });
                pending = true;
            }
            Section secR(this, "[host]   topHitJoin: refresh (incl. device)");
            std::vector<Besthit> all;
            {
                Section s2(this, "[host]     refresh: sweep (incl. device)");
                all = sweep(newnode, nActive, (int32_t) (2 * m));
            }
            std::vector<Hit> ownList;   /* sortSaveBestHits(newnode, allhits, ..., m), NJ.tcc:4473: the first m usable records */
            {
                int64_t jLast = -1;
                for (size_t t = 0; t < all.size() && (int64_t) ownList.size() < m; t++) {
                    if (all[t].i < 0) continue;
                    const int64_t j = all[t].j;
                    if (j != newnode && j != jLast && j >= 0) {
                        ownList.push_back(Hit{(int32_t) j, all[t].dist});
                        jLast = j;
                    }
                }
            }
            const bool devRefresh = devLists && q == 0 && refreshOnDevice;
            std::vector<int32_t> devLens;
            std::vector<Hit> devFirst;
            std::vector<int64_t> devWork;
            if (devRefresh) {
                /* one workgroup per node merges on the device (k_th_refresh); the host receives lengths and first hits */
                Section s2(this, "[host]     refresh: device merge (incl. device)");
                std::vector<int64_t> hj(all.size());
                std::vector<REAL> hd(all.size());
                for (size_t u = 0; u < all.size(); u++) {
                    hj[u] = all[u].i < 0 ? -1 : all[u].j;
                    hd[u] = all[u].dist;
                }
                for (int64_t iHit = 0; iHit < m && iHit < (int64_t) all.size(); iHit++) {
                    if (all[iHit].i < 0) continue;
                    const int64_t node = all[iHit].j;
                    if (parent[node] >= 0) continue;
                    devWork.push_back(node);
                }
                std::vector<int32_t> nNew(devWork.size(), (int32_t) m);
                devLens.resize(devWork.size());
                devFirst.resize(devWork.size());
                int rc = VFT_OK;
                chkT("vft_tophits_refresh", [&]() {
                    rc = vft_tophits_refresh(ctx, newnode, (int32_t) all.size(), hj.data(), hd.data(), (int32_t) ownList.size(), ownList.data(),
                                             (int64_t) devWork.size(), devWork.data(), nNew.data(), nActive, nDiffAllow(nActive), totdiam,
                                             devLens.data(), devFirst.data());
                    return rc == VFT_ERR_STATE ? VFT_OK : rc;
                });
                if (rc == VFT_ERR_STATE) {   /* lists too long for the merge kernel: the host merges from now on */
                    refreshOnDevice = false;
                    throw std::runtime_error("NJDriver: top-hit lists too long for the device merge (host lists were released)");
                }
                pending = false;
                listLen[(size_t) newnode] = (int32_t) ownList.size();
                for (size_t t = 0; t < devWork.size(); t++) {
                    const int64_t node = devWork[t];
                    age[(size_t) node] = 0;
                    listLen[(size_t) node] = devLens[t];
                    visible[(size_t) node] = devFirst[t];
                }
                lastRefreshWork = devWork;
                if (engineActive) {   /* the visible set lives on the device: the rewritten lists' first hits and ages go there first */
                    engineUploadNodes(devWork, 0);
                    chkT("vft_nj_engine_nodes_set", [&]() { return vft_nj_engine_nodes_set(ctx, 1, &newnode, nullptr, nullptr, 0); });
                }
                if (!hostLists) {
                    resetTopVisible(nActive);
                    return;
                }
                /* VFT_NJ_CHECK: the host merge below runs as well and every list is compared */
            }
            hits[(size_t) newnode] = ownList;
            /* NJ.tcc:4477-4515 — the reference runs this loop as an OpenMP parallel for: iterations only touch their
               own node's list.  Here: the host parts run in parallel per node, every distance that must be recomputed,
               for ALL nodes, goes to the device as ONE pair list, and every lazy out-distance refresh as one id list. */
            /* one record of a node's merged list: 16 bytes instead of a 32-byte Besthit (a refresh at a million sequences
               handles 3 000 of them for each of 1 000 nodes) */
            struct Ent {
                int32_t j;
                int32_t src;     /* >= 0: column of the distance block; -1: the distance is known; -2: recomputed as a list pair */
                REAL dist, crit;
            };
            struct Work {
                int64_t node, nNew;
                std::vector<Ent> out;
                std::vector<int32_t> listTodo;      /* records whose distance goes to the device as a pair list */
                std::vector<int64_t> staleCand;     /* ends of the records whose distance is known, if stale */
            };
            std::vector<Work> work;
            for (int64_t iHit = 0; iHit < m && iHit < (int64_t) all.size(); iHit++) {
                if (all[iHit].i < 0) continue;
                const int64_t node = all[iHit].j;
                if (parent[node] >= 0) continue;
                age[node] = 0;
                if (nActive <= 2 * m) hitSource[node] = -1;   /* abandon the 2nd-level heuristic */
                Work w;
                w.node = node;
                w.nNew = hitSource[node] >= 0 ? q : m;
                work.push_back(std::move(w));
            }
            const int64_t nW = (int64_t) work.size();
            {   /* setCriterion on every old hit (NJ.tcc:4491-4494): refresh what is stale, once, for all nodes */
                Section s2(this, "[host]     refresh: stale old hits (incl. device)");
                drain();
                const int64_t allow = nDiffAllow(nActive);
                std::vector<std::vector<int64_t> > cand((size_t) opt.hostThreads);
#pragma omp parallel for schedule(static) num_threads(opt.hostThreads)
                for (int64_t t = 0; t < nW; t++) {
#ifdef _OPENMP
                    std::vector<int64_t> &mine = cand[(size_t) omp_get_thread_num()];
#else
                    std::vector<int64_t> &mine = cand[0];
#endif
                    const int64_t node = work[(size_t) t].node;
                    for (const Hit &h: hits[node]) staleCandidates(nActive, allow, node, h.j, mine);
                }
                std::vector<int64_t> ids;
                for (auto &v: cand) ids.insert(ids.end(), v.begin(), v.end());
                staleMerge(nActive, ids);
                drain();
            }
            std::unique_ptr<Section> sHost(new Section(this, "[host]     refresh: transfer + unique (host threads)"));
            /* the transferred hits' partners, re-targeted to their active ancestors: the same for every node */
            std::vector<int64_t> target(all.size());
            for (size_t u = 0; u < all.size(); u++) target[u] = all[u].i < 0 ? -1 : activeAncestor(all[u].j);
            const int64_t allowR = nDiffAllow(nActive);
#pragma omp parallel for schedule(dynamic, 8) num_threads(opt.hostThreads)
            for (int64_t t = 0; t < nW; t++) {
                /* transferBestHits(..., updateDistances = false) + uniqueBestHits' host part (NJ.tcc:4580-4613, 4786-4817) on
                   compact records.  The node's own hits come first, the transferred ones after them; every record is
                   re-targeted to the active ancestor of its partner (updateBestHit); records without a partner or whose
                   partner is the node itself drop out; the sort by partner id keeps, among equal ids, the record that came
                   LAST (the tie rule of the reference's sort: ascending key, descending position) - a transferred record
                   over an old one.  Old hits whose partner changed, and - as in the reference, whose test is dist < 0 -
                   old hits with a negative distance, are recomputed as list pairs. */
                Work &w = work[t];
                const int64_t node = w.node;
                const std::vector<Hit> &l = hits[(size_t) node];
                std::vector<Ent> ents;
                ents.reserve(l.size() + (size_t) (2 * w.nNew));
                for (const Hit &h: l) {
                    const int64_t j = activeAncestor(h.j);
                    if (j < 0 || j == node) continue;
                    Ent e;
                    e.j = (int32_t) j;
                    e.src = (j == h.j && !(h.dist < 0.0)) ? -1 : -2;
                    e.dist = h.dist;
                    e.crit = (REAL) 1e20;
                    ents.push_back(e);
                }
                for (int64_t u = 0; u < 2 * w.nNew && u < (int64_t) target.size(); u++) {
                    const int64_t j = target[(size_t) u];
                    if (j < 0 || j == node) continue;
                    Ent e;
                    e.j = (int32_t) j;
                    e.src = (int32_t) u;   /* the pair (node, target[u]) is entry (t, u) of the distance block */
                    e.dist = (REAL) -1e20;
                    e.crit = (REAL) 1e20;
                    if (node == all[(size_t) u].i && j == all[(size_t) u].j) {   /* the swept node's own record (the sweep lists the
                                                                                   node itself among its hits): distance known */
                        e.dist = all[(size_t) u].dist;
                        e.src = e.dist < 0.0 ? -2 : -1;
                    }
                    ents.push_back(e);
                }
                std::vector<uint64_t> keys(ents.size());
                for (size_t k = 0; k < ents.size(); k++) keys[k] = (uint64_t) (uint32_t) ents[k].j;
                std::vector<uint32_t> order;
                radixOrder(keys, order);
                w.out.reserve(ents.size());
                for (size_t k = 0; k < order.size(); k++) {
                    const Ent &e = ents[order[k]];
                    if (!w.out.empty() && w.out.back().j == e.j) continue;
                    w.out.push_back(e);
                }
                for (size_t u = 0; u < w.out.size(); u++) {
                    const Ent &e = w.out[u];
                    if (e.src == -2) w.listTodo.push_back((int32_t) u);
                    else if (e.src == -1) staleCandidates(nActive, allowR, node, e.j, w.staleCand);
                }
                if (checkJoins) {   /* tools (VFT_NJ_CHECK): the same merge on the reference's records, step by step */
                    std::vector<Besthit> both = hitsToBestHits(hits[(size_t) node], node);
                    for (int64_t u = 0; u < 2 * w.nNew; u++) {
                        const Besthit &o = all[(size_t) u];
                        Besthit h;
                        h.i = node;
                        h.j = target[(size_t) u];
                        h.dist = o.dist;
                        if (h.j < 0 || h.j == node) h.dist = (REAL) -1e20;
                        else if (h.i != o.i || h.j != o.j) {
                            h.dist = (REAL) -1e20;
                            h.src = (int32_t) u;
                        }
                        both.push_back(h);
                    }
                    for (Besthit &h: both) updateBestHit(h, false, nullptr);
                    const std::vector<Besthit> unsorted(both);
                    sortByIJ(both);
                    std::vector<Besthit> ref;
                    int64_t last = -1;
                    for (size_t u = 0; u < both.size(); u++) {
                        const Besthit &h = both[u];
                        if (h.i < 0 || h.j < 0) continue;
                        if (last >= 0 && both[(size_t) last].i == h.i && both[(size_t) last].j == h.j) continue;
                        ref.push_back(h);
                        last = (int64_t) u;
                    }
                    bool same = ref.size() == w.out.size();
                    for (size_t u = 0; same && u < ref.size(); u++) {
                        const Ent &e = w.out[u];
                        const int cls = ref[u].dist < 0.0 ? (ref[u].src >= 0 ? 1 : 2) : 0, mine = e.src >= 0 ? 1 : e.src == -2 ? 2 : 0;
                        same = ref[u].j == e.j && cls == mine && (cls != 1 || ref[u].src == e.src) && (cls != 0 || ref[u].dist == e.dist);
                        if (!same) {
                            for (size_t q2 = 0; q2 < unsorted.size(); q2++)
                                if (unsorted[q2].j == ref[u].j)
                                    fprintf(stderr, "[check]   record at position %zu of %zu (nOld %zu): i %lld j %lld dist %g src %d\n", q2, unsorted.size(),
                                            hits[(size_t) node].size(), (long long) unsorted[q2].i, (long long) unsorted[q2].j, (double) unsorted[q2].dist, unsorted[q2].src);
                        }
                        if (!same)
                            fprintf(stderr, "[check] refresh merge of node %lld differs at %zu: j %lld / %d, class %d / %d, src %d / %d, dist %g / %g\n",
                                    (long long) node, u, (long long) ref[u].j, e.j, cls, mine, ref[u].src, e.src, (double) ref[u].dist, (double) e.dist);
                    }
                    if (!same) {
#pragma omp critical
                        checkFailures++;
                    }
                }
            }
            if (checkJoins && checkFailures) throw std::runtime_error("NJDriver: the compact refresh merge differs from the record-by-record one");
            sHost.reset();
            std::vector<REAL> block;
            int64_t nB = 0;
            {   /* uniqueBestHits, device part (NJ.tcc:4822-4831).  Every transferred record needs the distance of
                   (its node, target[u]): together they are the cross product work x target, which goes to the device as
                   two id lists and comes back as one block of distances (vft_block_distances) - at a million sequences
                   2m^2 = 2 000 000 pairs per refresh, 18 000 refreshes.  The few old hits whose partner was joined in
                   the meantime still go as a pair list.  Both calls refresh the stale out-distances of the nodes they
                   name; records whose distance is known name theirs through the stale set. */
                Section s2(this, "[host]     refresh: recomputed distances (incl. device)");
                for (const Work &w: work) nB = std::max<int64_t>(nB, 2 * w.nNew);
                nB = std::min<int64_t>(nB, (int64_t) target.size());
                std::vector<int64_t> nodesA((size_t) nW);
                for (int64_t t = 0; t < nW; t++) nodesA[(size_t) t] = work[(size_t) t].node;
                std::vector<Besthit> listRecs;
                std::vector<std::pair<int32_t, int32_t> > listRef;   /* (work index, record index) of every list pair */
                for (int64_t t = 0; t < nW; t++)
                    for (int32_t u: work[(size_t) t].listTodo) {
                        Besthit h;
                        h.i = work[(size_t) t].node;
                        h.j = work[(size_t) t].out[(size_t) u].j;
                        listRecs.push_back(h);
                        listRef.push_back(std::make_pair((int32_t) t, u));
                    }
                std::vector<Besthit *> todo(listRecs.size());
                for (size_t k = 0; k < listRecs.size(); k++) todo[k] = &listRecs[k];
                if (profiling) acc["[count]    refresh: pairs recomputed as a list"].calls += (int64_t) todo.size();
                if (profiling) acc["[count]    refresh: pairs recomputed as a block"].calls += nW * nB;
                if (nW > 0 && nB > 0) {
                    block.resize((size_t) (nW * nB));
                    const int64_t allow = nDiffAllow(nActive);
                    chkT("vft_block_distances", [&]() {
                        return vft_block_distances(ctx, nW, nodesA.data(), nB, target.data(), nActive, allow, totdiam, block.data());
                    });
                    pending = false;
                }
                setDistCriterionBatch(nActive, todo);
                for (size_t k = 0; k < listRecs.size(); k++) {
                    Ent &e = work[(size_t) listRef[k].first].out[(size_t) listRef[k].second];
                    e.dist = listRecs[k].dist;
                    e.crit = listRecs[k].criterion;
                }
                {   /* (candidates were collected before the two calls above: what those refreshed is skipped on the device) */
                    std::vector<int64_t> ids;
                    for (Work &w: work) ids.insert(ids.end(), w.staleCand.begin(), w.staleCand.end());
                    staleMerge(nActive, ids);
                }
                drain();
            }
            std::unique_ptr<Section> sSave(new Section(this, "[host]     refresh: criteria + sort + save (host threads)"));
#pragma omp parallel for schedule(dynamic, 8) num_threads(opt.hostThreads)
            for (int64_t t = 0; t < nW; t++) {
                /* criteria of the merged list (setCriterion's arithmetic: every out-distance involved is fresh enough now),
                   then sortSaveBestHits (NJ.tcc:4535-4578): ascending criterion, ties by descending position, the first nNew */
                Work &w = work[t];
                const int64_t node = w.node;
                std::vector<uint64_t> keys(w.out.size());
                for (size_t u = 0; u < w.out.size(); u++) {
                    Ent &e = w.out[u];
                    if (e.src >= 0) e.dist = block[(size_t) (t * nB + e.src)];
                    if (e.src != -2) {
                        Besthit h;
                        h.i = node;
                        h.j = e.j;
                        h.dist = e.dist;
                        h.criterion = (REAL) 1e20;
                        criterionFresh(nActive, h);
                        e.crit = h.criterion;
                    }
                    keys[u] = orderedKey(e.crit);
                }
                std::vector<uint32_t> order;
                radixOrder(keys, order);
                std::vector<Hit> &l = hits[(size_t) node];
                l.clear();
                for (size_t k = 0; k < order.size() && (int64_t) l.size() < w.nNew; k++) {
                    const Ent &e = w.out[order[k]];
                    l.push_back(Hit{e.j, e.dist});
                }
                visible[(size_t) node] = l[0];
            }
            sSave.reset();
            if (devRefresh) {   /* VFT_NJ_CHECK: the device's lists against the host merge */
                bool same = (int64_t) devWork.size() == nW;
                std::vector<Hit> got((size_t) m);
                for (int64_t t = 0; same && t <= nW; t++) {
                    const int64_t node = t < nW ? work[(size_t) t].node : newnode;
                    if (t < nW) same = devWork[(size_t) t] == node;
                    int32_t len = 0;
                    chk(vft_tophits_download(ctx, node, &len, got.data()));
                    const std::vector<Hit> &want = hits[(size_t) node];
                    same = same && len == (int32_t) want.size();
                    for (size_t u = 0; same && u < want.size(); u++) same = got[u].j == want[u].j && got[u].dist == want[u].dist;
                    if (!same) fprintf(stderr, "[check] refresh of node %lld (new node %lld, nActive %lld): the device list differs from the host merge (%d / %zu entries)\n",
                                       (long long) node, (long long) newnode, (long long) nActive, (int) len, want.size());
                }
                if (!same) throw std::runtime_error("NJDriver: device and host top-hits refresh differ");
            } else if (devLists) {
                std::vector<int64_t> changed((size_t) nW + 1);
                for (int64_t t = 0; t < nW; t++) changed[(size_t) t] = work[(size_t) t].node;
                changed[(size_t) nW] = newnode;
                uploadLists(changed);
            }
            resetTopVisible(nActive);
        }
    };
}

#endif
