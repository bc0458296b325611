// C entry point of the host NJ driver (include/vft_host.h).  Plain C++11, no HIP: links against libvft_hip.so.
#include "../../include/vft_host.h"

#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include <type_traits>

#include "NJDriver.h"
#include "KnuthRng.h"
#include "MLLengths.h"
#include "GtrModel.h"
#include "AAModels.h"

// CRC-32 (the zlib polynomial) of the last join order this process produced, per chunk of joins: lets a caller that only asked
// for the tree (vft_nj_newick at a million sequences) compare the join order with a prefix of the reference's `Join` lines
static std::mutex gJoinCrcMutex;
static std::vector<uint32_t> gJoinCrc;
static int64_t gJoinCrcChunk = 10000, gJoinCrcJoins = 0;

static uint32_t crc32Bytes(const unsigned char *p, size_t n) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

template<typename JOINS>
static void recordJoinCrc(const JOINS &js) {
    std::lock_guard<std::mutex> lock(gJoinCrcMutex);
    gJoinCrc.clear();
    gJoinCrcJoins = (int64_t) js.size();
    std::vector<int32_t> buf((size_t) gJoinCrcChunk * 3);
    for (size_t k0 = 0; k0 + (size_t) gJoinCrcChunk <= js.size(); k0 += (size_t) gJoinCrcChunk) {
        for (size_t k = 0; k < (size_t) gJoinCrcChunk; k++) {
            buf[3 * k] = (int32_t) js[k0 + k].i;
            buf[3 * k + 1] = (int32_t) js[k0 + k].j;
            buf[3 * k + 2] = (int32_t) js[k0 + k].newnode;
        }
        gJoinCrc.push_back(crc32Bytes((const unsigned char *) buf.data(), buf.size() * 4));
    }
    /* the joins behind the last complete chunk, as one shorter chunk (a finished run is compared to its last join) */
    const size_t done = js.size() / (size_t) gJoinCrcChunk * (size_t) gJoinCrcChunk;
    if (done < js.size()) {
        for (size_t k = 0; done + k < js.size(); k++) {
            buf[3 * k] = (int32_t) js[done + k].i;
            buf[3 * k + 1] = (int32_t) js[done + k].j;
            buf[3 * k + 2] = (int32_t) js[done + k].newnode;
        }
        gJoinCrc.push_back(crc32Bytes((const unsigned char *) buf.data(), (js.size() - done) * 3 * 4));
    }
}

extern "C" int vft_nj_last_join_crcs(int64_t *chunk, int64_t *nJoins, uint32_t *crcs, int64_t cap, int64_t *nCrcs) {
    std::lock_guard<std::mutex> lock(gJoinCrcMutex);
    if (chunk) *chunk = gJoinCrcChunk;
    if (nJoins) *nJoins = gJoinCrcJoins;
    if (nCrcs) *nCrcs = (int64_t) gJoinCrc.size();
    if (crcs)
        for (int64_t k = 0; k < cap && k < (int64_t) gJoinCrc.size(); k++) crcs[k] = gJoinCrc[(size_t) k];
    return VFT_OK;
}

/* wall-clock of the stages of the last vft_nj_newick / vft_nj_ml_newick of this process (vft_nj_last_stage_seconds) */
static double gStage[8];
static int64_t gLanes[4];
static int64_t gWalkDual[2];   /* SPR chains of the last tree: dual commands sent, continuations the walk server ran on its own (vft_nj_last_walk_dual) */
static int64_t gLaneExchange[2];   /* lanes across ranks of the last tree: all-gathers, bytes received (vft_nj_last_lane_exchange) */
static double gGamma[3];   /* `-gamma` of the last tree: Gamma(nCat) log-likelihood, alpha, length factor (vft_nj_last_gamma) */

/* treePartitioning (NJ.tcc:5540-5750) on a tree given as arrays: pure host code, no context (tests pin it to the reference's own
   partitions).  out[cap] receives the subtree roots in the reference's hand-out order. */
extern "C" int vft_tree_partitioning(int64_t nNodes, const int64_t *childIn, int64_t root, int32_t penalty, int32_t threads, int32_t window,
                                     int64_t *out, int64_t cap, int64_t *nOut, double *speedup) {
    if (nNodes < 4 || !childIn || root < 0 || root >= nNodes || penalty < 0 || threads < 1 || !nOut) return VFT_ERR_INVALID;
    if (threads > nNodes) threads = (int32_t) nNodes;   /* (more threads than nodes cannot change the partition) */
    /* the arrays come from the caller: every child id in range and every node the child of at most one parent - a tree, so the walk
       below ends and stays inside the arrays */
    std::vector<char> seen((size_t) nNodes, 0);
    for (int64_t i = 0; i < 3 * nNodes; i++) {
        const int64_t ch = childIn[i];
        if (ch < -1 || ch >= nNodes || ch == root) return VFT_ERR_INVALID;
        if (ch >= 0) {
            if (seen[(size_t) ch]) return VFT_ERR_INVALID;
            seen[(size_t) ch] = 1;
        }
    }
    std::vector<int64_t> child(childIn, childIn + 3 * nNodes), order;
    std::vector<std::pair<int64_t, int>> stack(1, std::make_pair(root, 0));
    while (!stack.empty()) {   /* post-order of the internal nodes, children in stored order */
        const int64_t v = stack.back().first;
        const int k = stack.back().second;
        if ((int64_t) stack.size() > nNodes) return VFT_ERR_INVALID;   /* (a cycle that does not pass through the root) */
        if (k < 3 && child[3 * v + k] >= 0) {
            stack.back().second++;
            stack.push_back(std::make_pair(child[3 * v + k], 0));
        } else {
            stack.pop_back();
            if (child[3 * v] >= 0) order.push_back(v);
        }
    }
    double sp = 0;
    const std::vector<int64_t> res = veryfasttree::MLLengths<float>::partitionTree(nNodes, child, root, order, penalty, threads, window > 0 ? window : 50, &sp);
    *nOut = (int64_t) res.size();
    if (speedup) *speedup = sp;
    for (int64_t k = 0; out && k < cap && k < (int64_t) res.size(); k++) out[k] = res[(size_t) k];
    return VFT_OK;
}

/* the exchange layout of the lanes across ranks (MLLengths::laneShare / laneRecord), for the CPU test that runs it over gloo:
   out[0] = per, out[1] = k0, out[2] = k1 for rank `rank` of `world`; out[3] = the record index of item `item` in the gathered buffer */
extern "C" int vft_nj_lane_share(int64_t nItems, int32_t world, int32_t rank, int64_t item, int64_t *out) {
    if (nItems < 0 || world < 1 || rank < 0 || rank >= world || !out) return VFT_ERR_INVALID;
    size_t per, k0, k1;
    veryfasttree::MLLengths<float>::laneShare((size_t) nItems, (size_t) world, (size_t) rank, per, k0, k1);
    out[0] = (int64_t) per;
    out[1] = (int64_t) k0;
    out[2] = (int64_t) k1;
    out[3] = per > 0 && item >= 0 ? (int64_t) veryfasttree::MLLengths<float>::laneRecord((size_t) item, per) : -1;
    return VFT_OK;
}

/* the layout of the out-profile blocks' exchange (NJDriver::outProfileBlock), for the CPU test that runs it over gloo:
   out[0] = owner rank of block `block` of `parts`, out[1] = its slot in that rank's share, out[2] = slots per share, out[3], out[4] = the
   block's entries [i0, i1) of a list of n */
extern "C" int vft_nj_out_profile_block(int32_t parts, int32_t world, int32_t block, int64_t n, int64_t *out) {
    if (parts < 1 || world < 1 || block < 0 || block >= parts || n < 0 || !out) return VFT_ERR_INVALID;
    veryfasttree::NJDriver<float>::outProfileBlock(parts, world, block, n, out[0], out[1], out[2], out[3], out[4]);
    return VFT_OK;
}

extern "C" int vft_nj_last_lane_exchange(int64_t *out) {
    if (!out) return VFT_ERR_INVALID;
    out[0] = gLaneExchange[0];
    out[1] = gLaneExchange[1];
    return VFT_OK;
}

extern "C" int vft_nj_last_walk_dual(int64_t *out) {
    if (!out) return VFT_ERR_INVALID;
    out[0] = gWalkDual[0];
    out[1] = gWalkDual[1];
    return VFT_OK;
}

extern "C" int vft_nj_last_gamma(double *out) {
    if (!out) return VFT_ERR_INVALID;
    for (int i = 0; i < 3; i++) out[i] = gGamma[i];
    return VFT_OK;
}

extern "C" int vft_nj_last_stage_seconds(double *seconds, int64_t *counts) {
    if (seconds)
        for (int i = 0; i < 8; i++) seconds[i] = gStage[i];
    if (counts)
        for (int i = 0; i < 4; i++) counts[i] = gLanes[i];
    return VFT_OK;
}

static veryfasttree::NJOptions toOptions(const vft_nj_options *o) {
    veryfasttree::NJOptions opt;
    if (o) {
        opt.fastest = o->fastest != 0;
        opt.tophitsMult = o->tophits_mult;
        opt.tophitsClose = o->tophits_close;
        opt.tophitsRefresh = o->tophits_refresh;
        opt.topvisibleMult = o->topvisible_mult;
        opt.staleOutLimit = o->stale_out_limit;
        opt.fResetOutProfile = o->f_reset_out_profile;
        opt.nResetOutProfile = o->n_reset_out_profile;
        opt.useTopHits2nd = o->use_tophits_2nd != 0;
        opt.tophits2Safety = o->tophits2_safety;
        opt.tophits2Mult = o->tophits2_mult;
        opt.tophits2Refresh = o->tophits2_refresh;
        opt.scoredist = o->scoredist != 0 || o->aa_model != 0;
        opt.aaModel = o->aa_model;
        opt.comm = o->comm;
        opt.threads = o->threads > 1 ? o->threads : 1;
        opt.gamma = o->gamma != 0;
        opt.outProfileParts = o->out_profile_parts >= 2 ? o->out_profile_parts : 0;
        if (o->debug_flags & VFT_NJ_DEBUG_HOST_JOINS) opt.deviceJoins = false;
        if (o->debug_flags & VFT_NJ_DEBUG_HOST_LISTS) opt.deviceLists = false;
        if (o->debug_flags & VFT_NJ_DEBUG_HOST_RESET) opt.deviceReset = false;
        if (o->debug_flags & VFT_NJ_DEBUG_NO_WALK_SERVER) opt.walkServer = false;
        if (o->debug_flags & VFT_NJ_DEBUG_NO_WALK_DUAL) opt.walkDual = false;
        if (o->debug_flags & VFT_NJ_DEBUG_SEED_BY_SEED) opt.seedBatch = 1;
        if (o->debug_flags & VFT_NJ_DEBUG_LEVEL_LENGTHS) opt.parallelLengths = true;
        if (o->debug_flags & VFT_NJ_SHARD_LEAF_BLOCKS) opt.shardLeafBlocks = true;
    }
    return opt;
}

template<typename REAL>
static std::string runTree(vft_ctx *ctx, const uint8_t *codes, int64_t nSeqs, int64_t nPos, const vft_nj_options *o,
                           bool meLengths, int32_t nBootstrap, const int64_t *uniqueFirst, const int64_t *alnNext,
                           int64_t nAll, const char *names, std::vector<double> &loglk, std::vector<double> &rates,
                           std::vector<int64_t> &ratecat, double *gtrOut) {
    veryfasttree::NJDriver<REAL> drv(ctx, codes, nSeqs, nPos, toOptions(o));
    for (double &x: gStage) x = 0;
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(now() - t).count(); };
    std::chrono::steady_clock::time_point t0 = now();
    recordJoinCrc(drv.run(-1));
    drv.finishRoot();
    gStage[0] = since(t0);
    static const bool stageTrace = std::getenv("VFT_STAGE_TRACE") != nullptr;   /* tools: the stages as they end, on stderr */
    if (stageTrace) fprintf(stderr, "[stage] NJ phase done after %.1f s\n", gStage[0]);
    t0 = now();
    if (o && o->me_nni) drv.meNNIRounds(o->spr);
    gStage[1] = since(t0);
    if (stageTrace) fprintf(stderr, "[stage] minimum-evolution NNIs + SPR done after %.1f s\n", gStage[1]);
    gStage[2] = drv.meSPRSeconds;
    t0 = now();
    if (meLengths) drv.updateBranchLengths();
    const bool ml = o && (o->mllen || o->ml_nni);
    if (nBootstrap > 0 && !ml) drv.computeSupports(nBootstrap);
    gStage[3] = since(t0);
    t0 = now();
    if (ml) {
        if (!meLengths) throw std::invalid_argument("vft_nj_ml_newick: the ML stage needs me_lengths (updateBranchLengths runs first)");
        /* with supports: SH-like (testSplitsML) instead of the local bootstrap */
        const int32_t nCat = o->mllen ? o->mllen : (o->ml_nni > 1 ? o->ml_nni : 1);
        loglk = drv.mlLengths(nCat, nBootstrap, o->mllen != 0, o->ml_nni != 0, o->me_nni != 0, o->gtr != 0);
        if (gtrOut) {
            for (int i = 0; i < 6; i++) gtrOut[i] = drv.gtrRates[i];
            for (int i = 0; i < 4; i++) gtrOut[6 + i] = drv.gtrFreq[i];
        }
        rates.assign(drv.mlRates.begin(), drv.mlRates.end());
        ratecat = drv.mlRateCat;
    }
    gStage[4] = since(t0);
    gStage[5] = drv.mlNNISeconds;
    gStage[6] = drv.mlSupportSeconds;
    gStage[7] = drv.mlModelSeconds;
    for (int i = 0; i < 3; i++) gGamma[i] = drv.gammaFit[i];
    gLanes[0] = drv.mlLaneSteps;
    gLanes[1] = drv.mlLaneWork;
    gLanes[2] = drv.meSPRSteps;
    gLanes[3] = drv.meSPRs;
    gWalkDual[0] = drv.meSPRDualSent;
    gWalkDual[1] = drv.meSPRDualTaken;
    gLaneExchange[0] = drv.laneGathers;
    gLaneExchange[1] = drv.laneGatherBytes;
    drv.report();
    std::vector<std::string> nm;
    const char *p = names;
    for (int64_t k = 0; k < nAll; k++) {
        nm.push_back(std::string(p));
        p += nm.back().size() + 1;
    }
    return drv.newick(nm, std::vector<int64_t>(uniqueFirst, uniqueFirst + nSeqs), std::vector<int64_t>(alnNext, alnNext + nAll));
}

extern "C" int vft_nj_ml_newick(vft_ctx *ctx, const uint8_t *codes, int64_t nSeqs, int64_t nPos, int32_t precision,
                                const vft_nj_options *opt, int32_t meLengths, int32_t nBootstrap, const int64_t *uniqueFirst,
                                const int64_t *alnNext, int64_t nAll, const char *names, char *out, int64_t outCap,
                                int64_t *outLen, double *loglk, int32_t loglkCap, int32_t *nRounds, double *ratesOut,
                                int32_t ratesCap, int32_t *nRates, int32_t *ratecatOut, double *gtrOut, char *err, int32_t errLen) {
    if (!ctx || !codes || !uniqueFirst || !alnNext || !names || !outLen) return VFT_ERR_INVALID;
    try {
        std::vector<double> ll, rates;
        std::vector<int64_t> ratecat;
        const std::string t = precision == 8 ? runTree<double>(ctx, codes, nSeqs, nPos, opt, meLengths != 0, nBootstrap, uniqueFirst, alnNext, nAll, names, ll, rates, ratecat, gtrOut)
                                             : runTree<float>(ctx, codes, nSeqs, nPos, opt, meLengths != 0, nBootstrap, uniqueFirst, alnNext, nAll, names, ll, rates, ratecat, gtrOut);
        if (nRates) *nRates = (int32_t) rates.size();
        if (ratesOut)
            for (size_t k = 0; k < rates.size() && (int32_t) k < ratesCap; k++) ratesOut[k] = rates[k];
        if (ratecatOut)
            for (size_t k = 0; k < ratecat.size(); k++) ratecatOut[k] = (int32_t) ratecat[k];
        if (nRounds) *nRounds = (int32_t) ll.size();
        if (loglk)
            for (size_t k = 0; k < ll.size() && (int32_t) k < loglkCap; k++) loglk[k] = ll[k];
        *outLen = (int64_t) t.size();
        if (out && outCap > (int64_t) t.size()) memcpy(out, t.c_str(), t.size() + 1);
        else if (out) return VFT_ERR_INVALID;   /* outLen tells how much is needed */
        return VFT_OK;
    } catch (const std::exception &e) {
        if (err && errLen > 0) snprintf(err, (size_t) errLen, "%s", e.what());
        return VFT_ERR_STATE;
    }
}

extern "C" int vft_nj_newick(vft_ctx *ctx, const uint8_t *codes, int64_t nSeqs, int64_t nPos, int32_t precision,
                             const vft_nj_options *opt, int32_t meLengths, int32_t nBootstrap, const int64_t *uniqueFirst, const int64_t *alnNext, int64_t nAll,
                             const char *names, char *out, int64_t outCap, int64_t *outLen, char *err, int32_t errLen) {
    vft_nj_options o;
    if (opt) {
        o = *opt;
        o.mllen = 0;
    }
    return vft_nj_ml_newick(ctx, codes, nSeqs, nPos, precision, opt ? &o : nullptr, meLengths, nBootstrap, uniqueFirst, alnNext,
                            nAll, names, out, outCap, outLen, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, nullptr, err, errLen);
}

extern "C" void vft_knuth_stream(double *out, int64_t n) {
    veryfasttree::KnuthRng g;
    for (int64_t i = 0; i < n; i++) out[i] = g.rand();
}

template<typename REAL>
static int64_t runDriver(vft_ctx *ctx, const uint8_t *codes, int64_t nSeqs, int64_t nPos, const vft_nj_options *o,
                         int64_t maxJoins, int64_t *joins, double *criterion) {
    const veryfasttree::NJOptions opt = toOptions(o);
    veryfasttree::NJDriver<REAL> drv(ctx, codes, nSeqs, nPos, opt);
    const auto &js = drv.run(maxJoins);
    recordJoinCrc(js);
    drv.report();
    for (size_t k = 0; k < js.size(); k++) {
        joins[3 * k] = js[k].i;
        joins[3 * k + 1] = js[k].j;
        joins[3 * k + 2] = js[k].newnode;
        if (criterion) criterion[k] = js[k].criterion;
    }
    return (int64_t) js.size();
}

extern "C" int vft_nj_run(vft_ctx *ctx, const uint8_t *codes, int64_t nSeqs, int64_t nPos, int32_t precision,
                          const vft_nj_options *opt, int64_t maxJoins, int64_t *joins, double *criterion,
                          int64_t *nJoins, char *err, int32_t errLen) {
    if (!ctx || !codes || !joins || !nJoins) return VFT_ERR_INVALID;
    try {
        *nJoins = precision == 8 ? runDriver<double>(ctx, codes, nSeqs, nPos, opt, maxJoins, joins, criterion)
                                 : runDriver<float>(ctx, codes, nSeqs, nPos, opt, maxJoins, joins, criterion);
        return VFT_OK;
    } catch (const std::exception &e) {
        if (err && errLen > 0) snprintf(err, (size_t) errLen, "%s", e.what());
        return VFT_ERR_STATE;
    }
}

template<typename REAL>
static void runMLLengths(vft_ctx *ctx, int64_t nSeqs, int64_t nNodes, int64_t nPos, const int64_t *parent, const int64_t *child,
                         int64_t root, void *bl, int32_t recomputeFirst, int32_t rounds, double ftol, double atol,
                         int64_t nLeafGaps, double *loglk, int64_t *evals) {
    veryfasttree::MLLengths<REAL> ml(ctx, nSeqs, nNodes, parent, child, root);
    ml.setLengths((const REAL *) bl);
    if (recomputeFirst & 1) ml.recomputeMLProfiles();
    for (int32_t r = 0; r < rounds; r++) {
        if (recomputeFirst & 2) ml.optimizeRoundParallel(ftol, atol);
        else ml.optimizeRound(ftol, atol);
        if (loglk) loglk[r] = ml.treeLogLk(nPos, nLeafGaps);
    }
    ml.getLengths((REAL *) bl);
    if (evals) *evals = ml.evaluations();
}

extern "C" int vft_ml_lengths(vft_ctx *ctx, int64_t nSeqs, int64_t nNodes, int64_t nPos, int32_t precision, const int64_t *parent,
                              const int64_t *child, int64_t root, void *bl, int32_t recomputeFirst, int32_t rounds, double ftol,
                              double atol, int64_t nLeafGaps, double *loglk, int64_t *evals, char *err, int32_t errLen) {
    if (!ctx || !parent || !child || !bl || nSeqs < 3 || nNodes <= nSeqs || rounds < 0) return VFT_ERR_INVALID;
    try {
        if (precision == 8) runMLLengths<double>(ctx, nSeqs, nNodes, nPos, parent, child, root, bl, recomputeFirst, rounds, ftol, atol, nLeafGaps, loglk, evals);
        else runMLLengths<float>(ctx, nSeqs, nNodes, nPos, parent, child, root, bl, recomputeFirst, rounds, ftol, atol, nLeafGaps, loglk, evals);
        return VFT_OK;
    } catch (const std::exception &e) {
        if (err && errLen > 0) snprintf(err, (size_t) errLen, "%s", e.what());
        return VFT_ERR_STATE;
    }
}

extern "C" int vft_gtr_tables(const double *rates, const double *freq, int32_t precision, double *stat, double *statinv, double *eigenval,
                              double *codefreq, double *eigeninv, double *eigeninvT) {
    if (!rates || !freq || !stat || !statinv || !eigenval || !codefreq || !eigeninv || !eigeninvT) return VFT_ERR_INVALID;
    try {
        veryfasttree::TransitionTables4 t;
        if (precision == 8) veryfasttree::createGTR<double>(rates, freq, t);
        else veryfasttree::createGTR<float>(rates, freq, t);
        for (int i = 0; i < 4; i++) {
            stat[i] = t.stat[i];
            statinv[i] = t.statinv[i];
            eigenval[i] = t.eigenval[i];
            for (int j = 0; j < 4; j++) {
                eigeninv[4 * i + j] = t.eigeninv[i][j];
                eigeninvT[4 * i + j] = t.eigeninvT[i][j];
            }
        }
        for (int i = 0; i < 5; i++)
            for (int j = 0; j < 4; j++) codefreq[4 * i + j] = t.codeFreq[i][j];
        return VFT_OK;
    } catch (const std::exception &) {
        return VFT_ERR_INVALID;
    }
}

template<typename REAL>
static void aaModelOut(const veryfasttree::TransitionTables20 &t, double *stat, double *statinv, double *eigenval, double *codefreq,
                       double *eigeninv, double *eigeninvT) {
    for (int i = 0; i < 20; i++) {
        stat[i] = t.stat[i];
        statinv[i] = t.statinv[i];
        eigenval[i] = t.eigenval[i];
        for (int j = 0; j < 20; j++) {
            eigeninv[20 * i + j] = t.eigeninv[i][j];
            eigeninvT[20 * i + j] = t.eigeninvT[i][j];
        }
    }
    for (int i = 0; i < 21; i++)
        for (int j = 0; j < 20; j++) codefreq[20 * i + j] = t.codeFreq[i][j];
}

extern "C" int vft_aa_model_tables(int32_t model, int32_t precision, double *stat, double *statinv, double *eigenval, double *codefreq,
                                   double *eigeninv, double *eigeninvT) {
    if (!stat || !statinv || !eigenval || !codefreq || !eigeninv || !eigeninvT) return VFT_ERR_INVALID;
    try {
        veryfasttree::TransitionTables20 t;
        if (precision == 8) veryfasttree::createAAModel<double>(model, t);
        else veryfasttree::createAAModel<float>(model, t);
        aaModelOut<double>(t, stat, statinv, eigenval, codefreq, eigeninv, eigeninvT);
        return VFT_OK;
    } catch (const std::exception &) {
        return VFT_ERR_INVALID;
    }
}

static void distanceTablesOut(const veryfasttree::DistanceTables20 &d, double *distances, double *codefreq, double *eigenval,
                              double *eigentot) {
    for (int i = 0; i < 20; i++) {
        eigenval[i] = d.eigenval[i];
        eigentot[i] = d.eigentot[i];
        for (int j = 0; j < 20; j++) {
            distances[20 * i + j] = d.distances[i][j];
            codefreq[20 * i + j] = d.codeFreq[i][j];
        }
    }
}

extern "C" int vft_blosum45_tables(int32_t precision, double *distances, double *codefreq, double *eigenval, double *eigentot) {
    if (!distances || !codefreq || !eigenval || !eigentot) return VFT_ERR_INVALID;
    veryfasttree::DistanceTables20 d;
    if (precision == 8) veryfasttree::blosum45Tables<double>(d);
    else veryfasttree::blosum45Tables<float>(d);
    distanceTablesOut(d, distances, codefreq, eigenval, eigentot);
    return VFT_OK;
}

extern "C" int vft_aa_model_as_distance_tables(int32_t model, int32_t precision, double *distances, double *codefreq,
                                               double *eigenval, double *eigentot) {
    if (!distances || !codefreq || !eigenval || !eigentot) return VFT_ERR_INVALID;
    try {
        veryfasttree::TransitionTables20 t;
        veryfasttree::DistanceTables20 d;
        if (precision == 8) {
            veryfasttree::createAAModel<double>(model, t);
            veryfasttree::transitionAsDistanceTables<double>(t, d);
        } else {
            veryfasttree::createAAModel<float>(model, t);
            veryfasttree::transitionAsDistanceTables<float>(t, d);
        }
        distanceTablesOut(d, distances, codefreq, eigenval, eigentot);
        return VFT_OK;
    } catch (const std::exception &) {
        return VFT_ERR_INVALID;
    }
}
