// HipOperations<Precision> — the MI355X backend in the reference's own plug-in slot.
//
// VeryFastTree selects its vector backend with a template-template parameter:
//     template<typename Precision, template<class> class Operations> class NeighbourJoining   (src/NeighbourJoining.h:19-22)
//     Operations<Precision> operations;                                                      (src/NeighbourJoining.h:256)
// and registers one by instantiating VeyFastTreeImpl<P, XOperations> (src/impl/VeryFastTreeFloatCuda.cpp:1-7) plus an
// `-ext` branch in src/VeryFastTree.cpp:305-314.  This header is what a maintainer drops next to
// src/operations/CudaOperations.h: it satisfies the trait of src/operations/BasicOperations.h:16-39 (ALIGNMENT,
// Allocator, numeric_t and the ten per-vector methods, implemented on the host exactly like BasicOperations because a
// 4- or 20-element vector is not a GPU-sized unit of work — src/operations/CudaOperations.cu:19-27 shows what happens
// otherwise) and adds the BATCHED profile operations that the hot loops of NeighbourJoining.tcc are lifted to.
// Every batched member forwards to the C ABI of include/vft_hip.h; INTEGRATION.md lists the call sites.
//
// Host code only (C++11, like the reference).  Links against veryfasttree_amd/lib/libvft_hip.so.
#ifndef VERYFASTTREE_HIPOPERATIONS_H
#define VERYFASTTREE_HIPOPERATIONS_H

#include <cmath>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/vft_hip.h"

namespace veryfasttree {

    template<typename Precision>
    class HipOperations {
    public:
        /* same contract as BasicOperations: no alignment beyond sizeof(Precision) is needed on the host side,
           the device arena has its own layout (veryfasttree_amd/csrc/vft_layout.h) */
        static constexpr int ALIGNMENT = sizeof(Precision);
        using Allocator = std::allocator<Precision>;
        typedef Precision numeric_t;

        HipOperations() : ctx(nullptr) {}

        ~HipOperations() { if (ctx) vft_destroy(ctx); }

        HipOperations(const HipOperations &) = delete;

        HipOperations &operator=(const HipOperations &) = delete;

        /* ---- the ten primitives of the trait (BasicOperations.tcc:6-120), host side, scalar semantics.
           They remain for the call sites that are not lifted (NJ.tcc:782, 825-857, 2034-2035, ...). */
        inline void vector_multiply(numeric_t f1[], numeric_t f2[], int64_t n, numeric_t fOut[]) {
            for (int64_t i = 0; i < n; i++) fOut[i] = f1[i] * f2[i];
        }

        inline numeric_t vector_multiply_sum(numeric_t f1[], numeric_t f2[], int64_t n) {
            numeric_t out = 0.0;
            for (int64_t i = 0; i < n; i++) out += f1[i] * f2[i];
            return out;
        }

        inline numeric_t vector_multiply3_sum(numeric_t f1[], numeric_t f2[], numeric_t f3[], int64_t n) {
            numeric_t sum = 0.0;
            for (int64_t i = 0; i < n; i++) sum += f1[i] * f2[i] * f3[i];
            return sum;
        }

        inline numeric_t vector_dot_product_rot(numeric_t f1[], numeric_t f2[], numeric_t fBy[], int64_t n) {
            numeric_t out1 = 0.0, out2 = 0.0;
            for (int64_t i = 0; i < n; i++) {
                out1 += f1[i] * fBy[i];
                out2 += f2[i] * fBy[i];
            }
            return out1 * out2;
        }

        inline void vector_add(numeric_t fTot[], numeric_t fAdd[], int64_t n) {
            for (int64_t i = 0; i < n; i++) fTot[i] += fAdd[i];
        }

        inline numeric_t vector_sum(numeric_t f1[], int64_t n) {
            numeric_t out = 0.0;
            for (int64_t i = 0; i < n; i++) out += f1[i];
            return out;
        }

        inline void vector_multiply_by(numeric_t f[], numeric_t fBy, int64_t n, numeric_t fOut[]) {
            for (int64_t i = 0; i < n; i++) fOut[i] = f[i] * fBy;
        }

        inline void vector_add_mult(numeric_t fTot[], numeric_t fAdd[], numeric_t weight, int64_t n) {
            for (int64_t i = 0; i < n; i++) fTot[i] += fAdd[i] * weight;
        }

        template<int row>
        inline void matrix_by_vector4(numeric_t mat[][row], numeric_t vec[], numeric_t out[]) {
            for (int64_t j = 0; j < 4; j++) {
                double sum = 0;
                for (int64_t k = 0; k < 4; k++) sum += vec[k] * mat[k][j];
                out[j] = sum;
            }
        }

        /* exp in place with the reference's four accuracy levels (`-fastexp`, BasicOperations.tcc:122-216): 0 = libm in
           double, 1 = libm in float, 2 / 3 = the Cephes rational approximation e^x = 2^m (1 + 2 r P(r^2) / (Q(r^2) - r P(r^2)))
           evaluated in double / float with the power of two assembled from exponent bits.  The device kernels build
           their P(t) tables at level 0 (the reference's default). */
        inline void fastexp(numeric_t fTot[], int64_t n, int lvl) {
            if (lvl == 0) {
                for (int64_t k = 0; k < n; k++) fTot[k] = (numeric_t) std::exp((double) fTot[k]);
            } else if (lvl == 1) {
                for (int64_t k = 0; k < n; k++) fTot[k] = (numeric_t) std::exp((float) fTot[k]);
            } else if (lvl == 2) {
                for (int64_t k = 0; k < n; k++) {
                    int64_t m;
                    const double mant = cephesMantissa<double, int64_t>((double) fTot[k], m);
                    union { double d; int64_t i; } u;
                    u.i = (m + 1023) << 52;
                    fTot[k] = (numeric_t) mant * u.d;
                }
            } else {
                for (int64_t k = 0; k < n; k++) {
                    int32_t m;
                    const float mant = cephesMantissa<float, int32_t>((float) fTot[k], m);
                    union { float f; int32_t i; } u;
                    u.i = (m + 127) << 23;
                    fTot[k] = (numeric_t) mant * u.f;
                }
            }
        }

    private:
        /* range reduction x = m ln2 + r (ln2 split in two constants), then the rational approximation, every step in T:
           returns e^r, m by reference.  The operation order is the reference's (Horner in r^2), so results are
           bit-identical to BasicOperations::fastexp. */
        template<typename T, typename I>
        static inline T cephesMantissa(T x, I &m) {
            const T log2e = (T) 1.4426950408889634073599, ln2hi = (T) 6.93145751953125E-1, ln2lo = (T) 1.42860682030941723212E-6;
            const T P[3] = {(T) 1.26177193074810590878E-4, (T) 3.02994407707441961300E-2, (T) 9.99999999999999999910E-1};
            const T Q[4] = {(T) 3.00198505138664455042E-6, (T) 2.52448340349684104192E-3, (T) 2.27265548208155028766E-1,
                            (T) 2.00000000000000000009E0};
            const T fl = std::floor(log2e * x + (T) 0.5);
            m = (I) fl;
            x -= fl * ln2hi;
            x -= fl * ln2lo;
            const T xx = x * x;
            T px = P[0];
            for (int i = 1; i < 3; i++) {
                px *= xx;
                px += P[i];
            }
            px *= x;
            T qx = Q[0];
            for (int i = 1; i < 4; i++) {
                qx *= xx;
                qx += Q[i];
            }
            x = px / (qx - px);
            x = (T) (1.0 + 2.0 * x);   /* in double, then narrowed: the reference's literals are double */
            return x;
        }

    public:

        /* ---- batched extension: device-resident profiles.  Errors surface as std::invalid_argument so that the
           reference's main() reports them cleanly (main.cpp:673-678 only catches that type). */
        void configHip(int device, int64_t nSeqs, int64_t nPos, int nCodes, int64_t maxNodes) {
            vft_config cfg;
            cfg.device = device;
            cfg.precision = (int32_t) sizeof(Precision);
            cfg.n_codes = nCodes;
            cfg.reserved = 0;
            cfg.n_seqs = nSeqs;
            cfg.n_pos = nPos;
            cfg.max_nodes = maxNodes;
            if (ctx) vft_destroy(ctx);
            ctx = nullptr;
            int rc = vft_create(&ctx, &cfg);
            if (rc != VFT_OK) {
                std::string msg = ctx ? vft_last_error(ctx) : "vft_create failed";
                if (ctx) vft_destroy(ctx);
                ctx = nullptr;
                throw std::invalid_argument("HipOperations: " + msg);
            }
        }

        bool ready() const { return ctx != nullptr; }

        /* seqsToProfiles (NJ.tcc:382-457): codes[nSeqs][nPos], NOCODE = 127 */
        void uploadLeaves(const uint8_t *codes) { chk(vft_upload_leaves(ctx, codes)); }

        void setDistanceMatrix(const numeric_t *distances, const numeric_t *codeFreq, const numeric_t *eigenval,
                               const numeric_t *eigentot) {
            chk(vft_set_distance_matrix(ctx, distances, codeFreq, eigenval, eigentot));
        }

        void setTransitionMatrix(const numeric_t *stat, const numeric_t *statinv, const numeric_t *eigenval,
                                 const numeric_t *codeFreq, const numeric_t *eigeninv, const numeric_t *eigeninvT) {
            chk(vft_set_transition_matrix(ctx, stat, statinv, eigenval, codeFreq, eigeninv, eigeninvT));
        }

        void setRates(const numeric_t *rates, int32_t nRates, const int64_t *ratecat) {
            chk(vft_set_rates(ctx, rates, nRates, ratecat));
        }

        void setParents(int64_t first, int64_t count, const int64_t *parent) {
            chk(vft_set_parents(ctx, first, count, parent));
        }

        void setNodeScalars(int64_t first, int64_t count, const numeric_t *diameter, const numeric_t *selfweight,
                            const numeric_t *selfdist) {
            chk(vft_set_node_scalars(ctx, first, count, diameter, selfweight, selfdist));
        }

        void setMaxNode(int64_t maxnode) { chk(vft_set_max_node(ctx, maxnode)); }

        /* profileDist / seqDist over a pair list (NJ.tcc:1167-1190): raw distance and weight */
        void profileDist(int64_t n, const int64_t *i, const int64_t *j, numeric_t *dist, numeric_t *weight) {
            chk(vft_profile_distances(ctx, n, i, j, dist, weight));
        }

        /* splitSupport (NJ.tcc:607-702) for n quartets over nBoot column resamples (col: [nBoot][nPos]) */
        void splitSupports(int64_t n, const int64_t *a, const int64_t *b, const int64_t *c, const int64_t *d, int32_t nBoot,
                           const int32_t *col, double *support) {
            chk(vft_split_supports(ctx, n, a, b, c, d, nBoot, col, support));
        }

        /* the state change of one join (NJ.tcc:2904-2909, 3003-3007, 254) in one launch */
        void joinNodes(int64_t i, int64_t j, int64_t newnode, numeric_t diameter, int64_t staleStamp) {
            chk(vft_join_nodes(ctx, i, j, newnode, (double) diameter, staleStamp));
        }

        /* averageProfile + the new node's self distance (NJ.tcc:3008, 3039-3042) */
        /* joinNodes + averageProfiles(new, i, j) + its self-distance + updateOutProfile in one launch (vft_join_fused) */
        void joinFused(int64_t i, int64_t j, int64_t newnode, numeric_t diameter, int64_t staleStamp, int64_t nActiveOld,
                       bool updateOutProfile) {
            chk(vft_join_fused(ctx, i, j, newnode, (double) diameter, staleStamp, nActiveOld, updateOutProfile ? 1 : 0));
        }

        /* TopHits on the device (NJ.h:206-248): the lists, and the three list walks of a join as one launch each */
        void topHitsCreate(int32_t m, int64_t nLists) { chk(vft_tophits_create(ctx, m, nLists)); }

        void topHitsUpload(int64_t count, const int64_t *nodes, const int32_t *lens, const void *packed) {
            chk(vft_tophits_upload(ctx, count, nodes, lens, packed));
        }

        /* getBestFromTopHits (NJ.tcc:4267-4298) */
        void topHitsBest(int64_t node, int32_t len, int64_t nActive, int64_t nDiffAllow, double totdiam, bool forceNode, vft_tophits_best_t &out) {
            chk(vft_tophits_best(ctx, node, len, nActive, nDiffAllow, totdiam, forceNode ? 1 : 0, &out));
        }

        /* topHitJoin's merge (NJ.tcc:4319-4362, 4786-4833, 4535-4578) */
        void topHitsJoin(int64_t newnode, int64_t c0, int32_t n0, int64_t c1, int32_t n1, int64_t nActive, int64_t nDiffAllow, double totdiam,
                         int32_t nSaveMax, int32_t need, bool ageOk, vft_tophits_join_t &info, int32_t *j, numeric_t *dist, numeric_t *criterion) {
            chk(vft_tophits_join(ctx, newnode, c0, n0, c1, n1, nActive, nDiffAllow, totdiam, nSaveMax, need, ageOk ? 1 : 0, &info, j, dist, criterion));
        }

        /* topHitJoin's refresh (NJ.tcc:4440-4517) after the sweep of the new node */
        void topHitsRefresh(int64_t newnode, int32_t nHits, const int64_t *hitJ, const numeric_t *hitDist, int32_t nOwn, const void *ownList,
                            int64_t nWork, const int64_t *work, const int32_t *nNew, int64_t nActive, int64_t nDiffAllow, double totdiam,
                            int32_t *lens, void *first) {
            chk(vft_tophits_refresh(ctx, newnode, nHits, hitJ, hitDist, nOwn, ownList, nWork, work, nNew, nActive, nDiffAllow, totdiam, lens, first));
        }

        /* the join loop itself on the device (NJ.tcc:2857-3047; vft_nj_engine_*, driven by NJDriver::runEngine) */
        void njEngineCreate(const vft_nj_engine_config &cfg) { chk(vft_nj_engine_create(ctx, &cfg)); }

        void njEngineEnqueue(int64_t joinIndex, int32_t phases, bool updateOutProfile) {
            chk(vft_nj_engine_enqueue(ctx, joinIndex, phases, updateOutProfile ? 1 : 0));
        }

        void njEnginePoll(int64_t &joinsDone, int32_t &halt, int32_t &haltJoin) { chk(vft_nj_engine_poll(ctx, &joinsDone, &halt, &haltJoin)); }

        void averageProfiles(int64_t n, const int64_t *out, const int64_t *a, const int64_t *b, const double *bionjWeight) {
            chk(vft_average_profiles(ctx, n, out, a, b, bionjWeight));
        }

        /* tree-refinement phase: averages into plain rows, chains of dependent averages in one launch */
        void setProfileRows(bool on) { chk(vft_set_profile_rows(ctx, on ? 1 : 0)); }

        void averageChain(int32_t n, const int64_t *out, const int64_t *a, const int64_t *b) { chk(vft_average_chain(ctx, n, out, a, b)); }

        /* branchlength[] on the device and the ML operations that read it (optimizeAllBranchLengths NJ.tcc:5006-5113,
           recomputeProfile :3436, getUpProfile :3382) */
        void setBranchLengths(int64_t first, int64_t count, const numeric_t *v) { chk(vft_branch_lengths_set(ctx, first, count, v)); }

        void getBranchLengths(int64_t first, int64_t count, numeric_t *v) { chk(vft_branch_lengths_get(ctx, first, count, v)); }

        void posteriorProfilesBlen(int64_t n, const int64_t *out, const int64_t *a, const int64_t *b, const int64_t *lenIdxA,
                                   const int64_t *lenIdxB) {
            chk(vft_posterior_profiles_blen(ctx, n, out, a, b, lenIdxA, lenIdxB));
        }

        void posteriorChainBlen(int32_t n, const int64_t *out, const int64_t *a, const int64_t *b, const int64_t *lenIdxA,
                                const int64_t *lenIdxB) {
            chk(vft_posterior_chain_blen(ctx, n, out, a, b, lenIdxA, lenIdxB));
        }

        /* the loop body of optimizeAllBranchLengths for n independent splits (NJ.tcc:5025-5064) */
        void mlOptimizeSplits(int64_t n, const int64_t *ids3, const int64_t *lenIdx3, const int64_t *recompute, double ftol, double atol) {
            chk(vft_ml_optimize_splits(ctx, n, ids3, lenIdx3, recompute, ftol, atol));
        }

        /* testSplitsML + SHSupport for n splits (NJ.tcc:6856-6999, 1126-1165) */
        void mlSplitTests(int64_t n, const int64_t *ids4, const int64_t *lenIdx5, double ftol, double atol, double closeLimit,
                          bool alwaysSecondPass, double *loglk3, int32_t nBoot, const int32_t *col, double *support, double *lengths) {
            chk(vft_ml_split_tests(ctx, n, ids4, lenIdx5, ftol, atol, closeLimit, alwaysSecondPass ? 1 : 0, loglk3, nBoot, col, support, lengths));
        }

        /* MLQuartetNNI (NJ.tcc:4885-5004) */
        void mlQuartetNNI(int64_t n, const int64_t *ids4, const int64_t *lenIdx5, double ftol, double atol, double closeLimit,
                          int32_t mlAccuracy, vft_quartet_nni *results) {
            chk(vft_ml_quartet_nni(ctx, n, ids4, lenIdx5, ftol, atol, closeLimit, mlAccuracy, results));
        }

        /* outProfile / updateOutProfile (NJ.tcc:3012-3036) */
        void outProfile(int64_t nActive, const int64_t *activeIds) { chk(vft_out_profile_full(ctx, nActive, activeIds)); }

        void updateOutProfile(int64_t old1, int64_t old2, int64_t newNode, int64_t nActiveOld) {
            chk(vft_out_profile_update(ctx, old1, old2, newNode, nActiveOld));
        }

        /* setOutDistance over a list (NJ.tcc:257-260, 2897-2898, 4451-4464); ids == nullptr: every active node */
        void setOutDistances(int64_t n, const int64_t *ids, int64_t nActive, double totdiam) {
            chk(vft_out_distances(ctx, n, ids, nActive, totdiam));
        }

        /* setBestHit + psort + the first k hits (NJ.tcc:3801-3811, 3927-3930, 4470-4471) */
        template<typename HitRecord>
        void setBestHit(int64_t node, int64_t nActive, int64_t nDiffAllow, double totdiam, int32_t k, HitRecord *hits,
                        int64_t *bestJ) {
            static_assert(sizeof(HitRecord) == (sizeof(Precision) == 4 ? sizeof(vft_hit_f32) : sizeof(vft_hit_f64)),
                          "hit record must be vft_hit_f32 / vft_hit_f64");
            chk(vft_sweep(ctx, node, nActive, nDiffAllow, totdiam, k, hits, nullptr, bestJ));
        }

        /* setDistCriterion over a pair list: transferBestHits / uniqueBestHits / getBestFromTopHits
           (NJ.tcc:4580-4613, 4786-4833, 4267-4298) */
        void setDistCriterion(int64_t n, const int64_t *i, const int64_t *j, int64_t nActive, int64_t nDiffAllow,
                              double totdiam, numeric_t *dist, numeric_t *weight, numeric_t *criterion) {
            chk(vft_pair_distances(ctx, n, i, j, nActive, nDiffAllow, totdiam, dist, weight, criterion));
        }

        /* pairLogLk for one level of treeLogLk (NJ.tcc:5123) or one Brent evaluation (NJ.tcc:1449-1458) */
        void pairLogLk(int64_t n, const int64_t *a, const int64_t *b, const double *length, double *loglk,
                       double *siteLikelihoods) {
            chk(vft_pair_loglk(ctx, n, a, b, length, loglk, siteLikelihoods));
        }

        /* posteriorProfile for one level of recomputeMLProfiles (NJ.tcc:3516-3539) */
        void posteriorProfiles(int64_t n, const int64_t *out, const int64_t *a, const int64_t *b, const double *len1,
                               const double *len2) {
            chk(vft_posterior_profiles(ctx, n, out, a, b, len1, len2));
        }

        vft_ctx *context() { return ctx; }

    private:
        vft_ctx *ctx;

        void chk(int rc) {
            if (rc != VFT_OK) throw std::invalid_argument(std::string("HipOperations: ") + vft_last_error(ctx));
        }
    };
}

#endif
