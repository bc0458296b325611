// Everything the pipeline does to a finished tree, on the host, over the C ABI of include/vft_hip.h (plain C++11, no
// HIP).  It started as the maximum-likelihood branch lengths on a fixed topology - recomputeMLProfiles (NJ.tcc:3516-3539),
// optimizeAllBranchLengths (NJ.tcc:5006-5113), treeLogLk (NJ.tcc:5114-5259) - and now also holds setMLRates
// (:5429-5488), setMLGtr (:6436-6500), testSplitsML (:6800-6999), DoNNI with either criterion (:5797-6200) and SPR
// (:6185-6404): one class owns parent[] / child[][3] and the up-profile cache, the device owns profiles and lengths.
//
// The device keeps the profiles, the up-profiles (node X -> id X + nSeqs, as in NJDriver.h) and branchlength[]; the host
// only decides WHAT runs in which order, which depends on the topology alone: the post-order walk, and the moment each
// up-profile is (re)built - getUpProfile caches them and optimizeAllBranchLengths drops a node's entry once the node is
// done (NJ.tcc:5061-5062), so an ancestor's up-profile dates from the first visit to its subtree.  Everything is queued
// on the context's stream; a round is ~5 launches per internal node and no host synchronisation.
#ifndef VFT_ML_LENGTHS_H
#define VFT_ML_LENGTHS_H

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/vft_hip.h"
#include "../../include/vft_host.h"
#include "GtrModel.h"

namespace veryfasttree {

    template<typename REAL>
    class MLLengths {
    public:
        /* child: [nNodes][3] (-1 = none), parent: [nNodes] (-1 at the root); the context must have max_nodes >=
           nNodes + nSeqs and the model (rates, transition matrix, ML limits) already set */
        MLLengths(vft_ctx *ctx, int64_t nSeqs, int64_t nNodes, const int64_t *parentIn, const int64_t *childIn, int64_t root)
                : ctx(ctx), nSeqs(nSeqs), nNodes(nNodes), root(root), parent(parentIn, parentIn + nNodes),
                  child(childIn, childIn + 3 * nNodes) {
            if (root < nSeqs || root >= nNodes || child[3 * root + 2] < 0) throw std::invalid_argument("MLLengths: the root must have three children");
            chk(vft_set_max_node(ctx, nNodes + nSeqs));
            /* the post-order walk (traversePostorder, NJ.tcc:3343-3380): children in stored order, then the node */
            std::vector<std::pair<int64_t, int>> stack(1, std::make_pair(root, 0));
            while (!stack.empty()) {
                const int64_t v = stack.back().first;
                const int k = stack.back().second;
                if (k < 3 && child[3 * v + k] >= 0) {
                    stack.back().second++;
                    stack.push_back(std::make_pair(child[3 * v + k], 0));
                } else {
                    stack.pop_back();
                    if (child[3 * v] >= 0) order.push_back(v);
                }
            }
        }

        void setLengths(const REAL *bl) { chk(vft_branch_lengths_set(ctx, 0, nNodes, bl)); }

        void getLengths(REAL *bl) { chk(vft_branch_lengths_get(ctx, 0, nNodes, bl)); }

        /* recomputeMLProfiles: every internal profile from its children, one device call per tree level */
        void recomputeMLProfiles() {
            std::vector<int64_t> level((size_t) nNodes, 0);
            int64_t top = 0;
            for (int64_t v: order) {
                if (v == root) continue;
                const int64_t a = child[3 * v], b = child[3 * v + 1];
                level[(size_t) v] = 1 + std::max(level[(size_t) a], level[(size_t) b]);
                top = std::max(top, level[(size_t) v]);
            }
            std::vector<std::vector<int64_t>> byLevel((size_t) top + 1);
            for (int64_t v: order)
                if (v != root) byLevel[(size_t) level[(size_t) v]].push_back(v);
            for (int64_t lv = 1; lv <= top; lv++) {
                const std::vector<int64_t> &out = byLevel[(size_t) lv];
                std::vector<int64_t> a, b;
                for (int64_t v: out) {
                    a.push_back(child[3 * v]);
                    b.push_back(child[3 * v + 1]);
                }
                chk(vft_posterior_profiles_blen(ctx, (int64_t) out.size(), out.data(), a.data(), b.data(), a.data(), b.data()));
            }
        }

        /* recomputeProfiles (NJ.tcc:3475-3500) without a matrix: every internal profile as the plain average of its
           children, one vft_average_profiles per tree level (through the tile streams: clears the dense ML rows) */
        void recomputeAverageProfiles() {
            std::vector<int64_t> level((size_t) nNodes, 0);
            int64_t top = 0;
            for (int64_t v: order) {
                if (v == root) continue;
                level[(size_t) v] = 1 + std::max(level[(size_t) child[3 * v]], level[(size_t) child[3 * v + 1]]);
                top = std::max(top, level[(size_t) v]);
            }
            std::vector<std::vector<int64_t>> byLevel((size_t) top + 1);
            for (int64_t v: order)
                if (v != root) byLevel[(size_t) level[(size_t) v]].push_back(v);
            for (int64_t lv = 1; lv <= top; lv++) {
                const std::vector<int64_t> &out = byLevel[(size_t) lv];
                std::vector<int64_t> a, b;
                for (int64_t v: out) {
                    a.push_back(child[3 * v]);
                    b.push_back(child[3 * v + 1]);
                }
                chk(vft_average_profiles(ctx, (int64_t) out.size(), out.data(), a.data(), b.data(), nullptr));
            }
        }

        /* one call of optimizeAllBranchLengths */
        void optimizeRound(double ftol, double atol) {
            std::vector<char> upHave((size_t) nNodes, 0), done((size_t) nNodes, 0);
            optimizeRoundFrom(ftol, atol, upHave, done);
        }

        /* the walk itself; done[v]: handled by a lane of optimizeRoundThreaded, upHave as the lanes left it */
        void optimizeRoundFrom(double ftol, double atol, std::vector<char> &upHave, const std::vector<char> &done) {
            std::vector<int64_t> path;
            for (int64_t v: order) {
                if (done[(size_t) v]) continue;
                int64_t ids[3], li[3], rec;
                if (v == root) {
                    for (int k = 0; k < 3; k++) ids[k] = li[k] = child[3 * v + k];
                    rec = -1;
                } else {
                    /* getUpProfile(v): missing up-profiles on the way from the root down to v */
                    path.clear();
                    for (int64_t x = v; x != root; x = parent[(size_t) x]) path.push_back(x);
                    for (size_t t = path.size(); t-- > 0;) {
                        const int64_t x = path[t];
                        if (upHave[(size_t) x]) continue;
                        int64_t c, d, lc, ld;
                        const int64_t p = parent[(size_t) x];
                        if (p == root) {   /* the two other children of the root */
                            int64_t sibs[2];
                            int n = 0;
                            for (int k = 0; k < 3; k++)
                                if (child[3 * root + k] != x) sibs[n++] = child[3 * root + k];
                            c = lc = sibs[0];
                            d = ld = sibs[1];
                        } else {           /* the sibling and the parent's up-profile with the parent's branch */
                            c = lc = child[3 * p] == x ? child[3 * p + 1] : child[3 * p];
                            d = p + nSeqs;
                            ld = p;
                        }
                        const int64_t out = x + nSeqs;
                        chk(vft_posterior_profiles_blen(ctx, 1, &out, &c, &d, &lc, &ld));
                        upHave[(size_t) x] = 1;
                    }
                    ids[0] = li[0] = child[3 * v];
                    ids[1] = li[1] = child[3 * v + 1];
                    ids[2] = v + nSeqs;
                    li[2] = v;
                    rec = v;
                }
                chk(vft_ml_optimize_splits(ctx, 1, ids, li, &rec, ftol, atol));
                upHave[(size_t) v] = 0;   /* NJ.tcc:5062 */
            }
        }

        /* A level-parallel variant of optimizeAllBranchLengths for callers that do not need the one-thread order: all
           up-profiles are built first (from the lengths as they are), then the splits of one tree height go down as ONE
           vft_ml_optimize_splits batch, bottom-up, the root last.  Splits of the same height own disjoint branches and
           only read finished children, so a batch is race-free; what differs from the sequential walk is that the
           up-profiles do not see the updates made earlier in the same round (a Jacobi instead of a Gauss-Seidel sweep -
           the kind of difference the reference's own -threads-level 3 mode has against one thread).  A round is
           O(tree height) launches instead of O(nodes). */
        void optimizeRoundParallel(double ftol, double atol) {
            allUpProfiles();
            std::vector<int64_t> height((size_t) nNodes, 0);
            int64_t top = 0;
            for (int64_t v: order) {
                if (v == root) continue;
                height[(size_t) v] = 1 + std::max(height[(size_t) child[3 * v]], height[(size_t) child[3 * v + 1]]);
                top = std::max(top, height[(size_t) v]);
            }
            std::vector<std::vector<int64_t>> byHeight((size_t) top + 1);
            for (int64_t v: order)
                if (v != root) byHeight[(size_t) height[(size_t) v]].push_back(v);
            const size_t maxBatch = 4096;
            for (int64_t h = 1; h <= top; h++) {
                const std::vector<int64_t> &lv = byHeight[(size_t) h];
                for (size_t k0 = 0; k0 < lv.size(); k0 += maxBatch) {
                    const size_t cnt = std::min(maxBatch, lv.size() - k0);
                    std::vector<int64_t> ids, li, rec;
                    for (size_t k = k0; k < k0 + cnt; k++) {
                        const int64_t v = lv[k];
                        const int64_t q[3] = {child[3 * v], child[3 * v + 1], v + nSeqs}, l[3] = {child[3 * v], child[3 * v + 1], v};
                        ids.insert(ids.end(), q, q + 3);
                        li.insert(li.end(), l, l + 3);
                        rec.push_back(v);
                    }
                    chk(vft_ml_optimize_splits(ctx, (int64_t) cnt, ids.data(), li.data(), rec.data(), ftol, atol));
                }
            }
            int64_t ids[3], rec = -1;
            for (int k = 0; k < 3; k++) ids[k] = child[3 * root + k];
            chk(vft_ml_optimize_splits(ctx, 1, ids, ids, &rec, ftol, atol));
        }

        /* treeLogLk without site likelihoods; nLeafGaps >= 0: the Jukes-Cantor correction (NJ.tcc:5236-5256) with that
           many gap characters in the nSeqs x nPos leaves, < 0: a matrix model, no correction */
        double treeLogLk(int64_t nPos, int64_t nLeafGaps) {
            std::vector<REAL> bl((size_t) nNodes);
            getLengths(bl.data());
            /* one pairLogLk per internal node, summed in the reference's post-order (traversePostorder, the root last):
               the total is then the reference's double sum, addend for addend */
            std::vector<int64_t> a, b;
            std::vector<double> len;
            for (int64_t v: order) {
                a.push_back(child[3 * v]);
                b.push_back(child[3 * v + 1]);
                const REAL sum = bl[(size_t) a.back()] + bl[(size_t) b.back()];   /* numeric_t sum, NJ.tcc:5124 */
                len.push_back((double) sum);
            }
            std::vector<double> ll(a.size());
            if (sharded()) {
                /* SURVEY 8e "ML level batches", the treeLogLk part: the nodes' pair likelihoods are independent reads of the
                   (replicated) profiles - rank r evaluates its share of the post-order list, one all-gather of 8 bytes per node,
                   and every rank adds the same doubles in the same order below.  (recomputeMLProfiles stays replicated: sharing
                   a level's posteriors means all-gathering one row per node where recomputing it moves three rows through the
                   local HBM - the arithmetic is in DESIGN.md section 5.) */
                size_t per, k0, k1;
                shareOf(a.size(), per, k0, k1);
                if (k1 > k0) chk(vft_pair_loglk(ctx, (int64_t) (k1 - k0), a.data() + k0, b.data() + k0, len.data() + k0, ll.data() + k0, nullptr));
                if ((int64_t) (per * sizeof(double)) > comm->h_cap) throw std::invalid_argument("MLLengths: vft_comm host buffers too small for a tree's pair likelihoods");
                std::memcpy(comm->h_send, ll.data() + k0, (k1 - k0) * sizeof(double));
                const char *rcv = gatherRecords(per, sizeof(double));
                for (size_t t = 0; t < a.size(); t++) std::memcpy(&ll[t], rcv + laneRecord(t, per) * sizeof(double), sizeof(double));
                treeLogLkSharded++;
            } else
            chk(vft_pair_loglk(ctx, (int64_t) a.size(), a.data(), b.data(), len.data(), ll.data(), nullptr));
            /* the third branch of the root against the posterior of the first two (NJ.tcc:5138-5151); the root's own
               slot holds that temporary */
            const int64_t r0 = child[3 * root], r1 = child[3 * root + 1], r2 = child[3 * root + 2];
            const double l2 = (double) bl[(size_t) r2];
            chk(vft_posterior_profiles_blen(ctx, 1, &root, &r0, &r1, &r0, &r1));   /* lengths from the device: the same values */
            double ll3 = 0;
            chk(vft_pair_loglk(ctx, 1, &root, &r2, &l2, &ll3, nullptr));
            double total = 0;
            for (size_t t = 0; t < order.size(); t++) {
                double atNode = 0;
                atNode += ll[t];
                if (order[t] == root) atNode += ll3;   /* traverseTreeLogLk adds both of the root's terms before returning */
                total += atNode;
            }
            if (nLeafGaps >= 0) {   /* NJ.tcc:5236-5256 */
                const double logNCodes = std::log(4.0);
                total -= (double) nPos * logNCodes;
                total += (double) nLeafGaps * logNCodes;
            }
            return total;
        }

        /* treeLogLk's per-site part (NJ.tcc:5116-5134, :5172-5227): the product over all splits of the per-site pair
           likelihoods, kept in range the reference's way, as a log.  Without the per-site Jukes-Cantor gap correction
           (NJ.tcc:5247-5252), which is the same for every rate and so cannot change setMLRates' choice.
           The pair likelihoods come back from the device split by split (vft_pair_loglk with site_lk); the running
           products are host work for now. */
        void siteLogLk(int64_t nPos, double *siteLoglk) {
            std::vector<REAL> bl((size_t) nNodes);
            getLengths(bl.data());
            std::vector<int64_t> a, b;
            std::vector<double> len;
            for (int64_t v: order) {   /* post-order, the root last */
                a.push_back(child[3 * v]);
                b.push_back(child[3 * v + 1]);
                const REAL sum = bl[(size_t) a.back()] + bl[(size_t) b.back()];
                len.push_back((double) sum);
            }
            std::vector<double> lik((size_t) nPos, 1.0);
            for (int64_t p = 0; p < nPos; p++) siteLoglk[p] = 0.0;
            const double under = 1.0e-4, underInv = 1.0e4, logUnder = 9.21034037197618;   /* Constants.h:13-15 */
            auto keepInRange = [&]() {
                for (int64_t p = 0; p < nPos; p++)
                    while (lik[(size_t) p] < under) {
                        lik[(size_t) p] *= underInv;
                        siteLoglk[p] -= logUnder;
                    }
            };
            const int64_t n = (int64_t) a.size();
            const int64_t chunk = std::max<int64_t>(1, (int64_t) ((256u << 20) / (8 * (size_t) nPos)));
            std::vector<double> ll, site;
            for (int64_t k0 = 0; k0 < n; k0 += chunk) {
                const int64_t cnt = std::min(chunk, n - k0);
                ll.resize((size_t) cnt);
                site.resize((size_t) (cnt * nPos));
                chk(vft_pair_loglk(ctx, cnt, a.data() + k0, b.data() + k0, len.data() + k0, ll.data(), site.data()));
                for (int64_t k = 0; k < cnt; k++) {
                    const double *row = site.data() + (size_t) (k * nPos);
                    for (int64_t p = 0; p < nPos; p++) lik[(size_t) p] *= row[p];
                    keepInRange();
                }
            }
            /* the root's third branch (NJ.tcc:5138-5151): multiplied in after the range check of the root's pair */
            const int64_t r0 = child[3 * root], r1 = child[3 * root + 1], r2 = child[3 * root + 2];
            const double l2 = (double) bl[(size_t) r2];
            chk(vft_posterior_profiles_blen(ctx, 1, &root, &r0, &r1, &r0, &r1));   /* lengths from the device: the same values */
            double ll3 = 0;
            site.resize((size_t) nPos);
            chk(vft_pair_loglk(ctx, 1, &root, &r2, &l2, &ll3, site.data()));
            for (int64_t p = 0; p < nPos; p++) siteLoglk[p] += std::log(lik[(size_t) p] * site[(size_t) p]);
        }

        /* setMLRates (NJ.tcc:5429-5488) for nCat > 1: the CAT approximation.  MLSiteRates (:5367-5378) spreads nCat
           rates evenly in log space over [1/nCat, nCat]; MLSiteLikelihoodsByRate (:5381-5410) evaluates the tree's
           site likelihoods with every site at each rate in turn; every site takes the rate with the best likelihood x
           Gamma(3, 1/3) prior; the rates are rescaled to mean 1 over the sites; the profiles are rebuilt.
           rates / ratecat receive the result (what the reference logs as "Rates" / "SiteCategories"). */
        void setMLRates(int32_t nCat, int64_t nPos, std::vector<REAL> &rates, std::vector<int64_t> &ratecat) {
            ratecat.assign((size_t) nPos, 0);
            if (nCat <= 1) {
                rates.assign(1, (REAL) 1);
                chk(vft_set_rates(ctx, rates.data(), 1, ratecat.data()));
                recomputeMLProfiles();
                return;
            }
            rates.resize((size_t) nCat);
            const double logNCat = std::log((double) nCat);
            const double logd = (logNCat + logNCat) / (double) (nCat - 1);
            for (int32_t i = 0; i < nCat; i++) rates[(size_t) i] = (REAL) std::exp(-logNCat + logd * (double) i);
            std::vector<double> siteLoglk((size_t) (nPos * nCat));
            for (int32_t i = 0; i < nCat; i++) {
                chk(vft_set_rates(ctx, &rates[(size_t) i], 1, ratecat.data()));
                recomputeMLProfiles();
                siteLogLk(nPos, siteLoglk.data() + (size_t) (nPos * i));
            }
            double sumRates = 0;
            for (int64_t p = 0; p < nPos; p++) {
                int64_t best = -1;
                double dBest = -1e20;
                for (int32_t i = 0; i < nCat; i++) {
                    const double r = (double) rates[(size_t) i];
                    const double withPrior = siteLoglk[(size_t) (nPos * i + p)] + 2.0 * std::log(r) - 3.0 * r;
                    if (withPrior > dBest) {
                        best = i;
                        dBest = withPrior;
                    }
                }
                ratecat[(size_t) p] = best;
                sumRates += (double) rates[(size_t) best];
            }
            const double avgRate = sumRates / (double) nPos;
            for (int32_t i = 0; i < nCat; i++) rates[(size_t) i] = (REAL) ((double) rates[(size_t) i] / avgRate);
            chk(vft_set_rates(ctx, rates.data(), nCat, ratecat.data()));
            recomputeMLProfiles();
        }

        /* `-gamma` (branchlengthScale, NJ.tcc:297-308; rescaleGammaLogLk :5295-5357; gammaLogLk :5261-5293): after the CAT tree is
           final, the site likelihoods at each of the nCat evenly spaced rates once more (MLSiteLikelihoodsByRate, with treeLogLk's
           per-site Jukes-Cantor gap terms this time: gapsPerPos, or null with a transition matrix), then - host arithmetic on
           those nCat x nPos numbers - the shape alpha of a discretised Gamma and a multiplier of the rates are fitted by
           alternating line searches (at most 10 rounds), and every branch length is multiplied by 1 / multiplier.  Returns
           {Gamma(nCat) log-likelihood, alpha, 1 / multiplier} - the numbers of the reference's "Gamma(20) LogLk" line. */
        struct GammaFit {
            double loglk, alpha, rescale;
        };

        static double lnGamma(double alpha) {   /* NJ.tcc:7192-7210 (Pike & Hill 1966, Algorithm 291) */
            double x = alpha, f = 0;
            if (x < 7) {
                f = 1;
                double z = x - 1;
                while (++z < 7) f *= z;
                x = z;
                f = -std::log(f);
            }
            const double z = 1 / (x * x);
            return f + (x - 0.5) * std::log(x) - x + .918938533204673
                   + (((-.000595238095238 * z + .000793650793651) * z - .002777777777778) * z + .083333333333333) / x;
        }

        static double incompleteGamma(double x, double alpha, double lnGammaAlpha) {   /* NJ.tcc:7212-7275 (Bhattacharjee 1970, AS 32) */
            const double accurate = 1e-8, overflow = 1e30;
            const double p = alpha;
            if (x == 0) return 0;
            if (x < 0 || p <= 0) return -1;
            const double factor = std::exp(p * std::log(x) - x - lnGammaAlpha);
            if (!(x > 1 && x >= p)) {   /* series expansion */
                double gin = 1, term = 1, rn = p;
                do {
                    rn++;
                    term *= x / rn;
                    gin += term;
                } while (term > accurate);
                return gin * (factor / p);
            }
            /* continued fraction */
            double a = 1 - p, b = a + x + 1, term = 0, pn[6];
            pn[0] = 1;
            pn[1] = x;
            pn[2] = x + 1;
            pn[3] = x * b;
            double gin = pn[2] / pn[3];
            for (;;) {
                a++;
                b += 2;
                term++;
                const double an = a * term;
                for (int i = 0; i < 2; i++) pn[i + 4] = b * pn[i + 2] - an * pn[i];
                if (pn[5] != 0) {
                    const double rn = pn[4] / pn[5], dif = std::fabs(gin - rn);
                    if (dif <= accurate && dif <= accurate * rn) break;
                    gin = rn;
                }
                for (int i = 0; i < 4; i++) pn[i] = pn[i + 2];
                if (!(std::fabs(pn[4]) < overflow))
                    for (int i = 0; i < 4; i++) pn[i] /= overflow;
            }
            return 1 - factor * gin;
        }

        static double pGamma(double x, double alpha) { return incompleteGamma(x * alpha, alpha, lnGamma(alpha)); }   /* NJ.tcc:5362-5365 */

        GammaFit branchlengthScale(int32_t nCat, int64_t nPos, const std::vector<REAL> &curRates, const std::vector<int64_t> &curRatecat,
                                   const int64_t *gapsPerPos) {
            std::vector<REAL> rates((size_t) nCat);
            const double logNCat = std::log((double) nCat);
            const double logd = (logNCat + logNCat) / (double) (nCat - 1);
            for (int32_t i = 0; i < nCat; i++) rates[(size_t) i] = (REAL) std::exp(-logNCat + logd * (double) i);
            std::vector<double> siteLoglk((size_t) (nPos * nCat));
            std::vector<int64_t> zero((size_t) nPos, 0);
            const double logNCodes = std::log(4.0);
            for (int32_t i = 0; i < nCat; i++) {
                chk(vft_set_rates(ctx, &rates[(size_t) i], 1, zero.data()));
                recomputeMLProfiles();
                double *row = siteLoglk.data() + (size_t) (nPos * i);
                siteLogLk(nPos, row);
                if (gapsPerPos)   /* NJ.tcc:5236-5252 */
                    for (int64_t p = 0; p < nPos; p++) {
                        row[p] += (double) gapsPerPos[p] * logNCodes;
                        row[p] -= logNCodes;
                    }
            }
            chk(vft_set_rates(ctx, curRates.data(), (int32_t) curRates.size(), curRatecat.data()));   /* "restore original rates and profiles" */
            recomputeMLProfiles();
            double mult = 1.0, alpha = 1.0;
            std::vector<double> dRate((size_t) nCat);
            auto gammaLogLk = [&](double *sites) {
                for (int32_t i = 0; i < nCat; i++) {
                    const double pMin = i == 0 ? 0.0 : pGamma(mult * (double) (rates[(size_t) i - 1] + rates[(size_t) i]) / 2.0, alpha);
                    const double pMax = i == nCat - 1 ? 1.0 : pGamma(mult * (double) (rates[(size_t) i] + rates[(size_t) i + 1]) / 2.0, alpha);
                    dRate[(size_t) i] = pMax - pMin;
                }
                double loglk = 0.0;
                for (int64_t p = 0; p < nPos; p++) {
                    double maxloglk = -1e20;
                    for (int32_t i = 0; i < nCat; i++) maxloglk = std::max(maxloglk, siteLoglk[(size_t) (nPos * i + p)]);
                    double rellk = 0;
                    for (int32_t i = 0; i < nCat; i++) rellk += std::exp(siteLoglk[(size_t) (nPos * i + p)] - maxloglk) * dRate[(size_t) i];
                    const double s = maxloglk + std::log(rellk);
                    loglk += s;
                    if (sites) sites[p] = s;
                }
                return loglk;
            };
            double fx = -gammaLogLk(nullptr);
            for (int round = 0; round < 10; round++) {
                const double start = fx;
                auto optAlpha = [&](double x) {
                    alpha = x;
                    return -gammaLogLk(nullptr);
                };
                alpha = hostMinimise(optAlpha, 0.01, alpha, 10.0, 0.001, 0.001);
                auto optMult = [&](double x) {
                    mult = x;
                    return -gammaLogLk(nullptr);
                };
                mult = hostMinimise(optMult, 0.01, mult, 10.0, 0.001, 0.001);
                fx = -gammaLogLk(nullptr);
                if (fx > start - 0.001) break;
            }
            GammaFit g;
            g.loglk = gammaLogLk(nullptr);
            g.alpha = alpha;
            g.rescale = 1.0 / mult;
            std::vector<REAL> bl((size_t) nNodes);
            getLengths(bl.data());
            for (REAL &x: bl) x = (REAL) ((double) x * g.rescale);   /* branchlength[i] *= scale */
            setLengths(bl.data());
            return g;
        }

        /* all up-profiles at once, breadth first from the root (getUpProfile with useML, NJ.tcc:3382-3434, on a tree
           that no longer changes): one vft_posterior_profiles_blen per depth */
        void allUpProfiles() {
            std::vector<int64_t> level;
            for (int k = 0; k < 3; k++)
                if (child[3 * root + k] >= nSeqs) level.push_back(child[3 * root + k]);
            while (!level.empty()) {
                std::vector<int64_t> out, a, b, la, lb, next;
                for (int64_t x: level) {
                    int64_t cd[2], lcd[2];
                    quartetCD(x, cd, lcd);
                    out.push_back(x + nSeqs);
                    a.push_back(cd[0]);
                    b.push_back(cd[1]);
                    la.push_back(lcd[0]);
                    lb.push_back(lcd[1]);
                    for (int k = 0; k < 2; k++)
                        if (child[3 * x + k] >= nSeqs) next.push_back(child[3 * x + k]);
                }
                chk(vft_posterior_profiles_blen(ctx, (int64_t) out.size(), out.data(), a.data(), b.data(), la.data(), lb.data()));
                level.swap(next);
            }
        }

        /* C and D of setupABCD (NJ.tcc:1942-1975) for node x - profile ids and the nodes whose branch lengths go with
           them: the two other children of the root, or the sibling and the parent's up-profile */
        void quartetCD(int64_t x, int64_t cd[2], int64_t lcd[2]) const {
            const int64_t p = parent[(size_t) x];
            if (p == root) {
                int n = 0;
                for (int k = 0; k < 3; k++)
                    if (child[3 * root + k] != x) {
                        cd[n] = lcd[n] = child[3 * root + k];
                        n++;
                    }
            } else {
                cd[0] = lcd[0] = child[3 * p] == x ? child[3 * p + 1] : child[3 * p];
                cd[1] = p + nSeqs;
                lcd[1] = p;
            }
        }

        struct SplitTests {
            int64_t nSplits = 0, nBadSplits = 0;
            double worstDelta = 0;           /* dWorstDeltaUnconstrained */
            std::vector<double> support;     /* per node, -1 where there is none */
        };

        /* testSplitsML (NJ.tcc:6800-6999) without constraints: every internal split's three quartet likelihoods in one
           device call; a split is bad when an alternative beats it by more than treeLogLkDelta = 0.1; nBootstrap > 0:
           SH-like supports from the resamples col[nBootstrap][nPos], 0 for bad splits */
        SplitTests testSplits(double ftol, double atol, int32_t nBootstrap, const int32_t *col, bool alwaysSecondPass = false) {
            allUpProfiles();
            std::vector<int64_t> nodes, ids, li;
            for (int64_t v: order) {   /* post-order, as the reference visits them */
                if (v == root) continue;
                int64_t cd[2], lcd[2];
                quartetCD(v, cd, lcd);
                nodes.push_back(v);
                const int64_t q[4] = {child[3 * v], child[3 * v + 1], cd[0], cd[1]};
                const int64_t l[5] = {child[3 * v], child[3 * v + 1], lcd[0], lcd[1], v};
                ids.insert(ids.end(), q, q + 4);
                li.insert(li.end(), l, l + 5);
            }
            const int64_t n = (int64_t) nodes.size();
            std::vector<double> loglk((size_t) (3 * n)), sup((size_t) n, 0.0);
            chk(vft_ml_split_tests(ctx, n, ids.data(), li.data(), ftol, atol, /*closeLogLkLimit*/5.0, alwaysSecondPass ? 1 : 0,
                                   loglk.data(), nBootstrap, col, nBootstrap > 0 ? sup.data() : nullptr, nullptr));
            SplitTests out;
            out.support.assign((size_t) nNodes, -1.0);
            for (int64_t k = 0; k < n; k++) {
                const double *l = &loglk[(size_t) (3 * k)];
                int choice;
                if (l[0] >= l[1] && l[0] >= l[2]) choice = 0;
                else if (l[1] >= l[0] && l[1] >= l[2]) choice = 1;
                else choice = 2;
                const bool bad = l[choice] > l[0] + 0.1;   /* Constants::treeLogLkDelta */
                out.nSplits++;
                if (bad) {
                    out.nBadSplits++;
                    out.worstDelta = std::max(out.worstDelta, l[choice] - l[0]);
                }
                if (nBootstrap > 0) out.support[(size_t) nodes[(size_t) k]] = bad ? 0.0 : sup[(size_t) k];
            }
            splitLoglk.swap(loglk);
            splitNodes.swap(nodes);
            return out;
        }

        std::vector<double> splitLoglk;     /* [3] per entry of splitNodes: AB|CD, AC|BD, AD|BC */
        std::vector<int64_t> splitNodes;

        /* ---------------------------------------------------------------------------------------------- NNIs
           DoNNI (NJ.tcc:5797-6200, one thread, fastNNI, no constraints): a post-order walk over the internal nodes that
           compares AB|CD with AC|BD and AD|BC around each - by log-corrected profile distances (chooseNNI,
           NJ.tcc:4836-4883; useML = false) or by optimised quartet likelihoods (MLQuartetNNI, one
           vft_ml_quartet_nni call; useML = true) - rearranges the tree on the spot and keeps the profiles around the node
           current.  The tree changes under the walk, so the device work of a node is queued when the walk gets there
           and the host waits for each verdict. */
        /* ---- lanes (see "the subtree schedule" above) */
        struct Lane {
            int64_t R = -1;                  /* root of the subtree; -1: the serial walk from the tree's root */
            int j = 0, k = 0;                /* NNIs: next branch = child k of child j of R (read when the walk gets there) */
            int64_t node = -1, branchRoot = -1;
            bool inBranch = false, finished = false, hasRequest = false;
            int64_t qnode = -1, q[4] = {-1, -1, -1, -1}, idD = -1;   /* the quartet waiting for its verdict */
            std::vector<int64_t> roots;      /* branch roots walked so far */
            std::vector<int64_t> nodes;      /* lengths: the internal nodes of the lane in post-order */
            size_t pos = 0;
            std::vector<int64_t> out, a, b, la, lb;   /* profile ops queued since the last step, in order */
        };
        struct SharedOp {
            int64_t depth, out, a, b, la, lb;
        };
        Lane *cur = nullptr;
        std::vector<SharedOp> sharedOps;

        struct NNIStats {
            int64_t age, subtreeAge;
            double delta, support;
        };

        void initNNIStats(std::vector<NNIStats> &stats) const {   /* NJ.tcc:7002-7016 */
            stats.assign((size_t) nNodes, NNIStats{0, 0, 0.0, 0.0});
            for (int64_t i = 0; i < nNodes; i++)
                if (i == root || i < nSeqs) stats[(size_t) i].age = stats[(size_t) i].subtreeAge = 1000000;
        }

        struct NNIParams {
            bool useML = false;
            bool scoredist = false;      /* logCorrect flavour of the ME criterion (NJ.tcc:322-330) */
            double ftol = 0.001, atol = 1e-4, minDelta = 1.0e-4 /* MEMinDelta */;
            int32_t mlAccuracy = 1;
        };

        int64_t doNNI(const NNIParams &prm, std::vector<NNIStats> &stats, double &dMaxDelta) {
            const double supportThreshold = prm.useML ? 0.1 /* treeLogLkDelta */ : prm.minDelta;
            int64_t nNNIThisRound = 0;
            dMaxDelta = 0.0;
            if (nSeqs <= 3) return 0;
            std::vector<char> traversal((size_t) nNodes, 0), upHave((size_t) nNodes, 0);
            /* nodes whose subtree has been quiet for two rounds are not entered (NJ.tcc:6047-6075) */
            for (int64_t node = nSeqs; node < nNodes; node++) {
                const NNIStats &st = stats[(size_t) node];
                if (node != root && st.age >= 2 && st.subtreeAge >= 2 && st.support > supportThreshold) {
                    int64_t q[4];
                    quartetNodes(node, q);
                    int i;
                    for (i = 0; i < 4; i++)
                        if (stats[(size_t) q[i]].age == 0 && stats[(size_t) q[i]].support > supportThreshold) break;
                    if (i == 4) traversal[(size_t) node] = 1;
                }
            }
            int64_t node = root;
            bool bUp = false;
            WalkServerGuard server(*this, !prm.useML);   /* minimum evolution: every step is a vft_walk_step */
            while ((node = nextPostorder(node, traversal, &bUp, root)) >= 0) {
                if (node < nSeqs || node == root) continue;
                if (bUp) {   /* back at a node whose surroundings were rearranged: refresh it (NJ.tcc:5809-5820) */
                    for (int k = 0; k < 2; k++) upHave[(size_t) child[3 * node + k]] = 0;
                    upHave[(size_t) node] = 0;
                    recomputeProfile(node, prm.useML);
                    continue;
                }
                int64_t q[4];
                quartetNodes(node, q);
                const int64_t nodeA = q[0], nodeB = q[1], nodeC = q[2], nodeD = q[3], par = parent[(size_t) node];
                int64_t idD = nodeD;   /* profile of D: the root's other child, or the parent's up-profile */
                if (par != root) {
                    ensureUpProfile(par, prm.useML, upHave);
                    idD = par + nSeqs;
                }
                int choice = 0;
                double criteria[3];
                if (prm.useML) {
                    const int64_t ids[4] = {nodeA, nodeB, nodeC, idD}, li[5] = {nodeA, nodeB, nodeC, nodeD, node};
                    vft_quartet_nni r;
                    flushPosteriors();
                    chk(vft_ml_quartet_nni(ctx, 1, ids, li, prm.ftol, prm.atol, /*closeLogLkLimit*/5.0, prm.mlAccuracy, &r));
                    choice = r.choice;
                    for (int i = 0; i < 3; i++) criteria[i] = r.criteria[i];
                    if (r.star) nStarTests++;
                } else {
                    int64_t qq[4];
                    meCriteria(node, prm.scoredist, upHave, qq, criteria);
                    if (criteria[1] < criteria[0] && criteria[1] <= criteria[2]) choice = 1;
                    else if (criteria[2] < criteria[0] && criteria[2] <= criteria[1]) choice = 2;
                    for (int i = 0; i < 3; i++) criteria[i] = -criteria[i];   /* higher is better, as for ML */
                }
                if (choice == 1) {          /* swap B and C */
                    replaceChild(node, nodeB, nodeC);
                    replaceChild(par, nodeC, nodeB);
                } else if (choice == 2) {   /* swap A and C */
                    replaceChild(node, nodeA, nodeC);
                    replaceChild(par, nodeC, nodeA);
                }
                NNIStats &st = stats[(size_t) node];
                if (choice == 0) {
                    st.age++;
                } else {
                    nNNIThisRound++;
                    st.age = 0;
                    stats[(size_t) nodeA].age = stats[(size_t) nodeB].age = stats[(size_t) nodeC].age = stats[(size_t) nodeD].age = 0;
                }
                st.delta = criteria[choice] - criteria[0];
                if (st.delta > dMaxDelta) dMaxDelta = st.delta;
                st.support = 1e20;
                for (int i = 0; i < 3; i++)
                    if (choice != i && criteria[choice] - criteria[i] < st.support) st.support = criteria[choice] - criteria[i];
                if (st.delta > supportThreshold) {
                    st.subtreeAge = 0;
                } else {
                    st.subtreeAge++;
                    for (int i = 0; i < 2; i++) {
                        const int64_t ch = child[3 * node + i];
                        if (st.subtreeAge > stats[(size_t) ch].subtreeAge) st.subtreeAge = stats[(size_t) ch].subtreeAge;
                    }
                }
                if (choice == 0) {
                    upHave[(size_t) nodeA] = upHave[(size_t) nodeB] = upHave[(size_t) nodeC] = 0;
                    recomputeProfile(node, prm.useML);
                } else {
                    updateForNNI(node, prm.useML, upHave);
                }
            }
            server.finish();
            flushPosteriors();
            rebuildOrder();
            return nNNIThisRound;
        }

        /* ---- lanes across ranks (include/vft_host.h, vft_comm; SURVEY 8e "ML phase").  Every rank runs the same walks on the same
           tree and keeps the whole device state; what is split is the expensive part of a lockstep step - the batch of quartets to
           judge (vft_ml_quartet_nni_flags: five Brent searches per pairing), of splits to optimise (vft_ml_optimize_splits), of
           minimum-evolution distances.  Rank r evaluates items [r * per, (r + 1) * per) of the batch, the ranks all-gather the
           verdicts together with the branch lengths the kernels wrote (the only device state a verdict kernel changes; the
           splits' kernel also rewrites its node's profile, which the other ranks redo with the gathered lengths), and scatter the
           others' lengths into their own branchlength[].  The same kernels judge the same inputs, whoever runs them: the tree is the
           single-rank tree byte for byte (tests/test_gpu_threads.py).  One all-gather of a few hundred bytes per item and step. */
        const vft_comm *comm = nullptr;
        int64_t laneGathers = 0, laneGatherBytes = 0, treeLogLkSharded = 0;
        bool sharded() const { return comm != nullptr && comm->world > 1; }
        /* rank r's share [k0, k1) of a batch of K items over W ranks, and the padded share `per` every rank sends; item t of the batch
           is record laneRecord(t, per) of the gathered buffer (exported as vft_nj_lane_share for the CPU test of the exchange layout) */
        static void laneShare(size_t K, size_t W, size_t r, size_t &per, size_t &k0, size_t &k1) {
            per = (K + W - 1) / W;
            k0 = std::min(K, r * per);
            k1 = std::min(K, k0 + per);
        }
        static size_t laneRecord(size_t t, size_t per) { return (t / per) * per + t % per; }
        void shareOf(size_t K, size_t &per, size_t &k0, size_t &k1) const { laneShare(K, (size_t) comm->world, (size_t) comm->rank, per, k0, k1); }
        /* all-gather `per` records of `rec` bytes per rank: send = this rank's records (h_send), received in rank order (h_recv) */
        const char *gatherRecords(size_t per, size_t rec) {
            const int64_t bytes = (int64_t) (per * rec);
            if (bytes > comm->h_cap) throw std::invalid_argument("MLLengths: vft_comm host buffers too small for a batch of lanes");
            if (comm->allgather(comm->user, bytes, 0) != 0) throw std::runtime_error("MLLengths: all-gather of the lanes' verdicts failed");
            laneGathers++;
            laneGatherBytes += bytes * (int64_t) comm->world;
            return (const char *) comm->h_recv;
        }

        /* traverseNNI (NJ.tcc:5797-5990) for a set of independent walks in lockstep: every lane advances to its next
           quartet, the queued profile work of all lanes goes down as one launch, all quartets are judged in one batch, every
           lane applies its verdict - the statements are doNNI's, the evaluation is deferred */
        void runNNILanes(std::vector<Lane> &lanes, const NNIParams &prm, std::vector<NNIStats> &stats, std::vector<char> &traversal,
                         std::vector<char> &upHave, bool starTest, int64_t &nNNIThisRound, double &dMaxDelta) {
            const double supportThreshold = prm.useML ? 0.1 /* treeLogLkDelta */ : prm.minDelta;
            std::vector<Lane *> asking;
            std::vector<int64_t> ids, li, pi, pj;
            std::vector<vft_quartet_nni> res;
            std::vector<REAL> d, w;
            /* (VFT_LANE_STATS: where a lockstep step's wall-clock goes - the lanes' host walks, the chain launches, the batch of quartets
               with its wait, the verdicts applied on the host) */
            static const bool laneStats = std::getenv("VFT_LANE_STATS") != nullptr;
            double tAdvance = 0, tChains = 0, tBatch = 0, tApply = 0;
            int64_t steps0 = laneSteps;
            auto now = []() { return std::chrono::steady_clock::now(); };
            auto since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
            struct Report {
                const bool on;
                const double &a, &c, &b, &p;
                const int64_t &s0, &s1;
                const size_t nLanes;
                const bool ml;
                ~Report() {
                    if (on && s1 > s0)
                        fprintf(stderr, "lanes (%s, %zu lanes): %lld lockstep steps; host walks %.3f s, chains %.3f s, quartet batches %.3f s, verdicts %.3f s\n", ml ? "ML" : "ME", nLanes,
                                (long long) (s1 - s0), a, c, b, p);
                }
            } report{laneStats, tAdvance, tChains, tBatch, tApply, steps0, laneSteps, lanes.size(), prm.useML};
            for (;;) {
                std::chrono::steady_clock::time_point t0 = now();
                for (Lane &ln: lanes)
                    if (!ln.finished && !ln.hasRequest) advanceNNILane(ln, prm, traversal, upHave);
                if (laneStats) { tAdvance += since(t0); t0 = now(); }
                runShared(prm.useML);
                runChains(lanes, prm.useML);
                if (laneStats) { tChains += since(t0); t0 = now(); }
                asking.clear();
                for (Lane &ln: lanes)
                    if (ln.hasRequest) asking.push_back(&ln);
                if (asking.empty()) break;
                const size_t K = asking.size();
                laneSteps++;
                laneWork += (int64_t) K;
                if (prm.useML) {
                    ids.resize(4 * K);
                    li.resize(5 * K);
                    res.resize(K);
                    for (size_t t = 0; t < K; t++) {
                        const Lane &ln = *asking[t];
                        const int64_t q4[4] = {ln.q[0], ln.q[1], ln.q[2], ln.idD}, l5[5] = {ln.q[0], ln.q[1], ln.q[2], ln.q[3], ln.qnode};
                        std::copy(q4, q4 + 4, ids.begin() + (long) (4 * t));
                        std::copy(l5, l5 + 5, li.begin() + (long) (5 * t));
                    }
                    const size_t maxBatch = 512;
                    size_t per = K, m0 = 0, m1 = K;
                    if (sharded()) shareOf(K, per, m0, m1);
                    for (size_t k0 = m0; k0 < m1; k0 += maxBatch) {
                        const size_t cnt = std::min(maxBatch, m1 - k0);
                        chk(vft_ml_quartet_nni_flags(ctx, (int64_t) cnt, ids.data() + 4 * k0, li.data() + 5 * k0, prm.ftol, prm.atol, /*closeLogLkLimit*/5.0,
                                                     prm.mlAccuracy, starTest ? 0 : VFT_QUARTET_NO_STAR_TEST, res.data() + k0));
                    }
                    if (sharded()) {   /* verdicts + the five lengths each verdict kernel wrote: gathered, the others' lengths scattered */
                        const size_t rec = sizeof(vft_quartet_nni) + 5 * sizeof(REAL);
                        std::vector<REAL> mine(5 * (m1 - m0));
                        chk(vft_branch_lengths_gather(ctx, (int64_t) mine.size(), li.data() + 5 * m0, mine.data()));
                        char *snd = (char *) comm->h_send;
                        if ((int64_t) (per * rec) > comm->h_cap) throw std::invalid_argument("MLLengths: vft_comm host buffers too small for a batch of lanes");
                        for (size_t t = m0; t < m1; t++) {
                            std::memcpy(snd + (t - m0) * rec, &res[t], sizeof(vft_quartet_nni));
                            std::memcpy(snd + (t - m0) * rec + sizeof(vft_quartet_nni), &mine[5 * (t - m0)], 5 * sizeof(REAL));
                        }
                        const char *rcv = gatherRecords(per, rec);
                        std::vector<int64_t> si;
                        std::vector<REAL> sv;
                        for (size_t t = 0; t < K; t++) {
                            if (t >= m0 && t < m1) continue;
                            const char *src = rcv + laneRecord(t, per) * rec;
                            std::memcpy(&res[t], src, sizeof(vft_quartet_nni));
                            REAL five[5];
                            std::memcpy(five, src + sizeof(vft_quartet_nni), sizeof(five));
                            for (int j = 0; j < 5; j++) {
                                si.push_back(li[5 * t + (size_t) j]);
                                sv.push_back(five[j]);
                            }
                        }
                        for (size_t f = 0; f < si.size(); f += 65536)
                            chk(vft_branch_lengths_scatter(ctx, (int64_t) std::min<size_t>(65536, si.size() - f), si.data() + f, sv.data() + f));
                    }
                } else {
                    pi.resize(6 * K);
                    pj.resize(6 * K);
                    d.resize(6 * K);
                    w.resize(6 * K);
                    for (size_t t = 0; t < K; t++) {
                        const Lane &ln = *asking[t];
                        const int64_t a[6] = {ln.q[0], ln.q[0], ln.q[0], ln.q[1], ln.q[1], ln.q[2]}, b[6] = {ln.q[1], ln.q[2], ln.idD, ln.q[2], ln.idD, ln.idD};
                        std::copy(a, a + 6, pi.begin() + (long) (6 * t));
                        std::copy(b, b + 6, pj.begin() + (long) (6 * t));
                    }
                    if (sharded()) {   /* six distances per quartet: this rank's share computed, all shares gathered */
                        size_t per, m0, m1;
                        shareOf(K, per, m0, m1);
                        if (m1 > m0) chk(vft_profile_distances(ctx, (int64_t) (6 * (m1 - m0)), pi.data() + 6 * m0, pj.data() + 6 * m0, d.data() + 6 * m0, w.data() + 6 * m0));
                        const size_t rec = 6 * sizeof(REAL);
                        if ((int64_t) (per * rec) > comm->h_cap) throw std::invalid_argument("MLLengths: vft_comm host buffers too small for a batch of lanes");
                        std::memcpy(comm->h_send, d.data() + 6 * m0, (m1 - m0) * rec);
                        const char *rcv = gatherRecords(per, rec);
                        for (size_t t = 0; t < K; t++)
                            if (t < m0 || t >= m1) std::memcpy(&d[6 * t], rcv + laneRecord(t, per) * rec, rec);
                    } else
                    chk(vft_profile_distances(ctx, (int64_t) (6 * K), pi.data(), pj.data(), d.data(), w.data()));
                }
                if (laneStats) { tBatch += since(t0); t0 = now(); }
                for (size_t t = 0; t < K; t++) {
                    Lane &ln = *asking[t];
                    int choice = 0;
                    double criteria[3];
                    if (prm.useML) {
                        choice = res[t].choice;
                        for (int i = 0; i < 3; i++) criteria[i] = res[t].criteria[i];
                        if (res[t].star) nStarTests++;
                    } else {
                        double c[6];
                        for (int i = 0; i < 6; i++) c[i] = logCorrect((double) d[6 * t + (size_t) i], prm.scoredist);
                        criteria[0] = c[0] + c[5];
                        criteria[1] = c[1] + c[4];
                        criteria[2] = c[2] + c[3];
                        if (criteria[1] < criteria[0] && criteria[1] <= criteria[2]) choice = 1;
                        else if (criteria[2] < criteria[0] && criteria[2] <= criteria[1]) choice = 2;
                        for (int i = 0; i < 3; i++) criteria[i] = -criteria[i];
                    }
                    cur = &ln;
                    applyNNIVerdict(ln.qnode, ln.q, choice, criteria, prm.useML, supportThreshold, stats, upHave, nNNIThisRound, dMaxDelta);
                    cur = nullptr;
                    ln.hasRequest = false;
                }
                if (laneStats) tApply += since(t0);
            }
        }

        /* the walk of one lane up to its next quartet */
        void advanceNNILane(Lane &ln, const NNIParams &prm, std::vector<char> &traversal, std::vector<char> &upHave) {
            cur = &ln;
            for (;;) {
                if (!ln.inBranch) {
                    if (ln.R < 0) {   /* the serial walk has the whole tree as its one branch */
                        ln.finished = true;
                        break;
                    }
                    /* the branches of a subtree: the children of its root's children, read when the walk gets there - an NNI
                       of an earlier branch may have put a different node in the slot (NJ.tcc:6139-6147) */
                    bool found = false;
                    while (ln.j < 2 && child[3 * ln.R + ln.j] >= 0) {
                        const int64_t dc = child[3 * ln.R + ln.j];
                        if (ln.k < 2 && child[3 * dc + ln.k] >= 0) {
                            ln.node = ln.branchRoot = child[3 * dc + ln.k];
                            ln.k++;
                            ln.roots.push_back(ln.branchRoot);
                            ln.inBranch = found = true;
                            break;
                        }
                        ln.j++;
                        ln.k = 0;
                    }
                    if (!found) {
                        ln.finished = true;
                        break;
                    }
                }
                bool bUp = false;
                const int64_t node = nextPostorder(ln.node, traversal, &bUp, ln.branchRoot);
                if (node < 0) {
                    ln.inBranch = false;
                    continue;
                }
                ln.node = node;
                if (node < nSeqs || node == root) continue;
                if (bUp) {   /* NJ.tcc:5809-5820 */
                    for (int k = 0; k < 2; k++) upHave[(size_t) child[3 * node + k]] = 0;
                    upHave[(size_t) node] = 0;
                    recomputeProfile(node, prm.useML);
                    continue;
                }
                quartetNodes(node, ln.q);
                const int64_t par = parent[(size_t) node];
                ln.idD = ln.q[3];
                if (par != root) {
                    ensureUpProfile(par, prm.useML, upHave);
                    ln.idD = par + nSeqs;
                }
                ln.qnode = node;
                ln.hasRequest = true;
                break;
            }
            cur = nullptr;
        }

        /* what traverseNNI does with a verdict (NJ.tcc:5880-5980): rearrange, statistics, profiles */
        void applyNNIVerdict(int64_t node, const int64_t q[4], int choice, const double criteria[3], bool useML, double supportThreshold,
                             std::vector<NNIStats> &stats, std::vector<char> &upHave, int64_t &nNNIThisRound, double &dMaxDelta) {
            const int64_t nodeA = q[0], nodeB = q[1], nodeC = q[2], nodeD = q[3], par = parent[(size_t) node];
            if (choice == 1) {          /* swap B and C */
                replaceChild(node, nodeB, nodeC);
                replaceChild(par, nodeC, nodeB);
            } else if (choice == 2) {   /* swap A and C */
                replaceChild(node, nodeA, nodeC);
                replaceChild(par, nodeC, nodeA);
            }
            NNIStats &st = stats[(size_t) node];
            if (choice == 0) {
                st.age++;
            } else {
                nNNIThisRound++;
                st.age = 0;
                stats[(size_t) nodeA].age = stats[(size_t) nodeB].age = stats[(size_t) nodeC].age = stats[(size_t) nodeD].age = 0;
            }
            st.delta = criteria[choice] - criteria[0];
            if (st.delta > dMaxDelta) dMaxDelta = st.delta;
            st.support = 1e20;
            for (int i = 0; i < 3; i++)
                if (choice != i && criteria[choice] - criteria[i] < st.support) st.support = criteria[choice] - criteria[i];
            if (st.delta > supportThreshold) {
                st.subtreeAge = 0;
            } else {
                st.subtreeAge++;
                for (int i = 0; i < 2; i++) {
                    const int64_t ch = child[3 * node + i];
                    if (st.subtreeAge > stats[(size_t) ch].subtreeAge) st.subtreeAge = stats[(size_t) ch].subtreeAge;
                }
            }
            if (choice == 0) {
                upHave[(size_t) nodeA] = upHave[(size_t) nodeB] = upHave[(size_t) nodeC] = 0;
                recomputeProfile(node, useML);
            } else {
                updateForNNI(node, useML, upHave);
            }
        }

        /* one node of traverseSPR (NJ.tcc:6213-6300): its up to four chains of forced minimum-evolution NNIs, best prefix kept, rest
           unwound; true when the tree changed */
        bool sprAttempt(int64_t node, bool scoredist, int maxSPRLength, std::vector<char> &upHave) {
            if (node == root) return false;
            struct Step {
                int64_t nodes[2];
                double deltaLength;
                bool bc;
            };
            Step steps[64];
            if (maxSPRLength > 64) throw std::invalid_argument("MLLengths::sprAttempt: chains of at most 64 steps");
            int64_t nodeAround[2];
            movePivots(node, nodeAround);
            bool bChanged = false;
            for (int iAround = 0; iAround < 2 && !bChanged; iAround++) {
                for (int acFirst = 0; acFirst < 2 && !bChanged; acFirst++) {
                    /* findSPRSteps.  The first NNI of a chain is forced (AC or AD by acFirst), its criteria only price it: the
                       second step is built and handed over before the first one's distances are waited for */
                    int64_t around = nodeAround[iAround], chainLength = 0;
                    MeTicket first;
                    bool firstOpen = false;
                    /* round 6: from the second step of a chain on, the step AFTER the one that is running is built for both outcomes of
                       its comparison and handed over as a dual command (specContinuations): `adopt` = the next step is already on the
                       device - the alternative the workgroups took - and the host state has been moved to it (txnRedo) */
                    bool adopt = false;
                    struct {
                        int64_t q[4];
                        MeTicket t;
                        uint32_t ticket;
                        int alt;
                    } adopted;
                    adopted.ticket = 0;
                    adopted.alt = 0;
                    for (; chainLength < maxSPRLength; chainLength++) {
                        if (around < nSeqs || around == root) break;   /* nChild != 2 */
                        int64_t q[4];
                        double criteria[3];
                        MeTicket tk;
                        bool verifyDual = false;
                        if (adopt) {
                            for (int i = 0; i < 4; i++) q[i] = adopted.q[i];
                            tk = adopted.t;
                            tk.pending = true;
                            tk.ticket = adopted.ticket;
                            verifyDual = true;
                            adopt = false;
                        } else {
                            meSubmit(around, upHave, q, tk);
                        }
                        sprSteps++;
                        Step &st = steps[(size_t) chainLength];
                        bool swapBC;
                        /* both continuations of THIS step, before its distances are waited for (the forced first step has one) */
                        bool dualSent = false;
                        if (chainLength >= 1 && walkDual && serverUp && tk.pending) {
                            if (walkStats) {
                                const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                                dualSent = specContinuations(node, around, q, (int) chainLength, maxSPRLength, scoredist, upHave);
                                walkSpecSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                            } else
                            dualSent = specContinuations(node, around, q, (int) chainLength, maxSPRLength, scoredist, upHave);
                        }
                        if (chainLength == 0) {
                            swapBC = acFirst != 0;
                            first = tk;
                            firstOpen = true;
                        } else {
                            if (firstOpen) {
                                double c0[3];
                                meCollect(first, scoredist, c0);
                                steps[0].deltaLength = (steps[0].bc ? c0[1] : c0[2]) - c0[0];
                                firstOpen = false;
                            }
                            meCollect(tk, scoredist, criteria);
                            if (verifyDual) {   /* the workgroups made the same comparison on the same six numbers */
                                int32_t alt = -1, skipped = -1;
                                chk(vft_walk_dual_choice(ctx, tk.ticket, &alt, &skipped));
                                if (alt != adopted.alt || skipped) throw std::logic_error("MLLengths::sprAttempt: the walk server took another continuation than the host");
                            }
                            swapBC = criteria[1] < criteria[2];
                            st.deltaLength = (swapBC ? criteria[1] : criteria[2]) - criteria[0];
                        }
                        st.bc = swapBC;
                        if (swapBC) {   /* swap B and C: AC together */
                            st.nodes[0] = q[1];
                            st.nodes[1] = q[2];
                        } else {        /* swap A and C: AD together */
                            st.nodes[0] = q[0];
                            st.nodes[1] = q[2];
                        }
                        if (dualSent && specAlt[swapBC ? 0 : 1].valid) {
                            /* the device is running the next step already: move the host state to where building it had left it */
                            SprAlt &A = specAlt[swapBC ? 0 : 1];
                            txnRedo(A.log);
                            walkDualTaken++;
                            for (int i = 0; i < 4; i++) adopted.q[i] = A.q[i];
                            adopted.t = A.t;
                            adopted.ticket = A.ticket;
                            adopted.alt = swapBC ? 0 : 1;
                            adopt = true;
                            around = A.aroundNext;
                            continue;
                        }
                        replaceChild(around, st.nodes[0], st.nodes[1]);
                        replaceChild(parent[(size_t) around], st.nodes[1], st.nodes[0]);
                        updateForNNI(around, false, upHave);
                        int64_t next[2];
                        movePivots(node, next);
                        around = next[next[0] == around ? 1 : 0];
                    }
                    if (adopt) {   /* (cannot happen: a continuation is only valid while the chain goes on - but a step on the device must be collected) */
                        MeTicket tk = adopted.t;
                        tk.pending = true;
                        tk.ticket = adopted.ticket;
                        double c[3];
                        meCollect(tk, scoredist, c);
                        throw std::logic_error("MLLengths::sprAttempt: a continuation was adopted beyond the chain's end");
                    }
                    if (firstOpen) {
                        double c0[3];
                        meCollect(first, scoredist, c0);
                        steps[0].deltaLength = (steps[0].bc ? c0[1] : c0[2]) - c0[0];
                    }
                    double dMinDelta = 0.0, dTotDelta = 0.0;
                    int64_t iCBest = -1;
                    for (int64_t iC = 0; iC < chainLength; iC++) {
                        dTotDelta += steps[(size_t) iC].deltaLength;
                        if (dTotDelta < dMinDelta) {
                            dMinDelta = dTotDelta;
                            iCBest = iC;
                        }
                    }
                    for (int64_t iC = chainLength - 1; iC > iCBest; iC--) {   /* unwindSPRStep */
                        const Step &st = steps[(size_t) iC];
                        const int64_t p0 = parent[(size_t) st.nodes[0]], p1 = parent[(size_t) st.nodes[1]];
                        replaceChild(p0, st.nodes[0], st.nodes[1]);
                        replaceChild(p1, st.nodes[1], st.nodes[0]);
                        updateForNNI(parent[(size_t) p0] == p1 ? p0 : p1, false, upHave);
                    }
                    if (iCBest >= 0) bChanged = true;
                }
            }
            if (bChanged) {
                std::fill(upHave.begin(), upHave.end(), 0);
                for (int64_t anc = parent[(size_t) node]; anc >= 0; anc = parent[(size_t) anc]) recomputeProfile(anc, false);
            }
            return bChanged;
        }

        /* SPR (NJ.tcc:6185-6404, one thread, fast flavour): every node in post-order is moved along chains of up to
           maxSPRLength minimum-evolution NNIs around its parent and its sibling (findSPRSteps, NJ.tcc:1805-1859; the
           first step forced to AC or AD), keeping the prefix of the chain with the best total length change and
           unwinding the rest (unwindSPRStep, NJ.tcc:1861-1879).  Returns the number of accepted moves. */
        int64_t doSPR(bool scoredist, int maxSPRLength = 10) {
            if (nSeqs <= 3 || maxSPRLength < 1) return 0;
            std::vector<char> traversal((size_t) nNodes, 0), upHave((size_t) nNodes, 0);
            std::vector<int64_t> nodeList;
            {   /* the walk is fixed before anything moves */
                int64_t node = root;
                bool up;
                while ((node = nextPostorder(node, traversal, &up, root, /*reportUp*/false)) >= 0) nodeList.push_back(node);
            }
            int64_t nSPR = 0;
            WalkServerGuard server(*this);
            if (serverUp) chk(vft_walk_scoredist(ctx, scoredist ? 1 : 0));
            static const bool stageTrace = std::getenv("VFT_STAGE_TRACE") != nullptr;
            int64_t nDone = 0;
            for (int64_t node: nodeList) {
                if (sprAttempt(node, scoredist, maxSPRLength, upHave)) nSPR++;
                if (stageTrace && (++nDone % 100000) == 0) fprintf(stderr, "[stage]   SPR: %lld of %zu nodes, %lld steps, %lld moves\n", (long long) nDone, nodeList.size(), (long long) sprSteps, (long long) nSPR);
            }
            server.finish();
            rebuildOrder();
            return nSPR;
        }

        /* ---------------------------------------------------------------------------------------------- the subtree schedule
           What the reference does with `-threads T` (T > 1, the default -threads-level 3): treePartitioning
           (NJ.tcc:5540-5750) picks an antichain of subtrees, every OpenMP thread walks its subtrees with a private up-profile
           cache, then one thread walks what is left from the root.  The result does not depend on the thread timing: inside
           the parallel phase a walk reads nothing outside its subtree that another walk writes (the subtree roots and, for
           NNIs, their children are never the centre of a step, NJ.tcc:6135-6147, :5092-5096), so the up-profiles at and
           above a subtree root are the same whoever builds them.  It DOES depend on T (the partition) and differs from the
           one-thread order.  Here every subtree is a LANE; all lanes advance in lockstep, one step = one launch of their
           up-profile / recompute chains (vft_*_chains) and one batch of their quartets / splits - the independent work the
           GPU needs.  Output is byte-identical to `VeryFastTree -threads T` (tests/test_gpu_threads.py, fixtures from
           oracle/gen_fixtures.py threads). */

        /* treePartitioning: roots of the subtrees, in the order the reference hands them to its threads */
        std::vector<int64_t> treePartitioning(int penalty, int threads, int window = 50) {
            return partitionTree(nNodes, child, root, order, penalty, threads, window, &partitionSpeedup);
        }

        /* the algorithm itself, on plain arrays (exported as vft_tree_partitioning for the CPU test that pins it to the reference's
           own partitions, tests/test_abi_cpu.py): child[nNodes][3], order = internal nodes in post-order, the root last */
        static std::vector<int64_t> partitionTree(int64_t nNodes, const std::vector<int64_t> &child, int64_t root, const std::vector<int64_t> &order,
                                                  int penalty, int threads, int window, double *speedupOut) {
            const size_t N = (size_t) nNodes;
            const bool ptTrace = std::getenv("VFT_PARTITION_TRACE") != nullptr;
            const std::chrono::steady_clock::time_point pt0 = std::chrono::steady_clock::now();
            auto ptMark = [&](const char *what) {
                if (ptTrace) fprintf(stderr, "[partition] %-28s %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - pt0).count());
            };
            std::vector<int64_t> size(N, 1), depth(N, 0);
            for (int64_t v: order) {   /* children before parents */
                int64_t w = 1;
                for (int k = 0; k < 3 && child[3 * v + k] >= 0; k++) w += size[(size_t) child[3 * v + k]];
                size[(size_t) v] = w;
            }
            int64_t deepest = 0;
            for (size_t t = order.size(); t-- > 0;) {   /* parents before children */
                const int64_t v = order[t];
                for (int k = 0; k < 3 && child[3 * v + k] >= 0; k++) {
                    const int64_t c = child[3 * v + k];
                    depth[(size_t) c] = depth[(size_t) v] + 1;
                    if (depth[(size_t) c] > deepest) deepest = depth[(size_t) c];
                }
            }
            ptMark("sizes and depths");
            /* the weight of a candidate: the nodes `penalty` levels below it and deeper (what a walk may touch), 0 for
               candidates too close to the deepest level.  Worked out when a node becomes a candidate (a few thousand of two million
               nodes do: filling the table for every node was a third of a partition, and a round of a million-sequence tree partitions
               once) */
            std::vector<int64_t> weight(N, -1), frontier, next;
            auto weigh = [&](int64_t i) {
                if (weight[(size_t) i] >= 0) return;
                int64_t w = 0;
                if (deepest - depth[(size_t) i] >= penalty) {
                    frontier.assign(1, i);
                    for (int lv = 0; lv < penalty; lv++) {
                        next.clear();
                        for (int64_t v: frontier)
                            for (int k = 0; k < 3 && child[3 * v + k] >= 0; k++) next.push_back(child[3 * v + k]);
                        frontier.swap(next);
                    }
                    for (int64_t v: frontier) w += size[(size_t) v];
                }
                weight[(size_t) i] = w;
            };
            const size_t T = (size_t) threads;
            auto speedup = [&](const std::vector<int64_t> &sol) -> double {
                int64_t denom;
                if (sol.empty()) {
                    denom = nNodes;
                } else if (sol.size() <= T) {
                    int64_t inside = 0;
                    for (int64_t v: sol) inside += weight[(size_t) v];
                    denom = nNodes - inside + weight[(size_t) sol.back()];
                } else {   /* greedy: every subtree to the least loaded thread; loads kept in descending order */
                    std::vector<int64_t> load(T, 0);
                    int64_t inside = 0;
                    for (int64_t v: sol) {
                        int64_t least = load.back();
                        load.pop_back();
                        inside += weight[(size_t) v];
                        least += weight[(size_t) v];
                        load.insert(std::lower_bound(load.begin(), load.end(), least, [](int64_t a, int64_t b) { return b < a; }), least);
                    }
                    denom = nNodes - inside + load[0];
                }
                return (double) nNodes / (double) denom;
            };
            std::vector<int64_t> sol, best;   /* ascending weight; a newcomer goes in front of its equals */
            auto lighter = [&](int64_t x, int64_t y) { return weight[(size_t) x] < weight[(size_t) y]; };
            auto put = [&](int64_t v) {
                weigh(v);
                if (weight[(size_t) v] > 0) sol.insert(std::lower_bound(sol.begin(), sol.end(), v, lighter), v);
            };
            for (int k = 0; k < 3; k++) put(child[3 * root + k]);
            best = sol;
            double cur = speedup(sol), bestSpeedup = cur;
            std::vector<double> recent;   /* the last `window` + 1 speed-ups: stop once they trend downwards */
            for (;;) {
                if (sol.empty()) break;
                recent.push_back(cur);
                if (sol.size() >= T && recent.size() > (size_t) window) {
                    int64_t balance = 0;
                    for (size_t a = 0; a < recent.size(); a++)
                        for (size_t b = a; b < recent.size(); b++) balance += recent[a] <= recent[b] ? 1 : -1;
                    recent.erase(recent.begin());
                    if (balance < 0) break;
                }
                const int64_t v = sol.back();   /* split the heaviest */
                sol.pop_back();
                if (child[3 * v] < 0) continue;
                put(child[3 * v]);
                put(child[3 * v + 1]);
                cur = speedup(sol);
                if (cur > bestSpeedup) {
                    best = sol;
                    bestSpeedup = cur;
                }
            }
            ptMark("splitting loop");
            /* the hand-out: lightest first, each to the least loaded thread (ties: the thread that was touched longest ago,
               initially the last); the list is read thread-major per round, which is the order returned */
            std::vector<std::vector<int64_t>> mine(T);
            std::vector<std::pair<int64_t, int64_t>> load(T);   /* (thread, load), descending load */
            for (size_t i = 0; i < T; i++) load[i] = std::make_pair((int64_t) (T - i - 1), (int64_t) 0);
            for (int64_t v: best) {
                std::pair<int64_t, int64_t> least = load.back();
                load.pop_back();
                least.second += weight[(size_t) v];
                mine[(size_t) least.first].push_back(v);
                load.insert(std::lower_bound(load.begin(), load.end(), least,
                                             [](const std::pair<int64_t, int64_t> &a, const std::pair<int64_t, int64_t> &b) { return b.second < a.second; }),
                            least);
            }
            std::vector<int64_t> out;
            for (size_t lv = 0;; lv++) {
                bool any = false;
                for (size_t t = 0; t < T; t++)
                    if (mine[t].size() > lv) {
                        out.push_back(mine[t][lv]);
                        any = true;
                    }
                if (!any) break;
            }
            if (speedupOut) *speedupOut = bestSpeedup;
            return out;
        }

        /* DoNNI with threads > 1 (NJ.tcc:6108-6160 + the serial traverseNNI that follows): the lanes of treePartitioning(2)
           in lockstep, then the rest of the tree from the root.  Maximum likelihood: the lanes test the star topology, the
           serial walk does not (MLQuartetNNI's `omp sections` branch, NJ.tcc:4902-4948). */
        int64_t doNNIThreaded(const NNIParams &prm, std::vector<NNIStats> &stats, double &dMaxDelta, int threads) {
            const double supportThreshold = prm.useML ? 0.1 : prm.minDelta;
            int64_t nNNIThisRound = 0;
            dMaxDelta = 0.0;
            if (nSeqs <= 3) return 0;
            std::vector<char> traversal((size_t) nNodes, 0), upHave((size_t) nNodes, 0);
            for (int64_t node = nSeqs; node < nNodes; node++) {   /* quiet subtrees are not entered (NJ.tcc:6047-6075) */
                const NNIStats &st = stats[(size_t) node];
                if (node != root && st.age >= 2 && st.subtreeAge >= 2 && st.support > supportThreshold) {
                    int64_t q[4];
                    quartetNodes(node, q);
                    int i;
                    for (i = 0; i < 4; i++)
                        if (stats[(size_t) q[i]].age == 0 && stats[(size_t) q[i]].support > supportThreshold) break;
                    if (i == 4) traversal[(size_t) node] = 1;
                }
            }
            const std::vector<int64_t> subtrees = treePartitioning(2, threads);
            std::vector<Lane> lanes(subtrees.size());
            for (size_t i = 0; i < subtrees.size(); i++) lanes[i].R = subtrees[i];
            runNNILanes(lanes, prm, stats, traversal, upHave, /*starTest*/true, nNNIThisRound, dMaxDelta);
            /* the threads' private caches end with the parallel region: what survives (moveUpProfile from every branch root,
               NJ.tcc:5766-5779, :6150-6155) are the entries on the way from each branch root up to the root of the tree - every
               other entry below the subtree's root is gone */
            std::vector<char> keep((size_t) nNodes, 0);
            std::vector<int64_t> stack;
            for (const Lane &ln: lanes) {
                for (int64_t g: ln.roots)
                    for (int64_t x = g; x != ln.R && x >= 0; x = parent[(size_t) x]) keep[(size_t) x] = 1;
                stack.assign(1, ln.R);
                while (!stack.empty()) {
                    const int64_t v = stack.back();
                    stack.pop_back();
                    for (int k = 0; k < 2 && child[3 * v + k] >= 0; k++) {
                        const int64_t c = child[3 * v + k];
                        if (!keep[(size_t) c]) upHave[(size_t) c] = 0;
                        stack.push_back(c);
                    }
                }
            }
            std::vector<Lane> rest(1);
            rest[0].node = rest[0].branchRoot = root;
            rest[0].inBranch = true;
            runNNILanes(rest, prm, stats, traversal, upHave, /*starTest*/false, nNNIThisRound, dMaxDelta);
            rebuildOrder();
            return nNNIThisRound;
        }

        /* optimizeAllBranchLengths with threads > 1 and -threads-level 3 (NJ.tcc:5083-5112): the lanes of treePartitioning(1)
           - both branches below every subtree root, the root's children included - then the rest of the tree */
        void optimizeRoundThreaded(double ftol, double atol, int threads) {
            std::vector<char> upHave((size_t) nNodes, 0), done((size_t) nNodes, 0);
            const std::vector<int64_t> subtrees = treePartitioning(1, threads);
            std::vector<Lane> lanes(subtrees.size());
            std::vector<std::pair<int64_t, int>> stack;
            size_t longest = 0;
            for (size_t i = 0; i < subtrees.size(); i++) {
                Lane &ln = lanes[i];
                ln.R = subtrees[i];
                for (int b = 0; b < 2 && child[3 * ln.R + b] >= 0; b++) {   /* post-order of each branch, internal nodes only */
                    stack.assign(1, std::make_pair(child[3 * ln.R + b], 0));
                    while (!stack.empty()) {
                        const int64_t v = stack.back().first;
                        const int k = stack.back().second;
                        if (k < 2 && child[3 * v + k] >= 0) {
                            stack.back().second++;
                            stack.push_back(std::make_pair(child[3 * v + k], 0));
                        } else {
                            stack.pop_back();
                            if (child[3 * v] >= 0) ln.nodes.push_back(v);
                        }
                    }
                }
                longest = std::max(longest, ln.nodes.size());
            }
            std::vector<int64_t> ids, li, rec;
            for (size_t step = 0; step < longest; step++) {
                ids.clear(); li.clear(); rec.clear();
                for (Lane &ln: lanes) {
                    if (ln.pos >= ln.nodes.size()) continue;
                    const int64_t v = ln.nodes[ln.pos++];
                    cur = &ln;
                    ensureUpProfile(v, true, upHave);
                    cur = nullptr;
                    const int64_t q[3] = {child[3 * v], child[3 * v + 1], v + nSeqs}, l[3] = {child[3 * v], child[3 * v + 1], v};
                    ids.insert(ids.end(), q, q + 3);
                    li.insert(li.end(), l, l + 3);
                    rec.push_back(v);
                    upHave[(size_t) v] = 0;   /* NJ.tcc:5062 */
                    done[(size_t) v] = 1;
                }
                runShared(true);
                runChains(lanes, true);
                const size_t maxBatch = 2048, K = rec.size();
                size_t per = K, m0 = 0, m1 = K;
                if (sharded()) shareOf(K, per, m0, m1);
                for (size_t k0 = m0; k0 < m1; k0 += maxBatch) {
                    const size_t cnt = std::min(maxBatch, m1 - k0);
                    chk(vft_ml_optimize_splits(ctx, (int64_t) cnt, ids.data() + 3 * k0, li.data() + 3 * k0, rec.data() + k0, ftol, atol));
                }
                if (sharded() && K > 0) {
                    /* the three lengths of every split: gathered, the others' scattered; the splits' kernel also leaves its node's
                       profile recomputed from the two lengths below it (k_ml_node_lengths' tail) - the same posterior, here for the
                       splits the other ranks optimised */
                    const size_t recB = 3 * sizeof(REAL);
                    std::vector<REAL> mine(3 * (m1 - m0));
                    chk(vft_branch_lengths_gather(ctx, (int64_t) mine.size(), li.data() + 3 * m0, mine.data()));
                    if ((int64_t) (per * recB) > comm->h_cap) throw std::invalid_argument("MLLengths: vft_comm host buffers too small for a batch of lanes");
                    std::memcpy(comm->h_send, mine.data(), mine.size() * sizeof(REAL));
                    const char *rcv = gatherRecords(per, recB);
                    std::vector<int64_t> si, po, pa, pb, pla, plb;
                    std::vector<REAL> sv;
                    for (size_t t = 0; t < K; t++) {
                        if (t >= m0 && t < m1) continue;
                        REAL three[3];
                        std::memcpy(three, rcv + laneRecord(t, per) * recB, sizeof(three));
                        for (int j = 0; j < 3; j++) {
                            si.push_back(li[3 * t + (size_t) j]);
                            sv.push_back(three[j]);
                        }
                        po.push_back(rec[t]);
                        pa.push_back(ids[3 * t]);
                        pb.push_back(ids[3 * t + 1]);
                        pla.push_back(li[3 * t]);
                        plb.push_back(li[3 * t + 1]);
                    }
                    for (size_t f = 0; f < si.size(); f += 65535)
                        chk(vft_branch_lengths_scatter(ctx, (int64_t) std::min<size_t>(65535, si.size() - f), si.data() + f, sv.data() + f));
                    if (!po.empty()) chk(vft_posterior_profiles_blen(ctx, (int64_t) po.size(), po.data(), pa.data(), pb.data(), pla.data(), plb.data()));
                }
                laneSteps++;
                laneWork += (int64_t) rec.size();
            }
            optimizeRoundFrom(ftol, atol, upHave, done);
        }

        double partitionSpeedup = 0;           /* the reference's "theoretical speedup" of the last partition */
        int64_t laneSteps = 0, laneWork = 0;   /* lockstep steps / quartets or splits evaluated in them (all rounds so far) */

        int64_t nStarTests = 0;
        bool walkStepFused = true;   /* meSubmit: averages + distances as one step of the walk server (false: the two plain calls) */
        int64_t walkDualSent = 0, walkDualTaken = 0;   /* statistics: dual commands sent / continuations the walk server ran on its own */
        bool walkDual = true;        /* SPR chains hand both continuations of a step to the walk server (specContinuations); false: every step waits
                                        for the host's verdict (vft_nj_options.debug_flags & VFT_NJ_DEBUG_NO_WALK_DUAL) */
        bool walkServer = true;      /* the walks' steps go to resident workgroups through a mailbox (vft_walk_server_start); false: the
                                        two plain calls per step (tests compare) */
        int64_t sprSteps = 0;        /* chain steps evaluated by the SPR rounds */
        bool walkValueNumbers = true;   /* value numbers for the walks' rows: redundant averages and repeated quartets are not computed again
                                           (false: every average is recorded, every quartet evaluated; tests compare) */

        const std::vector<int64_t> &children() const { return child; }
        const std::vector<int64_t> &parents() const { return parent; }

        /* ---------------------------------------------------------------------------------------------- GTR
           setMLGtr (NJ.tcc:6436-6500): base frequencies from the leaves (with a pseudocount each), then two rounds
           (-mlacc >= 2: that many) over the six exchange rates, each a line search on the whole tree's likelihood
           (GTRNegLogLk, NJ.tcc:6407-6434: build the model, recomputeMLProfiles, treeLogLk), the rates normalised to
           gt = 1, the model installed, the profiles rebuilt and all branch lengths re-optimised.  On the device an
           evaluation is one vft_set_transition_matrix, one posterior batch per tree level and one pairLogLk batch. */
        struct GtrFit {
            double rates[6], freq[4];
        };

        GtrFit setMLGtr(const int64_t leafCodeCounts[4], int64_t nPos, int32_t mlAccuracy, double ftol, double atol, int threads = 1) {
            GtrFit g;
            int64_t n[4], sum = 0;
            for (int i = 0; i < 4; i++) {
                n[i] = 1 + leafCodeCounts[i];
                sum += n[i];
            }
            for (int i = 0; i < 4; i++) g.freq[i] = (double) n[i] / (double) sum;
            for (int i = 0; i < 6; i++) g.rates[i] = 1.0;
            const int64_t nRounds = mlAccuracy < 2 ? 2 : mlAccuracy;
            for (int64_t r = 0; r < nRounds; r++)
                for (int iRate = 0; iRate < 6; iRate++) {
                    auto negLogLk = [&](double x) {
                        double rates[6];
                        for (int i = 0; i < 6; i++) rates[i] = g.rates[i];
                        rates[iRate] = x;
                        installGTR(rates, g.freq);
                        recomputeMLProfiles();
                        gtrEvaluations++;
                        return -treeLogLk(nPos, -1);
                    };
                    g.rates[iRate] = hostMinimise(negLogLk, 0.05, g.rates[iRate], 20.0, 0.001, 0.0001);
                }
            for (int i = 0; i < 5; i++) g.rates[i] /= g.rates[5];
            g.rates[5] = 1.0;
            installGTR(g.rates, g.freq);
            recomputeMLProfiles();
            if (threads > 1) optimizeRoundThreaded(ftol, atol, threads);
            else optimizeRound(ftol, atol);
            return g;
        }

        void installGTR(const double rates[6], const double freq[4]) {
            TransitionTables4 t;
            createGTR<REAL>(rates, freq, t);
            REAL stat[4], statinv[4], eval[4], cf[20], ei[16], eiT[16];
            for (int i = 0; i < 4; i++) {
                stat[i] = (REAL) t.stat[i];
                statinv[i] = (REAL) t.statinv[i];
                eval[i] = (REAL) t.eigenval[i];
                for (int j = 0; j < 4; j++) {
                    ei[4 * i + j] = (REAL) t.eigeninv[i][j];
                    eiT[4 * i + j] = (REAL) t.eigeninvT[i][j];
                }
            }
            for (int i = 0; i < 5; i++)
                for (int j = 0; j < 4; j++) cf[4 * i + j] = (REAL) t.codeFreq[i][j];
            chk(vft_set_transition_matrix(ctx, stat, statinv, eval, cf, ei, eiT));
        }

        int64_t gtrEvaluations = 0;

        /* onedimenmin + brent (NJ.tcc:7025-7178) for host-side objectives (the GTR rates) */
        template<typename F>
        static double hostMinimise(F &&f, double xmin, double xguess, double xmax, double ftol, double atol) {
            double lo, mid, hi;
            if (xguess == xmin) {
                lo = xmin;
                mid = 2.0 * xguess;
                hi = 10.0 * xguess;
            } else if (xguess <= 2.0 * xmin) {
                lo = xmin;
                mid = xguess;
                hi = 5.0 * xguess;
            } else {
                lo = 0.5 * xguess;
                mid = xguess;
                hi = 2.0 * xguess;
            }
            if (hi > xmax) hi = xmax;
            if (mid >= hi) mid = 0.5 * (lo + hi);
            double fLo = f(lo), fMid = f(mid), fHi = f(hi);
            while (fLo < fMid && lo > xmin) {
                lo = (lo + xmin) / 2.0;
                if (lo < 2.0 * xmin) lo = xmin;
                fLo = f(lo);
            }
            while (fHi < fMid && hi < xmax) {
                hi = (hi + xmax) / 2.0;
                if (hi > xmax * 0.95) hi = xmax;
                fHi = f(hi);
            }
            const double golden = 0.3819660, zeps = 1.0e-10;
            double a = std::min(lo, hi), b = std::max(lo, hi);
            double x = mid, fx = fMid, w, fw, v, fv;
            if (fLo < fHi) {
                w = lo; fw = fLo; v = hi; fv = fHi;
            } else {
                w = hi; fw = fHi; v = lo; fv = fLo;
            }
            double step = 0.0, prevStep = 0.0;
            for (int it = 0; it < 100; it++) {
                const double xm = 0.5 * (a + b);
                const double tol1 = ftol * std::fabs(x), tol2 = 2.0 * (tol1 + zeps);
                if (std::fabs(x - xm) <= (tol2 - 0.5 * (b - a)) || std::fabs(a - b) < atol) break;
                bool goldenStep = true;
                if (std::fabs(prevStep) > tol1) {
                    const double r = (x - w) * (fx - fv);
                    double q = (x - v) * (fx - fw);
                    double p = (x - v) * q - (x - w) * r;
                    q = 2.0 * (q - r);
                    if (q > 0.0) p = -p;
                    q = std::fabs(q);
                    const double before = prevStep;
                    prevStep = step;
                    if (!(std::fabs(p) >= std::fabs(0.5 * q * before) || p <= q * (a - x) || p >= q * (b - x))) {
                        step = p / q;
                        const double u = x + step;
                        if (u - a < tol2 || b - u < tol2) step = (xm - x) >= 0.0 ? std::fabs(tol1) : -std::fabs(tol1);
                        goldenStep = false;
                    }
                }
                if (goldenStep) {
                    prevStep = x >= xm ? a - x : b - x;
                    step = golden * prevStep;
                }
                const double u = std::fabs(step) >= tol1 ? x + step : x + (step >= 0.0 ? std::fabs(tol1) : -std::fabs(tol1));
                const double fu = f(u);
                if (fu <= fx) {
                    if (u >= x) a = x; else b = x;
                    v = w; w = x; x = u;
                    fv = fw; fw = fx; fx = fu;
                } else {
                    if (u < x) a = u; else b = u;
                    if (fu <= fw || w == x) {
                        v = w; w = u;
                        fv = fw; fw = fu;
                    } else if (fu <= fv || v == x || v == w) {
                        v = u;
                        fv = fu;
                    }
                }
            }
            return x;
        }

        int64_t evaluations() {
            int64_t n = 0;
            chk(vft_ml_eval_count(ctx, &n));
            return n;
        }

        int64_t splits() const { return (int64_t) order.size(); }

    private:
        static double logCorrect(double dist, bool scoredist) {   /* NJ.tcc:322-330 */
            const double maxscore = 3.0;
            if (!scoredist) dist = dist < 0.74 ? -0.75 * std::log(1.0 - dist * 4.0 / 3.0) : maxscore;
            else dist = dist < 0.99 ? -1.3 * std::log(1.0 - dist) : maxscore;
            return dist < maxscore ? dist : maxscore;
        }

        int64_t siblingOf(int64_t v) const {   /* NJ.tcc:1977-1990: not defined below the root */
            const int64_t p = parent[(size_t) v];
            return child[3 * p] == v ? child[3 * p + 1] : child[3 * p];
        }

        /* nodeABCD of setupABCD (NJ.tcc:1942-1975): children, then the sibling and the parent - or the root's two
           other children */
        void quartetNodes(int64_t node, int64_t q[4]) const {
            q[0] = child[3 * node];
            q[1] = child[3 * node + 1];
            const int64_t p = parent[(size_t) node];
            if (p == root) {
                int n = 2;
                for (int k = 0; k < 3; k++)
                    if (child[3 * root + k] != node) q[n++] = child[3 * root + k];
            } else {
                q[2] = siblingOf(node);
                q[3] = p;
            }
        }

        void replaceChild(int64_t par, int64_t oldChild, int64_t newChild) {   /* NJ.tcc:1929-1940 */
            put8(&parent[(size_t) newChild], par);
            for (int k = 0; k < 3; k++)
                if (child[3 * par + k] == oldChild) {
                    put8(&child[3 * par + k], newChild);
                    return;
                }
            throw std::logic_error("MLLengths::replaceChild: not a child");
        }

        /* ---- A TRANSACTION over the host state of a walk (round 6: both continuations of an SPR chain step are built before the step's
           own distances say which one it will be, see sprAttempt).  While `txn.on` every write of the functions a chain step runs
           through - the tree arrays (replaceChild), the up-profile flags (updateForNNI, ensureUpProfile), the rows' value numbers
           (setVer), the record of pending averages (queueAverage, schedule: kept as whole copies, a few dozen entries) and the
           walk's counters - is logged with its old and its new value.  txnEnd() puts everything back and hands the log over; txnRedo()
           applies a log forwards: the state the transaction had reached, without running it again.  NOT logged, on purpose: the
           value-number and memo tables and the next fresh number (facts about values: an entry made by a continuation that is never
           taken is still true, and a number is never handed out twice), the scratch stamps of schedule(). */
        struct PendOp {   /* a recorded average (see schedule()) */
            int64_t out, a, b;
        };
        struct WalkTxnLog {
            struct W8 { int64_t *p; int64_t oldV, newV; };
            struct WC { char *p; char oldV, newV; };
            struct WV { size_t row; uint64_t oldVer, newVer; uint32_t oldEp, newEp; };
            std::vector<W8> w8;
            std::vector<WC> wc;
            std::vector<WV> wv;
            std::vector<PendOp> pendOld, pendNew;
            int64_t cntOld[4], cntNew[4];   /* avgRedundant, avgQueued, avgDropped, stepsMemoised */
            void clear() {
                w8.clear();
                wc.clear();
                wv.clear();
            }
        };
        struct WalkTxn {
            bool on = false;
            WalkTxnLog *log = nullptr;
        } txn;
        void put8(int64_t *p, int64_t v) {
            if (txn.on) {
                typename WalkTxnLog::W8 e = {p, *p, v};
                txn.log->w8.push_back(e);
            }
            *p = v;
        }
        void putc(char *p, char v) {
            if (txn.on) {
                typename WalkTxnLog::WC e = {p, *p, v};
                txn.log->wc.push_back(e);
            }
            *p = v;
        }
        void txnBegin(WalkTxnLog &log) {
            if (txn.on || cur || !qOut.empty()) throw std::logic_error("MLLengths::txnBegin: not at a step boundary of a single walk");
            log.clear();
            log.pendOld = pend;
            log.cntOld[0] = avgRedundant;
            log.cntOld[1] = avgQueued;
            log.cntOld[2] = avgDropped;
            log.cntOld[3] = stepsMemoised;
            txn.log = &log;
            txn.on = true;
        }
        void txnEnd() {   /* roll back; the log keeps both directions */
            WalkTxnLog &log = *txn.log;
            txn.on = false;
            log.pendNew.swap(pend);
            pend = log.pendOld;
            log.cntNew[0] = avgRedundant;
            log.cntNew[1] = avgQueued;
            log.cntNew[2] = avgDropped;
            log.cntNew[3] = stepsMemoised;
            avgRedundant = log.cntOld[0];
            avgQueued = log.cntOld[1];
            avgDropped = log.cntOld[2];
            stepsMemoised = log.cntOld[3];
            for (size_t k = log.wv.size(); k-- > 0;) {
                rowVer[log.wv[k].row] = log.wv[k].oldVer;
                rowVerEpoch[log.wv[k].row] = log.wv[k].oldEp;
            }
            for (size_t k = log.wc.size(); k-- > 0;) *log.wc[k].p = log.wc[k].oldV;
            for (size_t k = log.w8.size(); k-- > 0;) *log.w8[k].p = log.w8[k].oldV;
            qOut.clear();
            qA.clear();
            qB.clear();
        }
        void txnRedo(const WalkTxnLog &log) {
            for (const typename WalkTxnLog::W8 &e: log.w8) *e.p = e.newV;
            for (const typename WalkTxnLog::WC &e: log.wc) *e.p = e.newV;
            for (const typename WalkTxnLog::WV &e: log.wv) {
                rowVer[e.row] = e.newVer;
                rowVerEpoch[e.row] = e.newEp;
            }
            pend = log.pendNew;
            avgRedundant += log.cntNew[0] - log.cntOld[0];
            avgQueued += log.cntNew[1] - log.cntOld[1];
            avgDropped += log.cntNew[2] - log.cntOld[2];
            stepsMemoised += log.cntNew[3] - log.cntOld[3];
        }

        /* updateForNNI, fast flavour (NJ.tcc:1902-1926): drop the up-profiles around the rearranged node, refresh its
           profile and its parent's */
        void updateForNNI(int64_t node, bool useML, std::vector<char> &upHave) {
            putc(&upHave[(size_t) node], 0);
            for (int k = 0; k < 2; k++) putc(&upHave[(size_t) child[3 * node + k]], 0);
            const int64_t ip = parent[(size_t) node];
            if (ip == root) {
                for (int k = 0; k < 3; k++)
                    if (child[3 * root + k] != node) putc(&upHave[(size_t) child[3 * root + k]], 0);
            } else {
                putc(&upHave[(size_t) ip], 0);
                putc(&upHave[(size_t) siblingOf(node)], 0);
            }
            if (ip != root && parent[(size_t) ip] != root) putc(&upHave[(size_t) siblingOf(ip)], 0);   /* the uncle */
            recomputeProfile(node, useML);
            recomputeProfile(ip, useML);
        }

        /* the two nodes a subtree can be moved around: its parent and its sibling, or its two siblings below the root */
        void movePivots(int64_t node, int64_t out[2]) const {
            const int64_t p = parent[(size_t) node];
            if (p == root) {
                int n = 0;
                for (int k = 0; k < 3; k++)
                    if (child[3 * root + k] != node) out[n++] = child[3 * root + k];
            } else {
                out[0] = p;
                out[1] = siblingOf(node);
            }
        }

        /* setupABCD + chooseNNI's criteria (NJ.tcc:4836-4846; lower is better): log-corrected distances AB+CD, AC+BD,
           AD+BC over the four profiles around `node`.  In two halves: meSubmit queues the up-profile of the quartet and hands the
           step - the averages queued since the last one + the six distances - to the device; meCollect waits for the distances.
           With the walk server up a caller that does not need a step's criteria to build the next step (sprAttempt: the first
           NNI of a chain is forced) keeps two steps in flight; otherwise meSubmit has waited already. */
        struct MeTicket {
            uint32_t ticket = 0;
            bool pending = false;
            bool keyed = false;     /* key = the value numbers of the quartet's rows: the distances go to the memo table */
            uint64_t key[4];
            REAL d[6];
        };
        /* ---- Both continuations of a chain step (round 6; csrc/vft_kernels_walk.h "DUAL command").  Step k of a chain (k >= 1) is on
           the device; which NNI it leads to - swap B and C, or A and C (findSPRSteps, NJ.tcc:1805-1859) - is one comparison of its own
           distances.  For either outcome: inside a transaction, make the swap (replaceChild x 2, updateForNNI), find the next pivot
           and, if the chain goes on there, build step k + 1 (meBuild: up-profile, memo table, the averages its rows depend on);
           keep the command and the transaction's log, put the state back.  The two commands go down as ONE dual command behind step
           k; the workgroups compare step k's distances themselves and run their alternative at once - the ~2.4 us in which they
           used to wait for the host's verdict are gone for every step whose continuation is a plain device step.  A continuation
           that is not one (the chain ends, the quartet is in the memo table, too many averages for one command) is sent as "not a
           device step"; taken, it costs an empty command and the host goes on as before.  When the step's answer arrives the host
           makes the comparison too, applies the log of the continuation taken (txnRedo) and checks the workgroups' choice against
           its own (vft_walk_dual_choice). */
        struct SprAlt {
            bool valid = false;
            int alt = 0;
            int64_t aroundNext = -1;
            int64_t q[4], q4[4];
            MeTicket t;
            uint32_t ticket = 0;
            std::vector<int64_t> out, a, b;
            WalkTxnLog log;
        };
        SprAlt specAlt[2];
        bool specContinuations(int64_t node, int64_t around, const int64_t q[4], int chainLength, int maxSPRLength, bool scoredist, std::vector<char> &upHave) {
            for (int alt = 0; alt < 2; alt++) {
                SprAlt &A = specAlt[alt];
                A.valid = false;
                A.out.clear();
                A.a.clear();
                A.b.clear();
                const int64_t n0 = alt == 0 ? q[1] : q[0], n1 = q[2];   /* alternative 0: swap B and C; 1: swap A and C */
                txnBegin(A.log);
                replaceChild(around, n0, n1);
                replaceChild(parent[(size_t) around], n1, n0);
                updateForNNI(around, false, upHave);
                int64_t next[2];
                movePivots(node, next);
                A.aroundNext = next[next[0] == around ? 1 : 0];
                if (chainLength + 1 < maxSPRLength && A.aroundNext >= nSeqs && A.aroundNext != root) {
                    if (meBuild(A.aroundNext, upHave, A.q, A.t, A.q4, true) && qOut.size() <= (size_t) VFT_WALK_DUAL_MAX_AVERAGES) {
                        A.valid = true;
                        A.out = qOut;
                        A.a = qA;
                        A.b = qB;
                    }
                }
                txnEnd();
            }
            if (specAlt[0].valid && specAlt[1].valid && specAlt[0].out.size() + specAlt[1].out.size() > (size_t) VFT_WALK_DUAL_MAX_AVERAGES)
                specAlt[specAlt[0].out.size() > specAlt[1].out.size() ? 0 : 1].valid = false;   /* (both do not fit one command: the shorter one) */
            if (!specAlt[0].valid && !specAlt[1].valid) return false;
            uint32_t ticket = 0;
            const SprAlt &A0 = specAlt[0], &A1 = specAlt[1];
            chk(vft_walk_submit_dual(ctx, (int32_t) A0.out.size(), A0.out.data(), A0.a.data(), A0.b.data(), A0.valid ? A0.q4 : nullptr,
                                     (int32_t) A1.out.size(), A1.out.data(), A1.a.data(), A1.b.data(), A1.valid ? A1.q4 : nullptr, scoredist ? 1 : 0, &ticket));
            specAlt[0].ticket = specAlt[1].ticket = ticket;
            walkDualSent++;
            return true;
        }

        /* the host half of a step up to the hand-over: the quartet, its up-profile, the memo table, the averages the quartet's rows
           depend on (left in qOut / qA / qB).  False: the distances came from the memo table, nothing goes to the device.  spec:
           inside a transaction - nothing may be sent: a step whose averages do not fit a command leaves qOut longer than that and
           the caller gives the continuation up. */
        bool meBuild(int64_t node, std::vector<char> &upHave, int64_t q[4], MeTicket &t, int64_t q4[4], bool spec) {
            quartetNodes(node, q);
            const int64_t par = parent[(size_t) node];
            int64_t idD = q[3];
            if (par != root) {
                ensureUpProfile(par, false, upHave);
                idD = par + nSeqs;
            }
            q4[0] = q[0];
            q4[1] = q[1];
            q4[2] = q[2];
            q4[3] = idD;
            t.pending = false;
            t.keyed = false;
            if (vnActive) {   /* the same four values as an earlier quartet: its distances (see vnBegin) */
                for (int i = 0; i < 4; i++) t.key[i] = verOf(q4[i]);
                t.keyed = true;
                const MemoEntry &m = memoTable[(size_t) (vnHash(vnHash(t.key[0], t.key[1]), vnHash(t.key[2], t.key[3])) & (memoTable.size() - 1))];
                if (m.used && m.q[0] == t.key[0] && m.q[1] == t.key[1] && m.q[2] == t.key[2] && m.q[3] == t.key[3]) {
                    for (int i = 0; i < 6; i++) t.d[i] = m.d[i];
                    t.keyed = false;
                    stepsMemoised++;
                    return false;
                }
            }
            schedule(q4, 4, false);   /* the recorded averages this quartet's rows depend on (see queueAverage) */
            if (!spec) sendLongHead(48);
            return true;
        }
        void meSubmit(int64_t node, std::vector<char> &upHave, int64_t q[4], MeTicket &t) {
            int64_t q4[4];
            if (!meBuild(node, upHave, q, t, q4, false)) return;
            const int64_t idD = q4[3];
            if (serverUp) {
                chk(vft_walk_submit(ctx, (int32_t) qOut.size(), qOut.data(), qA.data(), qB.data(), q4, &t.ticket));
                qOut.clear();
                qA.clear();
                qB.clear();
                t.pending = true;
                return;
            }
            const int64_t pi[6] = {q[0], q[0], q[0], q[1], q[1], q[2]}, pj[6] = {q[1], q[2], idD, q[2], idD, idD};
            REAL w[6];
            /* the queued averages and the six distances as one launch (vft_walk_step) while every profile is a plain row */
            bool fused = false;
            if (walkStepFused) {
                const int rc = vft_walk_step(ctx, (int32_t) qOut.size(), qOut.data(), qA.data(), qB.data(), q4, t.d);
                if (rc == VFT_OK) {
                    fused = true;
                    qOut.clear();
                    qA.clear();
                    qB.clear();
                } else if (rc != VFT_ERR_STATE) {
                    chk(rc);   /* (a HIP error, a bad argument, a timeout: not a reason to go on another way) */
                }
                /* VFT_ERR_STATE: no walk server - or it cannot take this step -: the two plain calls for THIS step; the next guard tries
                   to start the server again (one refused step does not switch it off for the rest of the tree) */
            }
            if (!fused) {
                flushAverages();
                chk(vft_profile_distances(ctx, 6, pi, pj, t.d, w));
            }
        }
        void meCollect(MeTicket &t, bool scoredist, double criteria[3]) {
            if (t.pending) {
                if (walkStats) {
                    const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                    chk(vft_walk_collect(ctx, t.ticket, t.d));
                    walkWaitSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    walkDeviceSteps++;
                } else
                chk(vft_walk_collect(ctx, t.ticket, t.d));
                t.pending = false;
            }
            if (t.keyed) {
                MemoEntry &m = memoTable[(size_t) (vnHash(vnHash(t.key[0], t.key[1]), vnHash(t.key[2], t.key[3])) & (memoTable.size() - 1))];
                for (int i = 0; i < 4; i++) m.q[i] = t.key[i];
                for (int i = 0; i < 6; i++) m.d[i] = t.d[i];
                m.used = true;
                t.keyed = false;
            }
            double c[6];
            for (int i = 0; i < 6; i++) c[i] = logCorrect((double) t.d[i], scoredist);
            criteria[0] = c[0] + c[5];
            criteria[1] = c[1] + c[4];
            criteria[2] = c[2] + c[3];
        }
        void meCriteria(int64_t node, bool scoredist, std::vector<char> &upHave, int64_t q[4], double criteria[3]) {
            MeTicket t;
            meSubmit(node, upHave, q, t);
            meCollect(t, scoredist, criteria);
        }

        /* traversePostorder (NJ.tcc:3343-3380) on the tree as it is now */
        int64_t nextPostorder(int64_t node, std::vector<char> &traversal, bool *up, int64_t branchRoot, bool reportUp = true) const {
            *up = false;
            for (;;) {
                bool found = false;
                for (int k = 0; k < 3 && child[3 * node + k] >= 0; k++) {
                    const int64_t c = child[3 * node + k];
                    if (!traversal[(size_t) c]) {
                        node = c;
                        found = true;
                        break;
                    }
                }
                if (found) continue;
                if (!traversal[(size_t) node]) {
                    traversal[(size_t) node] = 1;
                    return node;
                }
                if (node == branchRoot) return -1;
                node = parent[(size_t) node];
                if (reportUp && traversal[(size_t) node]) {
                    *up = true;
                    return node;
                }
            }
        }

        /* getUpProfile (NJ.tcc:3382-3434): cached; missing ones are built from the root down */
        void ensureUpProfile(int64_t node, bool useML, std::vector<char> &upHave) {
            if (upHave[(size_t) node]) return;
            std::vector<int64_t> &path = upPathScratch;   /* (a member: this runs three times per SPR chain step) */
            path.clear();
            for (int64_t x = node; x != root; x = parent[(size_t) x]) path.push_back(x);
            /* a lane: the up-profiles of its subtree root and above do not change during the parallel phase and belong to
               every lane below them - they are built once, level by level, before the lanes' chains (runShared) */
            size_t tR = path.size();
            if (cur && cur->R >= 0)
                for (size_t t = 0; t < path.size(); t++)
                    if (path[t] == cur->R) tR = t;
            for (size_t t = path.size(); t-- > 0;) {
                const int64_t x = path[t];
                if (upHave[(size_t) x]) continue;
                int64_t cd[2], lcd[2];
                quartetCD(x, cd, lcd);
                const int64_t out = x + nSeqs;
                if (t >= tR) {
                    SharedOp op = {(int64_t) (path.size() - t), out, cd[0], cd[1], lcd[0], lcd[1]};
                    sharedOps.push_back(op);
                } else if (useML) queuePosterior(out, cd[0], cd[1], lcd[0], lcd[1]);
                else queueAverage(out, cd[0], cd[1]);
                putc(&upHave[(size_t) x], 1);
            }
        }

        void runShared(bool useML) {
            if (sharedOps.empty()) return;
            std::stable_sort(sharedOps.begin(), sharedOps.end(), [](const SharedOp &x, const SharedOp &y) { return x.depth < y.depth; });
            std::vector<int64_t> o, a, b, la, lb;
            for (size_t i = 0; i < sharedOps.size();) {
                size_t e = i;
                o.clear(); a.clear(); b.clear(); la.clear(); lb.clear();
                for (; e < sharedOps.size() && sharedOps[e].depth == sharedOps[i].depth; e++) {
                    o.push_back(sharedOps[e].out);
                    a.push_back(sharedOps[e].a);
                    b.push_back(sharedOps[e].b);
                    la.push_back(sharedOps[e].la);
                    lb.push_back(sharedOps[e].lb);
                }
                if (useML) chk(vft_posterior_profiles_blen(ctx, (int64_t) o.size(), o.data(), a.data(), b.data(), la.data(), lb.data()));
                else chk(vft_average_profiles(ctx, (int64_t) o.size(), o.data(), a.data(), b.data(), nullptr));
                i = e;
            }
            sharedOps.clear();
        }

        /* the chains of all lanes as one launch (several when there are more than 4096 ops); a lane with more than 256 ops -
           a deep first descent - sends the head of its chain ahead on its own */
        void runChains(std::vector<Lane> &lanes, bool useML) {
            std::vector<int64_t> o, a, b, la, lb;
            std::vector<int32_t> off(1, 0);
            auto flush = [&]() {
                if (off.size() > 1) {
                    if (useML) chk(vft_posterior_chains_blen(ctx, (int32_t) off.size() - 1, off.data(), o.data(), a.data(), b.data(), la.data(), lb.data()));
                    else chk(vft_average_chains(ctx, (int32_t) off.size() - 1, off.data(), o.data(), a.data(), b.data()));
                }
                o.clear(); a.clear(); b.clear(); la.clear(); lb.clear();
                off.assign(1, 0);
            };
            for (Lane &ln: lanes) {
                size_t first = 0;
                const size_t n = ln.out.size();
                while (n - first > 256) {
                    if (useML) chk(vft_posterior_chain_blen(ctx, 256, &ln.out[first], &ln.a[first], &ln.b[first], &ln.la[first], &ln.lb[first]));
                    else chk(vft_average_chain(ctx, 256, &ln.out[first], &ln.a[first], &ln.b[first]));
                    first += 256;
                }
                if (n == first) {
                    if (n) { ln.out.clear(); ln.a.clear(); ln.b.clear(); ln.la.clear(); ln.lb.clear(); }
                    continue;
                }
                if (o.size() + (n - first) > 4096 || off.size() > 60000) flush();
                o.insert(o.end(), ln.out.begin() + (long) first, ln.out.end());
                a.insert(a.end(), ln.a.begin() + (long) first, ln.a.end());
                b.insert(b.end(), ln.b.begin() + (long) first, ln.b.end());
                la.insert(la.end(), ln.la.begin() + (long) first, ln.la.end());
                lb.insert(lb.end(), ln.lb.begin() + (long) first, ln.lb.end());
                off.push_back((int32_t) o.size());
                ln.out.clear(); ln.a.clear(); ln.b.clear(); ln.la.clear(); ln.lb.clear();
            }
            flush();
        }

        /* recomputeProfile (NJ.tcc:3436-3473) without BIONJ weighting */
        void recomputeProfile(int64_t node, bool useML) {
            if (node < nSeqs || node == root) return;
            const int64_t a = child[3 * node], b = child[3 * node + 1];
            if (useML) queuePosterior(node, a, b, a, b);
            else queueAverage(node, a, b);
        }

        /* the same for the posteriors of the ML walk (vft_posterior_chain_blen); the branch lengths are read on the
           device when the chain runs, i.e. after the verdict kernels queued before it */
        void queuePosterior(int64_t out, int64_t a, int64_t b, int64_t la, int64_t lb) {
            if (cur) {   /* a lane of the subtree schedule: its own chain, launched with the other lanes' */
                cur->out.push_back(out);
                cur->a.push_back(a);
                cur->b.push_back(b);
                cur->la.push_back(la);
                cur->lb.push_back(lb);
                return;
            }
            pOut.push_back(out);
            pA.push_back(a);
            pB.push_back(b);
            pLa.push_back(la);
            pLb.push_back(lb);
            if (pOut.size() >= 128) flushPosteriors();
        }

        void flushPosteriors() {
            if (pOut.empty()) return;
            chk(vft_posterior_chain_blen(ctx, (int32_t) pOut.size(), pOut.data(), pA.data(), pB.data(), pLa.data(), pLb.data()));
            pOut.clear();
            pA.clear();
            pB.clear();
            pLa.clear();
            pLb.clear();
        }

        std::vector<int64_t> pOut, pA, pB, pLa, pLb;
        std::vector<int64_t> upPathScratch;

        /* Minimum-evolution averages are queued and go down as one chain launch (vft_average_chain) right before
           something reads profiles: a step of an NNI / SPR walk is then two launches (chain, distances) and one wait */
        void queueAverage(int64_t out, int64_t a, int64_t b) {
            if (cur) {
                cur->out.push_back(out);
                cur->a.push_back(a);
                cur->b.push_back(b);
                cur->la.push_back(a);
                cur->lb.push_back(b);
                return;
            }
            if (vnActive) {
                const uint64_t va = verOf(a), vb = verOf(b);
                VnEntry &e = vnTable[(size_t) (vnHash(va, vb) & (vnTable.size() - 1))];
                if (!(e.v != 0 && e.a == va && e.b == vb)) {
                    e.a = va;
                    e.b = vb;
                    e.v = vnNext++;
                }
                if (verOf(out) == e.v) {   /* the row holds this very average already */
                    avgRedundant++;
                    return;
                }
                setVer(out, e.v);
            }
            PendOp op = {out, a, b};
            pend.push_back(op);
        }

        /* VALUE NUMBERS for the rows of a host-driven walk.  Every row carries the number of the value it (logically) holds; an
           average of two numbered values gets the number that the same two values got the last time they were averaged (a small
           direct-mapped table; a miss hands out a fresh number, which is always safe).  Two consequences, both exact because the
           device arithmetic is deterministic - equal numbers are equal bits:
             * an average that would write the value its row already holds is not recorded at all - the unwinding of a rejected SPR
               chain puts the tree back and recomputes the profiles around every undone step from the same children (NJ.tcc:1861-1879);
             * a quartet whose four rows carry the numbers of an earlier evaluation takes that evaluation's six distances - the second
               chain around a node (acFirst = 1) starts from the quartet the first one (acFirst = 0) started from, NJ.tcc:6240-6270 -
               and no step goes to the device.
           Numbers live for one walk (a WalkServerGuard): rows written by anything else have no number. */
        struct VnEntry {
            uint64_t a, b, v;
        };
        struct MemoEntry {
            uint64_t q[4];
            REAL d[6];
            bool used;
        };
        std::vector<uint64_t> rowVer;
        std::vector<uint32_t> rowVerEpoch;
        std::vector<VnEntry> vnTable;
        std::vector<MemoEntry> memoTable;
        uint32_t vnEpoch = 0;
        uint64_t vnNext = 0;
        bool vnActive = false;
        int64_t avgRedundant = 0, stepsMemoised = 0;   /* statistics */
        double walkWaitSeconds = 0, walkSeconds = 0;    /* (VFT_WALK_STATS) waiting for the device's answers / the walks in all */
        double walkSpecSeconds = 0;                     /* ... building both continuations of chain steps (specContinuations) */
        int64_t walkDeviceSteps = 0;
        bool walkStats = std::getenv("VFT_WALK_STATS") != nullptr;
        static uint64_t vnHash(uint64_t x, uint64_t y) {
            uint64_t h = x * 0x9E3779B97F4A7C15ull ^ (y + 0x7F4A7C159E3779B9ull + (x << 6) + (x >> 2));
            h ^= h >> 29;
            h *= 0xBF58476D1CE4E5B9ull;
            h ^= h >> 32;
            return h;
        }
        void vnBegin() {
            const size_t nIds = (size_t) (nNodes + nSeqs + 1);
            if (rowVer.size() < nIds) {
                rowVer.assign(nIds, 0);
                rowVerEpoch.assign(nIds, 0);
                vnTable.assign((size_t) 1 << 16, VnEntry{0, 0, 0});
                memoTable.assign((size_t) 1 << 15, MemoEntry{{0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}, false});
                vnEpoch = 0;
            }
            if (++vnEpoch == 0) {
                std::fill(rowVerEpoch.begin(), rowVerEpoch.end(), 0);
                vnEpoch = 1;
            }
            for (VnEntry &e: vnTable) e.v = 0;
            for (MemoEntry &e: memoTable) e.used = false;
            vnNext = (uint64_t) nIds + 1;   /* numbers 1 ... nIds are the rows' own (the value a row held when the walk began) */
            vnActive = true;
        }
        uint64_t verOf(int64_t row) const { return rowVerEpoch[(size_t) row] == vnEpoch ? rowVer[(size_t) row] : (uint64_t) row + 1; }
        void setVer(int64_t row, uint64_t v) {
            if (txn.on) {
                typename WalkTxnLog::WV e = {(size_t) row, rowVer[(size_t) row], v, rowVerEpoch[(size_t) row], vnEpoch};
                txn.log->wv.push_back(e);
            }
            rowVer[(size_t) row] = v;
            rowVerEpoch[(size_t) row] = vnEpoch;
        }

        /* The averages of a host-driven walk are evaluated LAZILY.  The reference recomputes profiles eagerly after every
           rearrangement (updateForNNI, the unwinding of a rejected SPR chain, NJ.tcc:1902-1926), and a third of those averages
           are overwritten before anything reads them (measured on 10 000 x 200: 34 %): on the CPU a cheap habit, here every one is
           a link of the dependent chain a step waits for.  queueAverage only records an average; schedule() - called when a
           step needs rows (the quartet of meSubmit), or everything (flushAverages) - walks the record backwards and picks
             needed  its output is one of the wanted rows, or an input of a picked average (and no later picked one rewrites it),
             forced  it reads a row that a picked LATER average writes (it must see the old value), or it is older than the
                     window of `pendKeep` averages the record is allowed to hold,
             dead    its output is rewritten by a picked later average and nothing picked reads it in between: dropped for good;
           everything else stays recorded.  Picked averages run in program order, so every value that is ever read - by a
           distance, by a later average, by whoever looks at the rows after the walk - is the value the eager order computes:
           the same operations on the same operands, fewer of them. */
        std::vector<PendOp> pend;
        std::vector<uint32_t> needStamp, writeStamp;
        std::vector<unsigned char> pendSel;
        uint32_t pendEpoch = 0;
        size_t pendKeep = 48;
        int64_t avgQueued = 0, avgDropped = 0;   /* statistics: averages recorded / never run */

        void schedule(const int64_t *rows, int nRows, bool all) {
            if (pend.empty()) return;
            const size_t nIds = (size_t) (nNodes + nSeqs + 1);
            if (needStamp.size() < nIds) {
                needStamp.assign(nIds, 0);
                writeStamp.assign(nIds, 0);
                pendEpoch = 0;
            }
            if (++pendEpoch == 0) {
                std::fill(needStamp.begin(), needStamp.end(), 0);
                std::fill(writeStamp.begin(), writeStamp.end(), 0);
                pendEpoch = 1;
            }
            const uint32_t ep = pendEpoch;
            for (int i = 0; i < nRows; i++) needStamp[(size_t) rows[i]] = ep;
            const size_t n = pend.size(), forcedBelow = n > pendKeep ? n - pendKeep : 0;
            pendSel.assign(n, 0);
            for (size_t k = n; k-- > 0;) {
                const PendOp &op = pend[k];
                unsigned char sel = 0;
                if (needStamp[(size_t) op.out] == ep) sel = 1;
                else if (writeStamp[(size_t) op.out] == ep) sel = 2;   /* rewritten by a picked later average, unread: dead */
                else if (all || k < forcedBelow || writeStamp[(size_t) op.a] == ep || writeStamp[(size_t) op.b] == ep) sel = 1;
                pendSel[k] = sel;
                if (sel == 1) {
                    needStamp[(size_t) op.out] = 0;
                    writeStamp[(size_t) op.out] = ep;
                    needStamp[(size_t) op.a] = ep;
                    needStamp[(size_t) op.b] = ep;
                }
            }
            size_t kept = 0;
            for (size_t k = 0; k < n; k++) {
                if (pendSel[k] == 1) {
                    qOut.push_back(pend[k].out);
                    qA.push_back(pend[k].a);
                    qB.push_back(pend[k].b);
                } else if (pendSel[k] == 0) {
                    pend[kept++] = pend[k];
                } else {
                    avgDropped++;
                }
            }
            avgQueued += (int64_t) (n - kept);
            pend.resize(kept);
        }

        /* what schedule() picked beyond one call's worth goes down ahead as averages alone (after an accepted SPR move every
           ancestor is re-averaged: thousands in a deep tree) */
        void sendLongHead(size_t leave) {
            while (qOut.size() > leave) {
                const size_t m = qOut.size() - leave < 128 ? qOut.size() - leave : 128;
                if (serverUp) {
                    uint32_t ticket;
                    chk(vft_walk_submit(ctx, (int32_t) m, qOut.data(), qA.data(), qB.data(), nullptr, &ticket));
                } else {
                    chk(vft_average_chain(ctx, (int32_t) m, qOut.data(), qA.data(), qB.data()));
                }
                qOut.erase(qOut.begin(), qOut.begin() + (long) m);
                qA.erase(qA.begin(), qA.begin() + (long) m);
                qB.erase(qB.begin(), qB.begin() + (long) m);
            }
        }

        /* The walk server around a host-driven walk (doSPR, the one-thread minimum-evolution NNIs): up for the lifetime of the
           guard when the context can run it (every profile a row, the alignment fits its staging), otherwise the walk keeps its
           launch per step.  finish() sends what is still queued and retires the server; the destructor only retires it (an
           exception is on its way). */
        struct WalkServerGuard {
            MLLengths &t;
            std::chrono::steady_clock::time_point born;
            explicit WalkServerGuard(MLLengths &tree, bool enable = true) : t(tree) {
                born = std::chrono::steady_clock::now();
                if (!enable) return;
                t.flushAverages();
                if (t.walkValueNumbers) t.vnBegin();
                if (!t.walkServer || !t.walkStepFused || t.serverUp) return;
                const int rc = vft_walk_server_start(t.ctx);
                if (rc == VFT_OK) t.serverUp = true;
                else if (rc != VFT_ERR_STATE) t.chk(rc);
            }
            void finish() {
                t.flushAverages();
                t.vnActive = false;
                if (t.walkStats) {
                    t.walkSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - born).count();
                    fprintf(stderr, "walk: %lld averages run, %lld dropped unread, %lld redundant (same value), %lld steps from the memo table; %lld device steps, "
                            "%.3f s waiting for their answers, %.3f s building continuations (%lld dual commands, %lld taken), %.3f s of walks in all\n",
                            (long long) (t.avgQueued - t.avgDropped), (long long) t.avgDropped, (long long) t.avgRedundant, (long long) t.stepsMemoised,
                            (long long) t.walkDeviceSteps, t.walkWaitSeconds, t.walkSpecSeconds, (long long) t.walkDualSent, (long long) t.walkDualTaken, t.walkSeconds);
                }
                if (t.serverUp) {
                    t.serverUp = false;
                    t.chk(vft_walk_server_stop(t.ctx));
                }
            }
            ~WalkServerGuard() {
                t.vnActive = false;
                if (t.serverUp) {
                    t.serverUp = false;
                    (void) vft_walk_server_stop(t.ctx);
                }
            }
        };
        bool serverUp = false;

        void flushAverages() {
            schedule(nullptr, 0, true);   /* everything recorded, minus what is dead by now */
            sendLongHead(0);
        }

        std::vector<int64_t> qOut, qA, qB;

        void rebuildOrder() {
            order.clear();
            std::vector<std::pair<int64_t, int>> stack(1, std::make_pair(root, 0));
            while (!stack.empty()) {
                const int64_t v = stack.back().first;
                const int k = stack.back().second;
                if (k < 3 && child[3 * v + k] >= 0) {
                    stack.back().second++;
                    stack.push_back(std::make_pair(child[3 * v + k], 0));
                } else {
                    stack.pop_back();
                    if (child[3 * v] >= 0) order.push_back(v);
                }
            }
        }

        void chk(int rc) {
            if (rc != VFT_OK) throw std::runtime_error(std::string("MLLengths: ") + vft_last_error(ctx));
        }

        vft_ctx *ctx;
        int64_t nSeqs, nNodes, root;
        std::vector<int64_t> parent, child, order;
    };

}

#endif
