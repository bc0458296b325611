/*
 * TEST INFRASTRUCTURE — white-box oracle harness (never shipped, never built on the GPU box).
 *
 * Compiles against the reference headers WHERE THEY LIE (/root/reference/src, include path given by
 * oracle/Makefile) and calls the reference's own private member functions of
 * NeighbourJoining<Precision,Operations> to dump golden input/output vectors for the hot path
 * (SURVEY.md §8a rows a2..a13).  No reference source is copied: everything numeric below is produced
 * by calling the reference; the only logic that lives here is choosing which inputs to call it with
 * and serialising the answers.
 *
 * Output container ("VFX1"): a sequence of records
 *     u32 name_len | name | u8 dtype ('f' f32,'d' f64,'i' i32,'q' i64,'b' u8) | u32 ndim | u64 dims[] | data
 * read by oracle/gen_fixtures.py and re-saved as tests/golden/*.npz.
 */
#include <sstream>
#include <fstream>
#include <iostream>
#include <memory>
#include <mutex>
#include <list>
#include <vector>
#include <string>
/* private members are reached with g++ -fno-access-control (see oracle/Makefile), not by macro tricks:
   "#define private public" would break the reference's "#pragma omp ... private(x)" clauses */
#include "Utils.h"
#include "operations/BasicOperations.h"
#include "operations/SSE128Operations.h"
#include "operations/AVX256Operations.h"
#include "NeighbourJoining.h"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "Knuth.h"

using namespace veryfasttree;

/* ---------------------------------------------------------------- container writer */
struct Dump {
    FILE *fp;

    explicit Dump(const char *path) : fp(fopen(path, "wb")) {
        if (!fp) {
            perror(path);
            exit(2);
        }
        fwrite("VFX1", 1, 4, fp);
    }

    ~Dump() { fclose(fp); }

    void raw(const std::string &name, char dtype, const std::vector<uint64_t> &dims, const void *data, size_t bytes) {
        uint32_t nl = (uint32_t) name.size();
        fwrite(&nl, 4, 1, fp);
        fwrite(name.data(), 1, nl, fp);
        fwrite(&dtype, 1, 1, fp);
        uint32_t nd = (uint32_t) dims.size();
        fwrite(&nd, 4, 1, fp);
        fwrite(dims.data(), 8, nd, fp);
        fwrite(data, 1, bytes, fp);
    }

    static char code(float) { return 'f'; }
    static char code(double) { return 'd'; }
    static char code(int32_t) { return 'i'; }
    static char code(int64_t) { return 'q'; }
    static char code(uint8_t) { return 'b'; }

    template<typename T>
    void vec(const std::string &name, const std::vector<T> &v) {
        raw(name, code(T()), {(uint64_t) v.size()}, v.data(), v.size() * sizeof(T));
    }

    template<typename T>
    void mat(const std::string &name, const std::vector<T> &v, uint64_t rows, uint64_t cols) {
        if (v.size() != rows * cols) {
            fprintf(stderr, "shape mismatch for %s\n", name.c_str());
            exit(2);
        }
        raw(name, code(T()), {rows, cols}, v.data(), v.size() * sizeof(T));
    }

    template<typename T>
    void scalar(const std::string &name, T v) {
        raw(name, code(T()), {}, &v, sizeof(T));
    }
};

/* deterministic PRNG for choosing test inputs (splitmix64) */
struct Rng {
    uint64_t s;

    explicit Rng(uint64_t seed) : s(seed) {}

    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }

    int64_t below(int64_t n) { return (int64_t) (next() % (uint64_t) n); }

    double unit() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
};

static uint64_t fnv(uint64_t h, const void *p, size_t n) {
    const unsigned char *c = (const unsigned char *) p;
    for (size_t i = 0; i < n; i++) {
        h ^= c[i];
        h *= 0x100000001B3ull;
    }
    return h;
}

/* ---------------------------------------------------------------- the harness proper */
template<typename P, template<class> class Op>
struct Harness {
    typedef NeighbourJoining<P, Op> NJ;
    typedef typename NJ::Profile Profile;
    typedef typename NJ::Besthit Besthit;
    typedef Op<P> op_t;

    NJ &nj;
    Dump &out;
    int nCodes;
    int64_t nPos, nSeqs;

    Harness(NJ &nj, Dump &out) : nj(nj), out(out), nCodes(nj.options.nCodes), nPos(nj.nPos), nSeqs(nj.nSeqs) {}

    /* dense view of a reference profile: weights[nPos], codes[nPos], freq[nPos*nCodes] (0 where no vector) */
    void dense(Profile &p, std::vector<P> &w, std::vector<uint8_t> &c, std::vector<P> &f) {
        w.assign(p.weights, p.weights + nPos);
        c.resize(nPos);
        for (int64_t i = 0; i < nPos; i++) c[i] = (uint8_t) p.codes[i];
        f.assign((size_t) nPos * nCodes, (P) 0);
        int64_t iv = 0;
        for (int64_t i = 0; i < nPos; i++) {
            if (p.weights[i] > 0 && p.codes[i] == NOCODE) {
                for (int k = 0; k < nCodes; k++) f[i * nCodes + k] = p.vectors[nj.nCodeSize * iv + k];
                iv++;
            }
        }
        if (iv != p.nVectors) {
            fprintf(stderr, "vector cursor mismatch %ld vs %ld\n", (long) iv, (long) p.nVectors);
            exit(2);
        }
    }

    void dumpProfile(const std::string &name, Profile &p) {
        std::vector<P> w, f;
        std::vector<uint8_t> c;
        dense(p, w, c, f);
        out.vec(name + ".w", w);
        out.vec(name + ".c", c);
        out.mat(name + ".f", f, nPos, nCodes);
        if (p.codeDistSize > 0) {
            std::vector<P> cd(p.codeDist, p.codeDist + nPos * nCodes);
            out.mat(name + ".cd", cd, nPos, nCodes);
        }
    }

    int64_t hashProfile(Profile &p) {
        std::vector<P> w, f;
        std::vector<uint8_t> c;
        dense(p, w, c, f);
        uint64_t h = 0xCBF29CE484222325ull;
        h = fnv(h, w.data(), w.size() * sizeof(P));
        h = fnv(h, c.data(), c.size());
        h = fnv(h, f.data(), f.size() * sizeof(P));
        return (int64_t) h;
    }

    void dumpAllProfileHashes(const std::string &name, int64_t upto) {
        std::vector<int64_t> h(upto), nv(upto);
        for (int64_t i = 0; i < upto; i++) {
            h[i] = hashProfile(nj.profiles[i]);
            nv[i] = nj.profiles[i].nVectors;
        }
        out.vec(name + ".hash", h);
        out.vec(name + ".nvec", nv);
    }

    /* one-vs-all sweep through the reference's setBestHit (NJ.tcc:3571) + its sort (Utils.h:126) */
    void sweep(const std::string &name, int64_t node, int64_t nActive) {
        std::vector<Besthit> all(nj.maxnode);
        Besthit best;
        nj.setBestHit(node, nActive, best, all.data(), false);
        std::vector<int64_t> hi(nj.maxnode), hj(nj.maxnode);
        std::vector<P> hw(nj.maxnode), hd(nj.maxnode), hc(nj.maxnode);
        for (int64_t j = 0; j < nj.maxnode; j++) {
            hi[j] = all[j].i;
            hj[j] = all[j].j;
            hw[j] = all[j].weight;
            hd[j] = all[j].dist;
            hc[j] = all[j].criterion;
        }
        out.vec(name + ".i", hi);
        out.vec(name + ".j", hj);
        out.vec(name + ".weight", hw);
        out.vec(name + ".dist", hd);
        out.vec(name + ".crit", hc);
        out.vec(name + ".best", std::vector<int64_t>{best.i, best.j});
        out.vec(name + ".bestval", std::vector<P>{best.weight, best.dist, best.criterion});
        /* the reference's sort of the sweep (tie rule of SURVEY §0.3) */
        psort(all.begin(), all.end(), typename NJ::CompareHitsByCriterion());
        std::vector<int64_t> order(nj.maxnode);
        for (int64_t k = 0; k < nj.maxnode; k++) order[k] = all[k].j;
        out.vec(name + ".sorted_j", order);
        /* state after the sweep: lazily refreshed out-distances (NJ.tcc:1092-1098) */
        std::vector<P> od(nj.outDistances.begin(), nj.outDistances.begin() + nj.maxnode);
        std::vector<int64_t> na(nj.nOutDistActive.begin(), nj.nOutDistActive.begin() + nj.maxnode);
        out.vec(name + ".outdist_after", od);
        out.vec(name + ".noutactive_after", na);
    }

    void dumpNodeArrays(const std::string &name) {
        int64_t n = nj.maxnode;
        std::vector<int64_t> par(nj.parent.begin(), nj.parent.begin() + n);
        std::vector<int64_t> ch((size_t) n * 3, -1), nch(n);
        for (int64_t i = 0; i < n; i++) {
            nch[i] = nj.child[i].nChild;
            for (int k = 0; k < nj.child[i].nChild; k++) ch[i * 3 + k] = nj.child[i].child[k];
        }
        out.vec(name + ".parent", par);
        out.mat(name + ".child", ch, n, 3);
        out.vec(name + ".nchild", nch);
        out.vec(name + ".branchlength", std::vector<P>(nj.branchlength.begin(), nj.branchlength.begin() + n));
        out.vec(name + ".diameter", std::vector<P>(nj.diameter.begin(), nj.diameter.begin() + n));
        out.vec(name + ".vardiameter", std::vector<P>(nj.varDiameter.begin(), nj.varDiameter.begin() + n));
        out.vec(name + ".selfdist", std::vector<P>(nj.selfdist.begin(), nj.selfdist.begin() + n));
        out.vec(name + ".selfweight", std::vector<P>(nj.selfweight.begin(), nj.selfweight.begin() + n));
        out.vec(name + ".outdist", std::vector<P>(nj.outDistances.begin(), nj.outDistances.begin() + n));
        out.vec(name + ".noutactive",
                std::vector<int64_t>(nj.nOutDistActive.begin(), nj.nOutDistActive.begin() + n));
        out.scalar<int64_t>(name + ".root", nj.root);
        out.scalar<double>(name + ".totdiam", nj.totdiam);
    }

    void pairDistances(const std::string &name, const std::vector<int64_t> &a, const std::vector<int64_t> &b) {
        std::vector<P> d(a.size()), w(a.size());
        for (size_t k = 0; k < a.size(); k++) {
            Besthit h;
            nj.profileDist(nj.profiles[a[k]], nj.profiles[b[k]], h);
            d[k] = h.dist;
            w[k] = h.weight;
        }
        out.vec(name + ".a", a);
        out.vec(name + ".b", b);
        out.vec(name + ".dist", d);
        out.vec(name + ".weight", w);
    }

    /* Re-create "the NJ state after J joins" from the finished tree: node v < nSeqs+J is active iff its
       parent was created later.  Every number is then produced by reference members on that state. */
    void midState(const std::string &name, int64_t J, Rng &rng, int nSweeps) {
        const int64_t savedMax = nj.maxnode;
        std::vector<int64_t> savedParent(nj.parent);
        std::vector<P> savedOut(nj.outDistances.begin(), nj.outDistances.end());
        std::vector<int64_t> savedNOut(nj.nOutDistActive);
        const double savedTot = nj.totdiam;

        const int64_t lim = nSeqs + J;
        std::vector<int64_t> active;
        for (int64_t v = 0; v < lim; v++) {
            if (savedParent[v] >= lim) {
                nj.parent[v] = -1;
                active.push_back(v);
            }
        }
        nj.maxnode = lim;
        const int64_t nActive = (int64_t) active.size();
        out.scalar<int64_t>(name + ".J", J);
        out.scalar<int64_t>(name + ".nActive", nActive);
        out.vec(name + ".active", active);

        /* out-profile of the active set: NJ.tcc:729 */
        std::vector<Profile *> ap;
        nj.totdiam = 0;
        for (int64_t v: active) {
            ap.push_back(&nj.profiles[v]);
            nj.totdiam += nj.diameter[v];
        }
        nj.outProfile(nj.outprofile, ap, nActive);
        dumpProfile(name + ".outprofile", nj.outprofile);
        out.scalar<double>(name + ".totdiam", nj.totdiam);

        /* fresh out-distances for every active node: NJ.tcc:1012 */
        for (int64_t v: active) {
            nj.nOutDistActive[v] = nSeqs * 10;
            nj.setOutDistance(v, nActive);
        }
        out.vec(name + ".outdist_fresh", std::vector<P>(nj.outDistances.begin(), nj.outDistances.begin() + lim));

        /* mix of fresh / allowed-stale / too-stale out-distances, as the top-hits phase leaves them */
        const int64_t nDiffAllow = nj.options.tophitsMult > 0 ? (int64_t) (nActive * nj.options.staleOutLimit) : 0;
        out.scalar<int64_t>(name + ".nDiffAllow", nDiffAllow);
        for (int64_t v: active) {
            int64_t r = rng.below(4);
            if (r == 1 && nDiffAllow > 0) {
                nj.nOutDistActive[v] = nSeqs * 10;
                nj.setOutDistance(v, nActive + 1 + rng.below(nDiffAllow));
            } else if (r == 2) {
                nj.nOutDistActive[v] = nSeqs * 10;
                nj.setOutDistance(v, nActive + nDiffAllow + 1 + rng.below(5));
            }
        }
        out.vec(name + ".outdist_in", std::vector<P>(nj.outDistances.begin(), nj.outDistances.begin() + lim));
        out.vec(name + ".noutactive_in",
                std::vector<int64_t>(nj.nOutDistActive.begin(), nj.nOutDistActive.begin() + lim));

        std::vector<P> stOut(nj.outDistances.begin(), nj.outDistances.end());
        std::vector<int64_t> stN(nj.nOutDistActive);
        std::vector<int64_t> queries;
        for (int s = 0; s < nSweeps; s++) {
            int64_t q = active[rng.below(nActive)];
            if (s == 0) {
                /* make sure an internal node is queried when one exists */
                for (int64_t v: active) if (v >= nSeqs) q = v;
            }
            queries.push_back(q);
            /* each sweep starts from the same input state so that the kernels can be tested one sweep at a time */
            std::copy(stOut.begin(), stOut.end(), nj.outDistances.begin());
            nj.nOutDistActive = stN;
            sweep(name + ".sweep" + std::to_string(s), q, nActive);
        }
        out.vec(name + ".queries", queries);

        /* incremental out-profile update for the next join (children of node `lim`): NJ.tcc:943 */
        if (lim < savedMax - 1 && nj.child[lim].nChild == 2) {
            int64_t a = nj.child[lim].child[0], b = nj.child[lim].child[1];
            nj.updateOutProfile(nj.outprofile, nj.profiles[a], nj.profiles[b], nj.profiles[lim], nActive);
            dumpProfile(name + ".outprofile_updated", nj.outprofile);
            out.vec(name + ".update_abn", std::vector<int64_t>{a, b, lim});
        }

        nj.parent = savedParent;
        nj.maxnode = savedMax;
        std::copy(savedOut.begin(), savedOut.end(), nj.outDistances.begin());
        nj.nOutDistActive = savedNOut;
        nj.totdiam = savedTot;
    }

    void dumpTransmat(const std::string &name) {
        auto &t = nj.transmat;
        std::vector<P> stat(t.stat, t.stat + nCodes), statinv(t.statinv, t.statinv + nCodes),
                eval(t.eigenval, t.eigenval + nCodes);
        std::vector<P> cf((size_t) (nCodes + 1) * nCodes), einv((size_t) nCodes * nCodes),
                einvT((size_t) nCodes * nCodes);
        for (int i = 0; i < nCodes; i++) {
            for (int k = 0; k < nCodes; k++) {
                cf[i * nCodes + k] = t.codeFreq[i][k];
                einv[i * nCodes + k] = t.eigeninv[i][k];
                einvT[i * nCodes + k] = t.eigeninvT[i][k];
            }
        }
        for (int k = 0; k < nCodes; k++) cf[nCodes * nCodes + k] = t.codeFreq[NOCODE][k];
        out.vec(name + ".stat", stat);
        out.vec(name + ".statinv", statinv);
        out.vec(name + ".eigenval", eval);
        out.mat(name + ".codefreq", cf, nCodes + 1, nCodes); /* last row = NOCODE (gap) */
        out.mat(name + ".eigeninv", einv, nCodes, nCodes);
        out.mat(name + ".eigeninvT", einvT, nCodes, nCodes);
    }

    /* ML block: rates -> recomputeMLProfiles (NJ.tcc:3516) -> treeLogLk (NJ.tcc:5160) + loose calls */
    void mlBlock(const std::string &name, Rng &rng, int nFull) {
        nj.recomputeMLProfiles();
        dumpAllProfileHashes(name + ".profiles", nj.maxnode - 1);
        Rng pick(rng.next());
        for (int k = 0; k < nFull; k++) {
            int64_t v = nSeqs + pick.below(nj.maxnode - 1 - nSeqs);
            out.scalar<int64_t>(name + ".full" + std::to_string(k) + ".node", v);
            dumpProfile(name + ".full" + std::to_string(k), nj.profiles[v]);
        }
        std::vector<double> site(nPos);
        double ll = nj.treeLogLk(site.data());
        out.scalar<double>(name + ".treeloglk", ll);
        out.vec(name + ".site_loglk", site);
        double ll2 = nj.treeLogLk(nullptr);
        out.scalar<double>(name + ".treeloglk_nosite", ll2);

        /* loose pairLogLk (NJ.tcc:1192) and posteriorProfile (NJ.tcc:2137) calls on assorted inputs */
        const int nPairs = 24;
        std::vector<int64_t> a(nPairs), b(nPairs);
        std::vector<double> len(nPairs), val(nPairs);
        std::vector<double> sites((size_t) nPairs * nPos);
        for (int k = 0; k < nPairs; k++) {
            a[k] = pick.below(nj.maxnode - 1);
            b[k] = pick.below(nj.maxnode - 1);
            double u = pick.unit();
            len[k] = k % 6 == 0 ? 1e-6 : (k % 6 == 1 ? 3.0 * u : 0.4 * u * u);
            std::vector<double> s(nPos, 1.0);
            val[k] = nj.pairLogLk(nj.profiles[a[k]], nj.profiles[b[k]], len[k], s.data());
            std::copy(s.begin(), s.end(), sites.begin() + (size_t) k * nPos);
        }
        out.vec(name + ".pll.a", a);
        out.vec(name + ".pll.b", b);
        out.vec(name + ".pll.len", len);
        out.vec(name + ".pll.val", val);
        out.mat(name + ".pll.site", sites, nPairs, nPos);

        const int nPost = 8;
        std::vector<int64_t> pa(nPost), pb(nPost);
        std::vector<double> l1(nPost), l2(nPost);
        for (int k = 0; k < nPost; k++) {
            pa[k] = pick.below(nj.maxnode - 1);
            pb[k] = pick.below(nj.maxnode - 1);
            l1[k] = k == 0 ? 0.0 : 0.5 * pick.unit();
            l2[k] = k == 1 ? 1e-7 : 0.5 * pick.unit() * pick.unit();
            Profile o(nPos, 0);
            nj.posteriorProfile(o, nj.profiles[pa[k]], nj.profiles[pb[k]], l1[k], l2[k]);
            dumpProfile(name + ".post" + std::to_string(k), o);
        }
        out.vec(name + ".post.a", pa);
        out.vec(name + ".post.b", pb);
        out.vec(name + ".post.len1", l1);
        out.vec(name + ".post.len2", l2);

        /* optimizeAllBranchLengths (NJ.tcc:5065) twice from this state, with the tree likelihood after each round;
           lengths and profiles are put back afterwards so that the blocks stay independent */
        auto savedLen = nj.branchlength;
        for (int round = 1; round <= 2; round++) {
            nj.optimizeAllBranchLengths();
            const std::string key = name + ".opt" + std::to_string(round);
            out.vec(key + ".branchlength", std::vector<P>(nj.branchlength.begin(), nj.branchlength.begin() + nj.maxnode));
            out.scalar<double>(key + ".treeloglk", nj.treeLogLk(nullptr));
        }
        nj.branchlength = savedLen;
        nj.recomputeMLProfiles();

        /* testSplitsML's per-split evaluation (NJ.tcc:6856-6925) for a few splits: MLQuartetLogLk of AB|CD, MLQuartetOptimize
           of AC|BD and AD|BC, second pass for the better alternative when close.  D's profile (the up-profile of the
           parent, or the root's other child) is dumped so that a test can put it on the device. */
        {
            std::vector<std::unique_ptr<Profile>> upProfiles(nj.maxnodes);
            std::vector<int64_t> picks;
            for (int64_t v = nSeqs; v < nj.maxnode - 1 && picks.size() < 6; v += std::max<int64_t>(1, (nj.maxnode - 1 - nSeqs) / 6))
                if (v != nj.root) picks.push_back(v);
            out.vec(name + ".quartet.nodes", picks);
            for (size_t k = 0; k < picks.size(); k++) {
                const int64_t node = picks[k];
                const std::string key = name + ".quartet" + std::to_string(k);
                Profile *profiles4[4];
                int64_t nodeABCD[4];
                nj.setupABCD(node, profiles4, upProfiles.data(), nodeABCD, /*useML*/true);
                out.vec(key + ".abcd", std::vector<int64_t>(nodeABCD, nodeABCD + 4));
                dumpProfile(key + ".D", *profiles4[3]);
                double len[5];
                for (int i = 0; i < 4; i++) len[i] = nj.branchlength[nodeABCD[i]];
                len[4] = nj.branchlength[node];
                out.vec(key + ".len", std::vector<double>(len, len + 5));
                double lenABvsCD[5] = {len[0], len[1], len[2], len[3], len[4]};
                double lenACvsBD[5] = {len[0], len[2], len[1], len[3], len[4]};
                double lenADvsBC[5] = {len[0], len[3], len[2], len[1], len[4]};
                std::vector<double> site(3 * nPos);
                double loglk[3];
                loglk[0] = nj.MLQuartetLogLk(*profiles4[0], *profiles4[1], *profiles4[2], *profiles4[3], lenABvsCD, &site[0]);
                loglk[1] = nj.MLQuartetOptimize(*profiles4[0], *profiles4[2], *profiles4[1], *profiles4[3], lenACvsBD, nullptr, &site[nPos]);
                loglk[2] = nj.MLQuartetOptimize(*profiles4[0], *profiles4[3], *profiles4[2], *profiles4[1], lenADvsBC, nullptr, &site[2 * nPos]);
                if (loglk[1] > loglk[2]) {
                    if (loglk[1] > loglk[0] - Constants::closeLogLkLimit)
                        loglk[1] = nj.MLQuartetOptimize(*profiles4[0], *profiles4[2], *profiles4[1], *profiles4[3], lenACvsBD, nullptr, &site[nPos]);
                } else {
                    if (loglk[2] > loglk[0] - Constants::closeLogLkLimit)
                        loglk[2] = nj.MLQuartetOptimize(*profiles4[0], *profiles4[3], *profiles4[2], *profiles4[1], lenADvsBC, nullptr, &site[2 * nPos]);
                }
                out.vec(key + ".loglk", std::vector<double>(loglk, loglk + 3));
                out.vec(key + ".lenAC", std::vector<double>(lenACvsBD, lenACvsBD + 5));
                out.vec(key + ".lenAD", std::vector<double>(lenADvsBC, lenADvsBC + 5));
                out.mat(key + ".site", site, 3, nPos);
            }
        }
    }
};

template<typename P, template<class> class Op>
static int run(Options &options, const std::string &fasta, const std::string &outPath, uint64_t seed, bool aa, bool partitionOnly = false) {
    typedef Op<P> op_t;
    std::ostringstream logbuf;
    std::ostream &log = logbuf;
    std::ifstream in(fasta);
    if (in.fail()) {
        fprintf(stderr, "cannot read %s\n", fasta.c_str());
        return 2;
    }
    Alignment aln(options, in, log);
    aln.readAlignment();
    Uniquify unique(aln);

    DistanceMatrix<P, op_t::ALIGNMENT> dmat;
    dmat.setted = false;
    TransitionMatrix<P, op_t::ALIGNMENT> tmat;
    if (options.useMatrix) {
        dmat.matrixBLOSUM45();
        dmat.setupDistanceMatrix(options, log);
    }
    ProgressReport progress(false, options.verbose, false);
    std::vector<std::string> noCons;
    std::unique_ptr<DiskMemory> consDisk;
    NeighbourJoining<P, Op> nj(options, log, progress, unique.uniqueSeq, aln.nPos, noCons, dmat, tmat, unique.disk,
                               consDisk);

    Dump out(outPath.c_str());
    Harness<P, Op> h(nj, out);
    Rng rng(seed);
    const int64_t nSeqs = nj.nSeqs, nPos = nj.nPos;
    const int nCodes = options.nCodes;
    out.scalar<int64_t>("nSeqs", nSeqs);
    out.scalar<int64_t>("nPos", nPos);
    out.scalar<int64_t>("nCodes", nCodes);
    out.scalar<int64_t>("nCodeSize", nj.nCodeSize);
    out.scalar<int64_t>("precisionBytes", (int64_t) sizeof(P));
    out.scalar<int64_t>("nAlnSeqs", (int64_t) aln.names.size());

    /* ---- leaves and the initial state (NJ ctor, NJ.tcc:210-272) */
    std::vector<uint8_t> codes((size_t) nSeqs * nPos);
    std::vector<int64_t> gaps(nSeqs);
    for (int64_t i = 0; i < nSeqs; i++) {
        for (int64_t p = 0; p < nPos; p++) codes[i * nPos + p] = (uint8_t) nj.profiles[i].codes[p];
        gaps[i] = nj.profiles[i].nGaps;
    }
    out.mat("leaf.codes", codes, nSeqs, nPos);
    out.vec("leaf.ngaps", gaps);
    h.dumpProfile("init.outprofile", nj.outprofile);
    out.vec("init.outdist", std::vector<P>(nj.outDistances.begin(), nj.outDistances.begin() + nSeqs));
    out.vec("init.selfweight", std::vector<P>(nj.selfweight.begin(), nj.selfweight.begin() + nSeqs));

    if (dmat) {
        std::vector<P> dd(nCodes * nCodes), cf(nCodes * nCodes), ev(dmat.eigenval, dmat.eigenval + nCodes),
                et(dmat.eigentot, dmat.eigentot + nCodes);
        for (int i = 0; i < nCodes; i++) {
            for (int k = 0; k < nCodes; k++) {
                dd[i * nCodes + k] = dmat.distances[i][k];
                cf[i * nCodes + k] = dmat.codeFreq[i][k];
            }
        }
        out.mat("dmat.distances", dd, nCodes, nCodes);
        out.mat("dmat.codefreq", cf, nCodes, nCodes);
        out.vec("dmat.eigenval", ev);
        out.vec("dmat.eigentot", et);
    }

    /* leaf x leaf seqDist (NJ.tcc:1601) on random pairs */
    {
        const int n = 64;
        std::vector<int64_t> a(n), b(n);
        std::vector<P> d(n), w(n);
        for (int k = 0; k < n; k++) {
            a[k] = rng.below(nSeqs);
            b[k] = k == 0 ? a[k] : rng.below(nSeqs);
            typename NeighbourJoining<P, Op>::Besthit hit;
            nj.seqDist(nj.profiles[a[k]].codes, nj.profiles[b[k]].codes, hit);
            d[k] = hit.dist;
            w[k] = hit.weight;
        }
        out.vec("seqdist.a", a);
        out.vec("seqdist.b", b);
        out.vec("seqdist.dist", d);
        out.vec("seqdist.weight", w);
    }

    /* one-vs-all leaf sweeps in the initial state (what setAllLeafTopHits does per seed, NJ.tcc:3801) */
    {
        std::vector<int64_t> qs;
        for (int s = 0; s < 3; s++) {
            int64_t q = s == 0 ? 0 : rng.below(nSeqs);
            qs.push_back(q);
            h.sweep("init.sweep" + std::to_string(s), q, nSeqs);
        }
        out.vec("init.queries", qs);
    }

    /* ---- the reference builds its NJ tree; every internal profile is then a real averageProfile output */
    nj.fastNJ();
    h.dumpNodeArrays("nj");
    if (partitionOnly) {
        /* treePartitioning (NJ.tcc:5540-5750, private) as the reference's threaded NNI / length rounds call it: penalty 2 and 1, for
           several thread counts - the partitions the subtree schedule of the backend has to reproduce */
        const int Ts[] = {2, 3, 4, 8, 16, 64};
        for (int T: Ts)
            for (int penalty = 1; penalty <= 2; penalty++) {
                options.threads = T;
                std::vector<int64_t> part = nj.treePartitioning(penalty);
                out.vec("part.T" + std::to_string(T) + ".p" + std::to_string(penalty), part);
            }
        options.threads = 1;
        return 0;
    }
    h.dumpAllProfileHashes("nj.profiles", nj.maxnode - 1);
    {
        /* a handful of complete internal profiles: early, middle and late joins */
        std::vector<int64_t> full = {nSeqs, nSeqs + 1, nSeqs + (nj.maxnode - 1 - nSeqs) / 2, nj.maxnode - 3,
                                     nj.maxnode - 2};
        for (size_t k = 0; k < full.size(); k++) {
            out.scalar<int64_t>("nj.full" + std::to_string(k) + ".node", full[k]);
            h.dumpProfile("nj.full" + std::to_string(k), nj.profiles[full[k]]);
        }
    }
    {
        const int n = 256;
        std::vector<int64_t> a(n), b(n);
        for (int k = 0; k < n; k++) {
            a[k] = rng.below(nj.maxnode - 1);
            b[k] = k % 16 == 0 ? a[k] : rng.below(nj.maxnode - 1);
            if (k % 5 == 1) a[k] = nSeqs + rng.below(nj.maxnode - 1 - nSeqs); /* force internal */
            if (k % 5 == 2) b[k] = nSeqs + rng.below(nj.maxnode - 1 - nSeqs);
        }
        h.pairDistances("pdist", a, b);
    }

    /* ---- mid-run states */
    {
        const int64_t nJoins = nj.maxnode - 1 - nSeqs;
        int64_t Js[3] = {nJoins / 8, nJoins / 2, nJoins - 6};
        for (int s = 0; s < 3; s++) h.midState("mid" + std::to_string(s), Js[s], rng, 3);
    }

    /* ---- ML: rate categories (CAT-style spread, arbitrary site assignment) */
    {
        const int nCat = 20;
        nj.rates.reset(nCat, nPos);
        for (int i = 0; i < nCat; i++) {
            nj.rates.rates[i] = (P) std::exp(-std::log(20.0) + i * 2.0 * std::log(20.0) / 19.0);
        }
        for (int64_t p = 0; p < nPos; p++) nj.rates.ratecat[p] = rng.below(nCat);
        out.vec("ml.rates", std::vector<P>(nj.rates.rates.begin(), nj.rates.rates.end()));
        out.vec("ml.ratecat", nj.rates.ratecat);
        /* NJ branch lengths can be negative; the ML code clamps them, keep them as they are */
        if (!aa) {
            h.mlBlock("jc", rng, 3);
            double gtrrates[6] = {1.2, 3.1, 0.7, 0.9, 3.6, 1.0};
            double gtrfreq[4] = {0.31, 0.19, 0.23, 0.27};
            tmat.createGTR(options, gtrrates, gtrfreq);
            h.dumpTransmat("gtr.tm");
            h.mlBlock("gtr", rng, 3);
        } else {
            tmat.createTransitionMatrixLG08(options);
            h.dumpTransmat("lg.tm");
            h.mlBlock("lg", rng, 3);
        }
    }
    fprintf(stderr, "whitebox: wrote %s (nSeqs=%ld nPos=%ld maxnode=%ld)\n", outPath.c_str(), (long) nSeqs, (long) nPos,
            (long) nj.maxnode);
    return 0;
}

/* Model constants and the tables the reference derives from them (TransitionMatrix.tcc:14-24, 158-232;
   DistanceMatrix.tcc:33-36, 102-155), for the product's data header (tools/gen_aa_tables.py) and its CPU test. */
template<typename P, int ALIGN>
static void dumpModelTables(Dump &d, const Options &options, const std::string &suffix) {
    const int n = 20;
    for (int model = 0; model < 3; model++) {
        TransitionMatrix<P, ALIGN> t;
        const char *name = model == 0 ? "jtt" : model == 1 ? "wag" : "lg";
        if (model == 0) t.createTransitionMatrixJTT92(options);
        else if (model == 1) t.createTransitionMatrixWAG01(options);
        else t.createTransitionMatrixLG08(options);
        std::vector<P> stat(t.stat, t.stat + n), statinv(t.statinv, t.statinv + n), eval(t.eigenval, t.eigenval + n);
        std::vector<P> cf((size_t) (n + 1) * n), einv((size_t) n * n), einvT((size_t) n * n);
        for (int i = 0; i < n; i++)
            for (int k = 0; k < n; k++) {
                cf[i * n + k] = t.codeFreq[i][k];
                einv[i * n + k] = t.eigeninv[i][k];
                einvT[i * n + k] = t.eigeninvT[i][k];
            }
        for (int k = 0; k < n; k++) cf[n * n + k] = t.codeFreq[NOCODE][k];
        const std::string pre = std::string(name) + "." + suffix + ".";
        d.vec(pre + "stat", stat);
        d.vec(pre + "statinv", statinv);
        d.vec(pre + "eigenval", eval);
        d.mat(pre + "codefreq", cf, n + 1, n);
        d.mat(pre + "eigeninv", einv, n, n);
        d.mat(pre + "eigeninvT", einvT, n, n);
    }
    DistanceMatrix<P, ALIGN> dm;
    std::ostringstream log;
    dm.matrixBLOSUM45();
    dm.setupDistanceMatrix(options, log);
    std::vector<P> dd((size_t) n * n), ei((size_t) n * n), cf((size_t) n * n), ev(dm.eigenval, dm.eigenval + n),
            et(dm.eigentot, dm.eigentot + n);
    for (int i = 0; i < n; i++)
        for (int k = 0; k < n; k++) {
            dd[i * n + k] = dm.distances[i][k];
            ei[i * n + k] = dm.eigeninv[i][k];
            cf[i * n + k] = dm.codeFreq[i][k];
        }
    d.mat("blosum45." + suffix + ".distances", dd, n, n);
    d.mat("blosum45." + suffix + ".eigeninv", ei, n, n);
    d.mat("blosum45." + suffix + ".codefreq", cf, n, n);
    d.vec("blosum45." + suffix + ".eigenval", ev);
    d.vec("blosum45." + suffix + ".eigentot", et);
}

static int dumpTables(const char *path) {
    Dump d(path);
    Options options;
    options.nCodes = 20;
    const int n = 20;
    typedef TransitionMatrix<double, 32> TM;
    const double *stats[3] = {TM::statJTT92, TM::statWAG01, TM::statLG08};
    const double (*mats[3])[MAXCODES] = {TM::matrixJTT92, TM::matrixWAG01, TM::matrixLG08};
    const char *names[3] = {"jtt", "wag", "lg"};
    for (int m = 0; m < 3; m++) {
        std::vector<double> st(stats[m], stats[m] + n), mx((size_t) n * n);
        for (int i = 0; i < n; i++)
            for (int k = 0; k < n; k++) mx[i * n + k] = mats[m][i][k];
        d.vec(std::string(names[m]) + ".raw.stat", st);
        d.mat(std::string(names[m]) + ".raw.matrix", mx, n, n);
    }
    {   /* BLOSUM45 as shipped: distances, eigeninv, eigenval (double literals; the float build narrows them) */
        const DistanceMatrix<double, 32> &b = DistanceMatrix<double, 32>::_matrixBLOSUM45;
        std::vector<double> dd((size_t) n * n), ei((size_t) n * n), ev(b.eigenval, b.eigenval + n);
        for (int i = 0; i < n; i++)
            for (int k = 0; k < n; k++) {
                dd[i * n + k] = b.distances[i][k];
                ei[i * n + k] = b.eigeninv[i][k];
            }
        d.mat("blosum45.raw.distances", dd, n, n);
        d.mat("blosum45.raw.eigeninv", ei, n, n);
        d.vec("blosum45.raw.eigenval", ev);
    }
    dumpModelTables<float, 16>(d, options, "f32");    /* SSE128Operations::ALIGNMENT */
    dumpModelTables<double, 32>(d, options, "f64");   /* AVX256Operations::ALIGNMENT */
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: whitebox <nt_f32|nt_f64|aa_f32|aa_f64> <fasta> <out.vfx> [seed]\n");
        return 2;
    }
    std::string mode = argv[1];
    if (mode == "knuth") {   /* the reference's random stream (Knuth.cpp: knuth_rand, never re-seeded by the pipeline) */
        Dump d(argv[3]);
        std::vector<double> r(5000);
        for (double &x: r) x = knuth_rand();
        d.vec("knuth.rand", r);
        return 0;
    }
    if (mode == "tables") return dumpTables(argv[3]);
    uint64_t seed = argc > 4 ? strtoull(argv[4], nullptr, 10) : 1;
    omp_set_num_threads(1);

    Options options;
    options.verbose = 1;
    options.showProgress = false;
    options.threads = 1;
    options.diskComputing = false;
    options.seed = 1;
    options.nBootstrap = 0;
    options.extension = "whitebox";
    bool aa = mode.substr(0, 2) == "aa";
    bool dbl = mode.substr(3, 3) == "f64";
    options.nCodes = aa ? 20 : 4;
    options.doublePrecision = dbl;
    options.bUseLg = aa;
    /* what VeryFastTree::settings() derives (VeryFastTree.cpp:93-129) */
    options.codesString = aa ? Constants::codesStringAA : Constants::codesStringNT;
    options.useMatrix = aa;
    if (dbl) {
        options.MLMinBranchLengthTolerance = Constants::MLMinBranchLengthToleranceDouble;
        options.MLFTolBranchLength = Constants::MLFTolBranchLengthDouble;
        options.MLMinBranchLength = Constants::MLMinBranchLengthDouble;
        options.MLMinRelBranchLength = Constants::MLMinRelBranchLengthDouble;
        options.fPostTotalTolerance = Constants::fPostTotalToleranceDouble;
    } else {
        options.MLMinBranchLengthTolerance = Constants::MLMinBranchLengthToleranceFloat;
        options.MLFTolBranchLength = Constants::MLFTolBranchLengthFloat;
        options.MLMinBranchLength = Constants::MLMinBranchLengthFloat;
        options.MLMinRelBranchLength = Constants::MLMinRelBranchLengthFloat;
        options.fPostTotalTolerance = Constants::fPostTotalToleranceFloat;
    }
    /* backends exactly as the dispatcher picks them (VeryFastTree.cpp:46-66): nt float -> SSE3,
       everything else with AVX available -> AVX2 */
    if (mode == "nt_f32:partition") return run<float, SSE128Operations>(options, argv[2], argv[3], seed, false, true);
    if (mode == "nt_f32") return run<float, SSE128Operations>(options, argv[2], argv[3], seed, false);
    if (mode == "nt_f64") return run<double, AVX256Operations>(options, argv[2], argv[3], seed, false);
    if (mode == "aa_f32") return run<float, SSE128Operations>(options, argv[2], argv[3], seed, true);
    if (mode == "aa_f64") return run<double, AVX256Operations>(options, argv[2], argv[3], seed, true);
    fprintf(stderr, "unknown mode %s\n", mode.c_str());
    return 2;
}
