"""Pins the CPU oracle (oracle/vft_oracle.c) against golden vectors dumped from the compiled reference
(oracle/whitebox.cpp -> tests/golden/wb_*.npz).  Integer and numeric_t outputs must match bit for bit;
log-likelihoods (double sums whose order the oracle shares with the reference) to 1e-12 relative."""
import numpy as np
import pytest

import golden_util as G
from oracle import Oracle, tolerances


@pytest.fixture(scope="module", params=G.WHITEBOX)
def fx(request):
    d = G.load(request.param)
    orc = Oracle(G.dtype_of(d))
    dm = G.dmat_of(d, orc)
    profs = G.build_nj_profiles(d, orc, dm)
    return dict(name=request.param, d=d, orc=orc, dm=dm, profs=profs)


def test_seqdist_pairs(fx):
    d, orc = fx["d"], fx["orc"]
    dist = d["dmat.distances"] if fx["dm"] is not None else None
    for a, b, gd, gw in zip(d["seqdist.a"], d["seqdist.b"], d["seqdist.dist"], d["seqdist.weight"]):
        od, ow = orc.seqdist(d["leaf.codes"][a], d["leaf.codes"][b], int(d["nCodes"]), dist)
        assert od == gd and ow == gw


def test_average_profile_chain_matches_every_nj_profile(fx):
    d, orc, profs = fx["d"], fx["orc"], fx["profs"]
    hashes = d["nj.profiles.hash"]
    assert len(hashes) == len(profs)
    got = np.array([orc.profile_hash(p) for p in profs], dtype=np.int64)
    assert np.array_equal(got, hashes)
    nvec = np.array([int(((p[0] > 0) & (p[1] == G.NOCODE)).sum()) for p in profs])
    assert np.array_equal(nvec, d["nj.profiles.nvec"])
    for k in range(5):
        node = int(d["nj.full%d.node" % k])
        assert G.profiles_equal(profs[node], G.fixture_profile(d, "nj.full%d" % k))


def test_profiledist_pairs(fx):
    d, orc, profs = fx["d"], fx["orc"], fx["profs"]
    for a, b, gd, gw in zip(d["pdist.a"], d["pdist.b"], d["pdist.dist"], d["pdist.weight"]):
        od, ow = orc.profiledist(profs[a], profs[b], None, fx["dm"])
        assert od == gd and ow == gw, (a, b)


def test_selfdist_selfweight(fx):
    d, orc, profs = fx["d"], fx["orc"], fx["profs"]
    n_seqs = int(d["nSeqs"])
    for v in G.internal_nodes(d):
        od, ow = orc.profiledist(profs[v], profs[v], None, fx["dm"])
        assert od == d["nj.selfdist"][v] and ow == d["nj.selfweight"][v]
    assert np.array_equal(d["nj.selfweight"][:n_seqs], (int(d["nPos"]) - d["leaf.ngaps"]).astype(orc.dt))


def _init_state(fx):
    d, orc, profs = fx["d"], fx["orc"], fx["profs"]
    n_seqs = int(d["nSeqs"])
    tol = tolerances(orc.dt)[2]
    W, Cc, F = G.pack(profs[:n_seqs], orc.dt)
    outp, cd = orc.out_profile(W, Cc, F, fx["dm"], tol)
    return W, Cc, F, outp, cd


def test_initial_out_profile_and_out_distances(fx):
    d, orc = fx["d"], fx["orc"]
    n_seqs = int(d["nSeqs"])
    W, Cc, F, outp, cd = _init_state(fx)
    assert G.profiles_equal(outp, G.fixture_profile(d, "init.outprofile"))
    if cd is not None:
        assert np.array_equal(cd, d["init.outprofile.cd"])
    for i in range(n_seqs):
        dd, ww = orc.profiledist((W[i], Cc[i], F[i]), outp, cd, fx["dm"])
        od = orc.out_distance(dd, ww, n_seqs, d["init.selfweight"][i], 0.0, 0.0, 0.0)
        assert od == d["init.outdist"][i]


def _check_sweep(d, orc, st, key, node, n_active, n_diff_allow, od_in, na_in):
    res = orc.set_best_hit(st, node, n_active, n_diff_allow, od_in, na_in)
    assert np.array_equal(res["i"], d[key + ".i"])
    assert np.array_equal(res["j"], d[key + ".j"])
    assert np.array_equal(res["weight"], d[key + ".weight"])
    assert np.array_equal(res["dist"], d[key + ".dist"])
    assert np.array_equal(res["crit"], d[key + ".crit"])
    assert res["best_j"] == int(d[key + ".best"][1])
    n = st.maxnode
    assert np.array_equal(res["outdist"][:n], d[key + ".outdist_after"][:n])
    assert np.array_equal(res["noutactive"][:n], d[key + ".noutactive_after"][:n])
    # the reference's sort of this sweep: ascending criterion, ties by descending position
    assert np.array_equal(orc.sort_hits(res["crit"]), d[key + ".sorted_j"])


def test_initial_leaf_sweeps(fx):
    d, orc = fx["d"], fx["orc"]
    n_seqs = int(d["nSeqs"])
    W, Cc, F, outp, cd = _init_state(fx)
    z = np.zeros(n_seqs, orc.dt)
    st = orc.state(n_seqs, W, Cc, F, np.full(n_seqs, -1, np.int64), z, d["init.selfweight"], z, 0.0, outp, cd,
                   fx["dm"])
    na = np.full(n_seqs, n_seqs, np.int64)
    for s, q in enumerate(d["init.queries"]):
        _check_sweep(d, orc, st, "init.sweep%d" % s, int(q), n_seqs, int(n_seqs * 0.01), d["init.outdist"], na)


@pytest.mark.parametrize("mid", ["mid0", "mid1", "mid2"])
def test_mid_run_state(fx, mid):
    d, orc, profs = fx["d"], fx["orc"], fx["profs"]
    n_seqs = int(d["nSeqs"])
    tol = tolerances(orc.dt)[2]
    J, n_active = int(d[mid + ".J"]), int(d[mid + ".nActive"])
    lim = n_seqs + J
    parent = G.mid_parent(d, J)
    active = np.nonzero(parent < 0)[0]
    assert np.array_equal(active, d[mid + ".active"])
    W, Cc, F = G.pack(profs[:lim], orc.dt)
    # out-profile over the active set, in ascending node order (NJ.tcc:3017-3031)
    outp, cd = orc.out_profile(W[active], Cc[active], F[active], fx["dm"], tol)
    assert G.profiles_equal(outp, G.fixture_profile(d, mid + ".outprofile"))
    if cd is not None:
        assert np.array_equal(cd, d[mid + ".outprofile.cd"])
    totdiam = 0.0
    for v in active:
        totdiam += float(d["nj.diameter"][v])
    assert totdiam == float(d[mid + ".totdiam"])
    # fresh out-distances
    fresh = d[mid + ".outdist_fresh"]
    for v in active:
        dd, ww = orc.profiledist(profs[v], outp, cd, fx["dm"])
        od = orc.out_distance(dd, ww, n_active, d["nj.selfweight"][v], d["nj.selfdist"][v], d["nj.diameter"][v],
                              totdiam)
        assert od == fresh[v]
    st = orc.state(n_seqs, W, Cc, F, parent, d["nj.diameter"][:lim], d["nj.selfweight"][:lim],
                   d["nj.selfdist"][:lim], totdiam, outp, cd, fx["dm"])
    for s, q in enumerate(d[mid + ".queries"]):
        _check_sweep(d, orc, st, "%s.sweep%d" % (mid, s), int(q), n_active, int(d[mid + ".nDiffAllow"]),
                     d[mid + ".outdist_in"], d[mid + ".noutactive_in"])
    if mid + ".update_abn" in d:
        a, b, n = [int(x) for x in d[mid + ".update_abn"]]
        upd, ucd = orc.update_out_profile(outp, cd, profs[a], profs[b], profs[n], n_active, fx["dm"], tol)
        assert G.profiles_equal(upd, G.fixture_profile(d, mid + ".outprofile_updated"))
        if ucd is not None:
            assert np.array_equal(ucd, d[mid + ".outprofile_updated.cd"])


def _ml_profiles(fx, model):
    """recomputeMLProfiles (NJ.tcc:3516): posterior profile of every internal node, children first."""
    d, orc = fx["d"], fx["orc"]
    n_seqs, n_codes = int(d["nSeqs"]), int(d["nCodes"])
    min_len, min_rel, _ = tolerances(orc.dt)
    tm = None if model == "jc" else G.tmat_of(d, orc, model)
    profs = [orc.leaf_profile(d["leaf.codes"][i], n_codes) for i in range(n_seqs)]
    bl, child = d["nj.branchlength"], d["nj.child"]
    for v in G.internal_nodes(d):
        a, b = int(child[v, 0]), int(child[v, 1])
        profs.append(orc.posterior_profile(profs[a], profs[b], float(bl[a]), float(bl[b]), d["ml.rates"],
                                           d["ml.ratecat"], tm, min_len, min_rel))
    return profs, tm


def _models(fx):
    return ["lg"] if "_aa_" in fx["name"] else ["jc", "gtr"]


def test_ml_profiles_loglk_and_loose_calls(fx):
    d, orc = fx["d"], fx["orc"]
    n_seqs, n_pos = int(d["nSeqs"]), int(d["nPos"])
    min_len, min_rel, _ = tolerances(orc.dt)
    bl, child, root = d["nj.branchlength"], d["nj.child"], int(d["nj.root"])
    for model in _models(fx):
        profs, tm = _ml_profiles(fx, model)
        got = np.array([orc.profile_hash(p) for p in profs], dtype=np.int64)
        assert np.array_equal(got, d[model + ".profiles.hash"]), model
        for k in range(3):
            node = int(d["%s.full%d.node" % (model, k)])
            assert G.profiles_equal(profs[node], G.fixture_profile(d, "%s.full%d" % (model, k)))

        # treeLogLk (NJ.tcc:5114-5259): one pairLogLk per internal node + the root's third branch
        site = np.ones(n_pos)
        total = 0.0
        for v in G.internal_nodes(d) + [root]:
            a, b = int(child[v, 0]), int(child[v, 1])
            length = float(orc.dt.type(bl[a]) + orc.dt.type(bl[b]))  # numeric_t sum, NJ.tcc:5124
            total += orc.pair_loglk(profs[a], profs[b], length, d["ml.rates"], d["ml.ratecat"], tm, min_rel, site)
        c = int(child[root, 2])
        ab = orc.posterior_profile(profs[int(child[root, 0])], profs[int(child[root, 1])],
                                   float(bl[int(child[root, 0])]), float(bl[int(child[root, 1])]), d["ml.rates"],
                                   d["ml.ratecat"], tm, min_len, min_rel)
        total += orc.pair_loglk(ab, profs[c], float(bl[c]), d["ml.rates"], d["ml.ratecat"], tm, min_rel, site)
        site_loglk = np.log(site)
        if model == "jc":
            gaps_per_pos = (d["leaf.codes"] == G.NOCODE).sum(axis=0)
            total += (gaps_per_pos.sum() - n_pos) * np.log(4.0)
            site_loglk += (gaps_per_pos - 1) * np.log(4.0)
        assert total == pytest.approx(float(d[model + ".treeloglk"]), rel=1e-12)
        assert np.allclose(site_loglk, d[model + ".site_loglk"], rtol=1e-9, atol=1e-9)

        # loose pairLogLk calls with per-site likelihoods
        for k in range(len(d[model + ".pll.a"])):
            a, b = int(d[model + ".pll.a"][k]), int(d[model + ".pll.b"][k])
            s = np.ones(n_pos)
            v = orc.pair_loglk(profs[a], profs[b], float(d[model + ".pll.len"][k]), d["ml.rates"], d["ml.ratecat"],
                               tm, min_rel, s)
            assert v == float(d[model + ".pll.val"][k]), (model, k)
            assert np.array_equal(s, d[model + ".pll.site"][k])
        # loose posteriorProfile calls (including clamped tiny lengths)
        for k in range(len(d[model + ".post.a"])):
            a, b = int(d[model + ".post.a"][k]), int(d[model + ".post.b"][k])
            o = orc.posterior_profile(profs[a], profs[b], float(d[model + ".post.len1"][k]),
                                      float(d[model + ".post.len2"][k]), d["ml.rates"], d["ml.ratecat"], tm,
                                      min_len, min_rel)
            assert G.profiles_equal(o, G.fixture_profile(d, "%s.post%d" % (model, k))), (model, k)


def test_knuth_ran_array_restatement_is_pinned():
    """oracle/vft_knuth.h against Knuth's published check value (TAOCP 3.6, rng.c) and against 5000 values of the
    reference's own stream (tests/golden/wb_knuth.npz, oracle/whitebox.cpp mode `knuth`)."""
    import oracle as O
    assert O.knuth_selftest(2009, 1009) == 995235265
    assert O.knuth_selftest(1009, 2009) == 995235265
    ref = G.load("wb_knuth")["knuth.rand"]
    assert np.array_equal(O.knuth_stream(len(ref)), ref)


def _ml_tolerances(dtype):
    """(MLFTolBranchLength, MLMinBranchLengthTolerance): Constants.h:26-30."""
    return (0.001, 1.0e-4) if np.dtype(dtype) == np.float32 else (0.001, 1.0e-9)


def test_ml_branch_length_rounds(fx):
    """optimizeAllBranchLengths (NJ.tcc:5065) restated in oracle/ml_lengths.py: two rounds from the fixture's ML state
    reproduce the reference's branch lengths and tree likelihoods."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import ml_lengths as ML
    d, orc = fx["d"], fx["orc"]
    min_len, min_rel, _ = tolerances(orc.dt)
    ftol, atol = _ml_tolerances(orc.dt)
    child, n_child, parent, root = d["nj.child"], d["nj.nchild"], d["nj.parent"], int(d["nj.root"])
    for model in _models(fx):
        profs, tm = _ml_profiles(fx, model)
        profs.append(None)   # the root has no profile
        bl = d["nj.branchlength"].astype(orc.dt).copy()
        for rnd in (1, 2):
            ML.optimize_all_branch_lengths(orc, profs, child, n_child, parent, root, bl, d["ml.rates"], d["ml.ratecat"],
                                           tm, min_len, min_rel, ftol, atol)
            want = d["%s.opt%d.branchlength" % (model, rnd)]
            # the line searches follow the reference evaluation by evaluation: lengths agree to rounding
            assert np.allclose(bl[:root], want[:root], rtol=1e-5, atol=1e-9), (model, rnd)
            ll = ML.tree_loglk(orc, profs, child, root, bl, d["ml.rates"], d["ml.ratecat"], tm, min_len, min_rel,
                               d["leaf.codes"])
            assert ll == pytest.approx(float(d["%s.opt%d.treeloglk" % (model, rnd)]), rel=1e-7), (model, rnd)


def test_quartet_likelihoods_of_split_tests(fx):
    """MLQuartetLogLk / MLQuartetOptimize as testSplitsML runs them (oracle/ml_lengths.py) against the reference's three
    log-likelihoods and optimised lengths for six splits per model."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import ml_lengths as ML
    d, orc = fx["d"], fx["orc"]
    min_len, min_rel, _ = tolerances(orc.dt)
    ftol, atol = _ml_tolerances(orc.dt)
    for model in _models(fx):
        profs, tm = _ml_profiles(fx, model)
        for k, node in enumerate(d[model + ".quartet.nodes"]):
            key = "%s.quartet%d" % (model, k)
            a, b, c, _ = [int(x) for x in d[key + ".abcd"]]
            pd = G.fixture_profile(d, key + ".D")
            loglk, l_ac, l_ad = ML.split_test(orc, profs[a], profs[b], profs[c], pd, [float(x) for x in d[key + ".len"]],
                                              d["ml.rates"], d["ml.ratecat"], tm, min_len, min_rel, ftol, atol)
            assert np.allclose(loglk, d[key + ".loglk"], rtol=1e-9, atol=0), (model, k, loglk, d[key + ".loglk"])
            assert np.allclose(l_ac, d[key + ".lenAC"], rtol=1e-6, atol=1e-12), (model, k)
            assert np.allclose(l_ad, d[key + ".lenAD"], rtol=1e-6, atol=1e-12), (model, k)


@pytest.mark.parametrize("name", ["wb_nt_f32", "wb_nt_f32_gappy"])
def test_avx2_sweep_equals_scalar_oracle(name):
    """bench.py's CPU baseline (oracle/vft_oracle_avx2.c: AVX2 + OpenMP restatement of the one-vs-all sweep) against the
    scalar oracle - which is pinned to the reference's own sweeps above - on the initial state (leaf x leaf) and on three
    mid-run states (leaf and internal seeds, internal targets with vectors), bit for bit, at 1 and at 4 threads."""
    d = G.load(name)
    orc = Oracle(np.float32)
    n_seqs = int(d["nSeqs"])
    profs = G.build_nj_profiles(d, orc)
    rng = np.random.default_rng(3)
    for mid in ("mid0", "mid1", "mid2"):
        J, n_active = int(d[mid + ".J"]), int(d[mid + ".nActive"])
        lim = n_seqs + J
        parent = G.mid_parent(d, J)
        active = np.nonzero(parent < 0)[0]
        W, Cc, F = G.pack(profs[:lim], orc.dt)
        outp, _ = orc.out_profile(W[active], Cc[active], F[active], None, 1e-10)
        st = orc.state(n_seqs, W, Cc, F, parent, d["nj.diameter"][:lim], d["nj.selfweight"][:lim], d["nj.selfdist"][:lim],
                       float(d[mid + ".totdiam"]), outp)
        od = d[mid + ".outdist_fresh"][:lim].astype(np.float32)
        na = np.full(lim, n_active, np.int64)
        na[rng.integers(0, lim, 5)] = n_active + 3          # a few stale stamps: the rescaled criterion of NJ.tcc:1099-1107
        queries = [int(q) for q in d[mid + ".queries"]] + [int(active[0]), int(active[-1])]
        for q in queries:
            want = orc.set_best_hit(st, q, n_active, 10 ** 9, od, na)    # nDiffAllow huge: no lazy refresh on either side
            for threads in (1, 4):
                got = orc.avx2_sweep(st, q, n_active, od, na, threads=threads)
                for key in ("dist", "weight", "crit"):
                    assert np.array_equal(got[key], want[key]), (name, mid, q, key, threads)
        # the timed leg of bench.py (its own first-touched copy of the state, static blocks): same bits; the last sweep it
        # returns is the last query's
        secs, done, got = orc.avx2_sweep_bench(st, queries, n_active, od, na, threads=3, budget=0.0)
        assert done == len(queries) and secs > 0
        want = orc.set_best_hit(st, queries[-1], n_active, 10 ** 9, od, na)
        for key in ("dist", "weight", "crit"):
            assert np.array_equal(got[key], want[key]), (name, mid, key, "bench leg")
