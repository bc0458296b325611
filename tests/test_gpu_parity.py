"""GPU parity: the HIP backend, called through the C ABI (veryfasttree_amd.backend -> include/vft_hip.h),
against (a) the golden vectors dumped from the compiled reference and (b) the CPU oracle on seeded inputs.

Bar: bit-exact for everything on the NJ path (integer counts, numeric_t distances/criteria, sort order,
profiles); ML profiles bit-exact except where a device exp() differs from glibc's by an ulp before narrowing
(tolerance 2e-6 relative in float, written below); log-likelihoods 1e-9 relative (north star: 1e-4)."""
import numpy as np
import pytest

import golden_util as G
from oracle import Oracle, tolerances

pytestmark = pytest.mark.gpu


def _ops(d, **kw):
    from veryfasttree_amd import HipProfileOps
    ops = HipProfileOps(int(d["nSeqs"]), int(d["nPos"]), int(d["nCodes"]), G.dtype_of(d), **kw)
    ops.upload_leaves(d["leaf.codes"])
    if "dmat.distances" in d:
        ops.set_distance_matrix(d["dmat.distances"], d["dmat.codefreq"], d["dmat.eigenval"], d["dmat.eigentot"])
    return ops


def _build_nj_profiles(d, ops):
    """averageProfile for every join of the reference's NJ tree, in creation order, batched by dependency level."""
    n_seqs = int(d["nSeqs"])
    child = d["nj.child"]
    level = np.zeros(int(d["nj.root"]) + 1, np.int64)
    nodes = G.internal_nodes(d)
    for v in nodes:
        level[v] = 1 + max(level[child[v, 0]], level[child[v, 1]])
    for lv in range(1, int(level.max()) + 1):
        batch = [v for v in nodes if level[v] == lv]
        if batch:
            ops.averageProfile(batch, child[batch, 0], child[batch, 1])
    ops.set_max_node(int(d["nj.root"]))
    return n_seqs


@pytest.fixture(scope="module", params=G.WHITEBOX)
def fx(request):
    d = G.load(request.param)
    ops = _ops(d)
    _build_nj_profiles(d, ops)
    orc = Oracle(G.dtype_of(d))
    yield dict(name=request.param, d=d, ops=ops, orc=orc)
    ops.close()


def test_average_profiles_match_reference(fx):
    d, ops, orc = fx["d"], fx["ops"], fx["orc"]
    n_seqs = int(d["nSeqs"])
    nodes = G.internal_nodes(d)
    got = np.array([orc.profile_hash(ops.profile_download(v)) for v in nodes], dtype=np.int64)
    assert np.array_equal(got, d["nj.profiles.hash"][n_seqs:])
    for k in range(5):
        node = int(d["nj.full%d.node" % k])
        assert G.profiles_equal(ops.profile_download(node), G.fixture_profile(d, "nj.full%d" % k))
    # self-distances computed on the device at join time (NJ.tcc:3039-3042)
    _, sw, sd = ops.get_node_scalars(n_seqs, len(nodes))
    assert np.array_equal(sd, d["nj.selfdist"][n_seqs:n_seqs + len(nodes)])
    assert np.array_equal(sw, d["nj.selfweight"][n_seqs:n_seqs + len(nodes)])


def test_leaf_profile_download(fx):
    d, ops, orc = fx["d"], fx["ops"], fx["orc"]
    for i in (0, int(d["nSeqs"]) - 1):
        assert G.profiles_equal(ops.profile_download(i), orc.leaf_profile(d["leaf.codes"][i], int(d["nCodes"])))


def test_pair_distances(fx):
    d, ops = fx["d"], fx["ops"]
    n = int(d["nj.root"])
    z = np.zeros(n, ops.dt)
    ops.set_node_scalars(0, diameter=z)
    ops.set_out_distances(0, z, np.full(n, 5))
    a, b = d["pdist.a"], d["pdist.b"]
    dist, weight, _ = ops.setDistCriterion(a, b, 5, 10 ** 6, 0.0)
    n_seqs = int(d["nSeqs"])
    both_leaves = (a < n_seqs) & (b < n_seqs)
    ok = ~(both_leaves & (d["pdist.weight"] == ops.dt.type(0.01)))   # seqDist reports weight 0 there, see NJ.tcc:1621
    assert np.array_equal(dist[ok], d["pdist.dist"][ok])
    assert np.array_equal(weight[ok], d["pdist.weight"][ok])
    # the excluded pairs are two leaves without a common column: the fixture's numbers come from profileDist (weight 0.01,
    # NJ.tcc:1186-1188), the path under test goes through seqDist like the reference's own callers (NJ.tcc:1601-1624),
    # whose answer for them is distance 1, weight 0
    assert np.all(weight[~ok] == 0) and np.all(dist[~ok] == 1)
    # leaf x leaf through seqDist
    dist, weight, _ = ops.setDistCriterion(d["seqdist.a"], d["seqdist.b"], 5, 10 ** 6, 0.0)
    assert np.array_equal(dist, d["seqdist.dist"])
    assert np.array_equal(weight, d["seqdist.weight"])


def _check_sweep(d, ops, key, node, n_active, n_diff_allow, totdiam, maxnode):
    k = min(maxnode, 4096)
    hits, best = ops.setBestHit(node, n_active, n_diff_allow, totdiam, k)
    dist, weight, crit = ops.sweep_results(0, maxnode)
    act = d[key + ".i"] >= 0
    assert np.array_equal(dist, d[key + ".dist"])
    assert np.array_equal(crit, d[key + ".crit"])
    assert np.array_equal(weight[act], d[key + ".weight"][act])
    assert best == int(d[key + ".best"][1])
    na = int(act.sum())
    want = d[key + ".sorted_j"][:min(na, k)]
    assert np.array_equal(hits["j"][:len(want)], want)
    assert np.array_equal(hits["criterion"][:len(want)], d[key + ".crit"][want])
    assert np.array_equal(hits["dist"][:len(want)], d[key + ".dist"][want])
    assert np.all(hits["j"][na:] == -1)
    od, nout = ops.get_out_distances(0, maxnode)
    assert np.array_equal(od[act], d[key + ".outdist_after"][:maxnode][act])
    assert np.array_equal(nout[act], d[key + ".noutactive_after"][:maxnode][act])


def test_initial_state_and_leaf_sweeps(fx):
    d, ops = fx["d"], fx["ops"]
    n_seqs, total = int(d["nSeqs"]), int(d["nj.root"])
    ops.set_max_node(n_seqs)
    ops.set_parents(0, np.full(total, -1))
    ops.outProfile(np.arange(n_seqs))
    outp, cd = ops.out_profile_download("dmat.distances" in d)
    assert G.profiles_equal(outp, G.fixture_profile(d, "init.outprofile"))
    if cd is not None:
        assert np.array_equal(cd, d["init.outprofile.cd"])
    z = np.zeros(n_seqs, ops.dt)
    ops.set_node_scalars(0, z, d["init.selfweight"], z)
    ops.set_out_distances(0, z, np.full(n_seqs, 10 * n_seqs))
    ops.setOutDistance(None, n_seqs, 0.0)
    od, _ = ops.get_out_distances(0, n_seqs)
    assert np.array_equal(od, d["init.outdist"])
    for s, q in enumerate(d["init.queries"]):
        ops.set_out_distances(0, d["init.outdist"], np.full(n_seqs, n_seqs))
        _check_sweep(d, ops, "init.sweep%d" % s, int(q), n_seqs, int(n_seqs * 0.01), 0.0, n_seqs)
    ops.set_max_node(total)


@pytest.mark.parametrize("mid", ["mid0", "mid1", "mid2"])
def test_mid_run_state(fx, mid):
    d, ops = fx["d"], fx["ops"]
    n_seqs, total = int(d["nSeqs"]), int(d["nj.root"])
    J, n_active = int(d[mid + ".J"]), int(d[mid + ".nActive"])
    lim = n_seqs + J
    parent = np.full(total, 0, np.int64)
    parent[:lim] = G.mid_parent(d, J)
    ops.set_parents(0, parent)
    ops.set_max_node(lim)
    ops.set_node_scalars(0, d["nj.diameter"][:lim], d["nj.selfweight"][:lim], d["nj.selfdist"][:lim])
    ops.outProfile(d[mid + ".active"])
    outp, cd = ops.out_profile_download("dmat.distances" in d)
    assert G.profiles_equal(outp, G.fixture_profile(d, mid + ".outprofile"))
    if cd is not None:
        assert np.array_equal(cd, d[mid + ".outprofile.cd"])
    totdiam = float(d[mid + ".totdiam"])
    ops.set_out_distances(0, np.zeros(lim, ops.dt), np.full(lim, 10 * n_seqs))
    ops.setOutDistance(d[mid + ".active"], n_active, totdiam)
    od, _ = ops.get_out_distances(0, lim)
    act = parent[:lim] < 0
    assert np.array_equal(od[act], d[mid + ".outdist_fresh"][act])
    for s, q in enumerate(d[mid + ".queries"]):
        ops.set_out_distances(0, d[mid + ".outdist_in"], d[mid + ".noutactive_in"])
        _check_sweep(d, ops, "%s.sweep%d" % (mid, s), int(q), n_active, int(d[mid + ".nDiffAllow"]), totdiam, lim)
    # pair-list path (transferBestHits): same pairs as the first sweep, gathered
    key = mid + ".sweep0"
    q = int(d[mid + ".queries"][0])
    targets = d[mid + ".active"]
    ops.set_out_distances(0, d[mid + ".outdist_in"], d[mid + ".noutactive_in"])
    dist, weight, crit = ops.setDistCriterion(np.full(len(targets), q), targets, n_active, int(d[mid + ".nDiffAllow"]),
                                              totdiam)
    assert np.array_equal(dist, d[key + ".dist"][targets])
    assert np.array_equal(weight, d[key + ".weight"][targets])
    assert np.array_equal(crit, d[key + ".crit"][targets])
    if mid + ".update_abn" in d:
        a, b, n = [int(x) for x in d[mid + ".update_abn"]]
        ops.outProfile(d[mid + ".active"])
        ops.updateOutProfile(a, b, n, n_active)
        upd, ucd = ops.out_profile_download("dmat.distances" in d)
        assert G.profiles_equal(upd, G.fixture_profile(d, mid + ".outprofile_updated"))
        if ucd is not None:
            assert np.array_equal(ucd, d[mid + ".outprofile_updated.cd"])
    ops.set_max_node(total)


def _close(a, b, dt):
    tol = 2e-6 if np.dtype(dt) == np.float32 else 1e-13
    return np.allclose(a, b, rtol=tol, atol=tol * 1e-3)


def test_ml_posterior_levels_and_tree_loglk(fx):
    """recomputeMLProfiles level by level + treeLogLk, against the reference's numbers."""
    d, name = fx["d"], fx["name"]
    models = ["lg"] if "_aa_" in name else ["jc", "gtr"]
    n_seqs, n_pos, root = int(d["nSeqs"]), int(d["nPos"]), int(d["nj.root"])
    bl, child = d["nj.branchlength"], d["nj.child"]
    nodes = G.internal_nodes(d)
    level = np.zeros(root + 1, np.int64)
    for v in nodes:
        level[v] = 1 + max(level[child[v, 0]], level[child[v, 1]])
    for model in models:
        ops = _ops(d)
        ops.set_max_node(min(root + 1, ops.max_nodes))
        ops.set_rates(d["ml.rates"], d["ml.ratecat"])
        ops.set_ml_limits(*tolerances(ops.dt))
        if model != "jc":
            k = model + ".tm."
            ops.set_transition_matrix(d[k + "stat"], d[k + "statinv"], d[k + "eigenval"], d[k + "codefreq"],
                                      d[k + "eigeninv"], d[k + "eigeninvT"])
        for lv in range(1, int(level.max()) + 1):
            batch = np.array([v for v in nodes if level[v] == lv])
            a, b = child[batch, 0], child[batch, 1]
            ops.posteriorProfile(batch, a, b, bl[a].astype(np.float64), bl[b].astype(np.float64))
        exact = 0
        for k in range(3):
            node = int(d["%s.full%d.node" % (model, k)])
            got, want = ops.profile_download(node), G.fixture_profile(d, "%s.full%d" % (model, k))
            assert np.array_equal(got[1], want[1])
            assert _close(got[0], want[0], ops.dt)
            has = (want[0] > 0) & (want[1] == G.NOCODE)
            assert _close(got[2][has], want[2][has], ops.dt)
            exact += int(G.profiles_equal(got, want))
        # treeLogLk: one pairLogLk per internal node (NJ.tcc:5123) + the root's third branch (NJ.tcc:5146-5148)
        allp = nodes + [root]
        a, b = child[allp, 0], child[allp, 1]
        length = (bl[a] + bl[b]).astype(np.float64)   # numeric_t sum
        ll, site = ops.pairLogLk(a, b, length, site_lk=True)
        r0, r1, r2 = [int(x) for x in child[root]]
        ops.posteriorProfile([root], [r0], [r1], [float(bl[r0])], [float(bl[r1])])
        ll3, site3 = ops.pairLogLk([root], [r2], [float(bl[r2])], site_lk=True)
        total = float(ll.sum() + ll3[0])
        site_loglk = np.log(site).sum(axis=0) + np.log(site3[0])
        if model == "jc":
            gaps = (d["leaf.codes"] == G.NOCODE).sum(axis=0)
            total += (gaps.sum() - n_pos) * np.log(4.0)
            site_loglk += (gaps - 1) * np.log(4.0)
        assert total == pytest.approx(float(d[model + ".treeloglk"]), rel=1e-9)
        assert np.allclose(site_loglk, d[model + ".site_loglk"], rtol=1e-6, atol=1e-6)
        ops.close()
