"""Bit-level census of the ML operators under matrix models: how many per-site likelihoods (pairLogLk) and posterior
vector entries of the HIP kernels are bit-identical to the CPU oracle's (which itself equals the reference bit for bit,
tests/test_oracle_golden.py).  Prints the census; the asserted bar is stated per check."""
import numpy as np
import pytest

import golden_util as G
from oracle import Oracle, tolerances

pytestmark = pytest.mark.gpu


def _ulps(a, b):
    a, b = np.asarray(a), np.asarray(b)
    it = np.int32 if a.dtype == np.float32 else np.int64
    return np.abs(a.view(it).astype(np.int64) - b.view(it).astype(np.int64))


@pytest.mark.parametrize("rows", [False, True])
@pytest.mark.parametrize("name", ["wb_nt_f32", "wb_nt_f64", "wb_aa_f32", "wb_aa_f64"])
def test_matrix_model_operators_bit_census(name, rows):
    """rows: the context in row mode (vft_set_profile_rows, the ML stage's layout) - posteriors write dense rows directly, proteins through
    the quad-of-lanes kernels (k_posterior_quad, k_pair_loglk_quad); otherwise tile streams through stash + commit and the whole-column
    kernels for the posteriors."""
    from veryfasttree_amd import HipProfileOps
    d = G.load(name)
    dt = G.dtype_of(d)
    orc = Oracle(dt)
    model = "lg" if "_aa_" in name else "gtr"
    n_seqs, n_pos, n_codes, root = int(d["nSeqs"]), int(d["nPos"]), int(d["nCodes"]), int(d["nj.root"])
    min_len, min_rel, _ = tolerances(dt)
    tm = G.tmat_of(d, orc, model)
    bl, child = d["nj.branchlength"], d["nj.child"]
    nodes = G.internal_nodes(d)
    level = np.zeros(root + 1, np.int64)
    for v in nodes:
        level[v] = 1 + max(level[child[v, 0]], level[child[v, 1]])
    # oracle profiles
    profs = [orc.leaf_profile(d["leaf.codes"][i], n_codes) for i in range(n_seqs)]
    for v in nodes:
        a, b = int(child[v, 0]), int(child[v, 1])
        profs.append(orc.posterior_profile(profs[a], profs[b], float(bl[a]), float(bl[b]), d["ml.rates"], d["ml.ratecat"], tm,
                                           min_len, min_rel))
    ops = HipProfileOps(n_seqs, n_pos, n_codes, dt)
    ops.upload_leaves(d["leaf.codes"])
    ops.set_max_node(min(root + 1, ops.max_nodes))
    if rows:
        ops.set_profile_rows(True)
    ops.set_rates(d["ml.rates"], d["ml.ratecat"])
    ops.set_ml_limits(*tolerances(dt))
    k = model + ".tm."
    ops.set_transition_matrix(d[k + "stat"], d[k + "statinv"], d[k + "eigenval"], d[k + "codefreq"], d[k + "eigeninv"], d[k + "eigeninvT"])
    for lv in range(1, int(level.max()) + 1):
        batch = np.array([v for v in nodes if level[v] == lv])
        a, b = child[batch, 0], child[batch, 1]
        ops.posteriorProfile(batch, a, b, bl[a].astype(np.float64), bl[b].astype(np.float64))
        nbad = ntot = 0
        worst = 0
        for v in batch:
            got, want = ops.profile_download(int(v)), profs[int(v)]
            assert np.array_equal(got[1], want[1])
            has = (want[0] > 0) & (want[1] == G.NOCODE)
            u = _ulps(got[2][has], want[2][has])
            uw = _ulps(got[0], want[0])
            nbad += int((u > 0).sum()) + int((uw > 0).sum())
            ntot += u.size + uw.size
            worst = max(worst, int(u.max()) if u.size else 0, int(uw.max()))
        print("%s %s level %d: %d nodes, %d of %d values differ, worst %d ulp" % (name, model, lv, len(batch), nbad, ntot, worst))
        if lv == 1:
            lvl1_bad = nbad
            total_bad = 0
        total_bad += nbad
    # per-site likelihoods of leaf x leaf pairs (exact inputs): sum_j expeigen[j] cf[a][j] cf[b][j]
    rng = np.random.default_rng(5)
    a = rng.integers(0, n_seqs, 64)
    b = rng.integers(0, n_seqs, 64)
    length = rng.uniform(0.001, 1.5, 64)
    ll, site = ops.pairLogLk(a, b, length, site_lk=True)
    bad = tot = 0
    worst = 0
    for t in range(64):
        s = np.ones(n_pos)
        v = orc.pair_loglk(profs[int(a[t])], profs[int(b[t])], float(length[t]), d["ml.rates"], d["ml.ratecat"], tm, min_rel, s)
        u = _ulps(site[t], s)
        bad += int((u > 0).sum())
        tot += u.size
        worst = max(worst, int(u.max()))
        assert abs(ll[t] - v) <= 1e-9 * abs(v)
    print("%s %s leaf x leaf site likelihoods: %d of %d differ, worst %d ulp (double)" % (name, model, bad, tot, worst))
    ops.close()
    # the P(t) tables use glibc's exp (vft_glibc_log.h) and every other step keeps the reference's arithmetic, so both
    # precisions are bit-identical
    assert bad == 0, "per-site likelihoods of exact inputs must be bit-identical"
    assert lvl1_bad == 0, "posteriors of leaf children must be bit-identical"
    assert total_bad == 0, "every posterior profile must be bit-identical"
