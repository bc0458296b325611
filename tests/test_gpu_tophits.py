"""The top-hit lists on the device (vft_tophits_*, veryfasttree_amd/csrc/vft_kernels_tophits.h) against the same walks
restated on the host from calls that are already pinned to the reference: vft_pair_distances_refresh (setDistCriterion with
the forced / lazy out-distance refreshes) for the distances and criteria, numpy for the list logic (updateBestHit's
re-targeting NJ.tcc:1626-1648, the first strict minimum of getBestFromTopHits :4286-4295, the distinct candidates of
uniqueBestHits :4786-4833, sortSaveBestHits' order :4535-4578 = criterion ascending, ties by descending partner id)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _state(dt, seed=31):
    """two contexts in the same mid-NJ state: 400 leaves, 200 random joins, scattered stale out-distance stamps"""
    from veryfasttree_amd import HipProfileOps, synth
    n, L, nj = 400, 190, 200
    codes = synth.random_descent_codes(n, L, 4, 0.05, 0.05, seed=seed)
    ctxs = []
    for _ in range(2):
        ops = HipProfileOps(n, L, 4, dt, max_nodes=2 * n)
        ops.upload_leaves(codes)
        ops.set_node_scalars(0, np.zeros(n, dt), (codes != 127).sum(1).astype(dt), np.zeros(n, dt))
        ops.outProfile(np.arange(n))
        active = list(range(n))
        parent = np.full(2 * n, -1, np.int64)
        order = np.random.default_rng(6)
        for k in range(nj):
            i, j = (int(x) for x in order.choice(len(active), 2, replace=False))
            i, j = active[i], active[j]
            ops.join_fused(i, j, n + k, 0.02 * (k % 5), 10 * n, len(active), True)
            parent[i] = parent[j] = n + k
            active = [v for v in active if v != i and v != j] + [n + k]
        nact = len(active)
        rng = np.random.default_rng(12)
        stamps = np.full(n + nj, nact, np.int64)
        stamps[rng.choice(active, 60, replace=False)] = nact + 50      # staler than allowed (allow = 5)
        stamps[rng.choice(active, 30, replace=False)] = nact + 3       # stale but allowed: rescaled in the criterion
        ops.set_out_distances(0, np.random.default_rng(2).uniform(0, 30, n + nj).astype(dt), stamps)
        ctxs.append(ops)
    return ctxs, np.array(active, np.int64), parent, nact, n, nj


def _ancestor(parent, v):
    while parent[v] >= 0:
        v = parent[v]
    return v


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_tophits_best_equals_the_host_walk(dt):
    (o1, o2), active, parent, nact, n, nj = _state(dt)
    m, allow, totdiam = 64, 5, 2.5
    o1.tophits_create(m)
    rng = np.random.default_rng(3)
    for rep in range(12):
        node = int(active[rng.integers(len(active))])
        # partners: any node ever created (joined ones are re-targeted to their active ancestor), some duplicates after
        # re-targeting, the node's own descendants (ancestor == node: dropped), an empty slot
        ln = int(rng.integers(5, m + 1))
        js = rng.integers(0, n + nj, ln).astype(np.int32)
        js[js == node] = int(active[0]) if node != active[0] else int(active[1])
        if rep % 3 == 0:
            js[1] = -1
        stored = rng.uniform(0.01, 0.5, ln).astype(dt)
        o1.tophits_upload([node], [(js, stored)])
        got_j, got_d = o1.tophits_download(node)
        assert np.array_equal(got_j, js) and np.array_equal(got_d, stored)
        j1, pos, d1, c1 = o1.tophits_best(node, ln, nact, allow, totdiam, force_node=True)
        # the host walk on the second context
        anc = np.array([_ancestor(parent, int(v)) if v >= 0 else -1 for v in js], np.int64)
        valid = (anc >= 0) & (anc != node)
        retarget = valid & (anc != js)
        pi = np.full(int(retarget.sum()), node, np.int64)
        dist = stored.copy()
        crit = np.full(ln, 1e20, dt)
        dd, _, cc = o2.setDistCriterionRefresh(pi, anc[retarget], np.array([node], np.int64), nact, allow, totdiam)
        dist[retarget] = dd
        crit[retarget] = cc
        # entries whose partner is unchanged: setCriterion on the stored distance (lazy refresh of both ends, arithmetic)
        keep = valid & ~retarget
        if keep.any():
            o2.setDistCriterion(np.full(int(keep.sum()), node, np.int64), anc[keep], nact, allow, totdiam)   # (for its lazy refreshes)
            od, na = o2.get_out_distances(0, n + nj)
            for t in np.nonzero(keep)[0]:
                oi, oj = float(od[node]), float(od[anc[t]])
                if na[node] != nact:
                    oi *= (nact - 1) / float(na[node] - 1)
                if na[anc[t]] != nact:
                    oj *= (nact - 1) / float(na[anc[t]] - 1)
                crit[t] = dt(float(dist[t]) - (oi + oj) / float(nact - 2))
        best = -1
        for t in range(ln):
            if valid[t] and crit[t] < (crit[best] if best >= 0 else dt(1e20)):
                best = t
        assert best >= 0
        assert (j1, pos) == (int(anc[best]), best), (rep, j1, pos, int(anc[best]), best)
        assert d1 == dist[best] and c1 == crit[best]
        # every refresh the walk made is the refresh the host path makes
        assert all(np.array_equal(x, y) for x, y in zip(o1.get_out_distances(0, n + nj), o2.get_out_distances(0, n + nj))), rep
    o1.close()
    o2.close()


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_tophits_join_equals_the_host_merge(dt):
    (o1, o2), active, parent, nact, n, nj = _state(dt, seed=32)
    m, allow, totdiam = 96, 5, 2.5
    o1.tophits_create(m)
    rng = np.random.default_rng(4)
    newnode = int(active[-1])                       # the most recent join: its children carry the lists
    c0, c1 = (int(v) for v in np.nonzero(parent == newnode)[0])
    for rep in range(10):
        n0, n1 = int(rng.integers(3, m + 1)), int(rng.integers(3, m + 1))
        l0 = rng.integers(0, n + nj, n0).astype(np.int32)
        l1 = rng.integers(0, n + nj, n1).astype(np.int32)
        k = min(n0, n1) // 3
        l1[:k] = l0[:k]                             # shared partners
        o1.tophits_upload([c0, c1], [(l0, rng.uniform(0, 1, n0).astype(dt)), (l1, rng.uniform(0, 1, n1).astype(dt))])
        # the new node as a join leaves it: "unreasonably" stale
        for o in (o1, o2):
            od, na = o.get_out_distances(newnode, 1)
            o.set_out_distances(newnode, od, np.array([10 * n], np.int64))
        need, age_ok, save_max = (10 ** 6, True, m) if rep % 4 == 3 else (8, rep % 4 != 2, m if rep % 2 else m // 3)
        res = o1.tophits_join(newnode, c0, n0, c1, n1, nact, allow, totdiam, save_max, need, age_ok)
        cand = sorted({_ancestor(parent, int(v)) for v in np.concatenate([l0, l1]) if v >= 0} - {newnode})
        cand = np.array(cand, np.int64)
        d2, _, c2 = o2.setDistCriterion(np.full(len(cand), newnode, np.int64), cand, nact, allow, totdiam)
        # ascending criterion, ties by descending position in the ascending-id list
        order = sorted(range(len(cand)), key=lambda t: (c2[t] + 0.0, -t))
        assert res["n_unique"] == len(cand)
        assert np.array_equal(res["j"], cand[order]), rep
        assert np.array_equal(res["dist"], d2[order]) and np.array_equal(res["criterion"], c2[order])
        use = len(cand) == nact - 1 or (age_ok and len(cand) >= need)
        assert res["use_unique"] == use
        if use:
            ns = min(len(cand), save_max)
            assert res["n_save"] == ns
            gj, gd = o1.tophits_download(newnode)
            assert np.array_equal(gj, cand[order][:ns]) and np.array_equal(gd, d2[order][:ns])
        assert all(np.array_equal(x, y) for x, y in zip(o1.get_out_distances(0, n + nj), o2.get_out_distances(0, n + nj))), rep
    o1.close()
    o2.close()


@pytest.mark.parametrize("name,fastest,second", [("bb_nt_200", False, False), ("bb_nt_1500", False, False),
                                                 ("bb_nt_600_fastest", True, True), ("bb_nt_600_fastest_no2nd", True, False)])
def test_driver_cross_checks_every_device_walk_against_the_host_walk(name, fastest, second, monkeypatch):
    """VFT_NJ_CHECK=1: the C++ driver runs the host walk of round 2 behind every device walk (getBestFromTopHits, the merge of
    a join) and throws at the first difference; the join order must still be the reference's."""
    import golden_util as G
    from test_nj_driver_cpu import unique_codes
    from veryfasttree_amd import HipProfileOps
    from veryfasttree_amd.backend import nj_run
    monkeypatch.setenv("VFT_NJ_CHECK", "1")
    d = G.load(name)
    codes = unique_codes(d["codes"])
    ops = HipProfileOps(codes.shape[0], codes.shape[1], 4, np.float32)
    joins, crit = nj_run(ops, codes, fastest=fastest, second_level=second)
    assert np.array_equal(joins, d["joins"])
    ops.close()


@pytest.mark.parametrize("fastest,wide", [(False, False), (True, False), (False, True)])
def test_join_engine_equals_the_host_driven_loop(fastest, wide, monkeypatch):
    """The join loop on the device (vft_nj_engine_*: kernels that take their arguments from a device-resident state block,
    the host only handling resets and refreshes) against the host-driven loop of the same driver on a 6 000 x 150 alignment:
    every join (i, j, new node) and every criterion, bit for bit.  (The fixtures of test_gpu_nj_driver.py pin both to the
    reference.)  wide: the 1 024-thread instance of the glue kernel, which lists of more than 1 024 hits (beyond a million
    sequences) select, forced at this size (VFT_DEBUG_WIDE_GLUE)."""
    from veryfasttree_amd import HipProfileOps, synth
    from veryfasttree_amd.backend import nj_run
    codes = synth.random_descent_codes(6000, 150, 4, 0.04, 0.02, seed=77)
    codes = codes[np.sort(np.unique(codes, axis=0, return_index=True)[1])]
    runs = []
    from veryfasttree_amd.backend import DEBUG_HOST_JOINS
    for host in (False, True):
        ops = HipProfileOps(codes.shape[0], codes.shape[1], 4, np.float32)
        if wide and not host:
            ops.debug_option(7, 1)
        runs.append(nj_run(ops, codes, fastest=fastest, second_level=False, debug_flags=DEBUG_HOST_JOINS if host else 0))
        ops.close()
    assert np.array_equal(runs[0][0], runs[1][0])
    assert np.array_equal(runs[0][1], runs[1][1])


def test_seed_sweeps_taken_ahead_give_the_seed_by_seed_lists():
    """setAllLeafTopHits asks for the sweeps of the next eight unvisited seeds in one vft_sweep_batch (its leaf seeds share passes over
    the targets); a sweep taken ahead whose seed becomes a close neighbour first is dropped.  Every join and criterion equals the run
    with one vft_sweep per seed (vft_nj_options.debug_flags & VFT_NJ_DEBUG_SEED_BY_SEED), with and without second-level lists."""
    from veryfasttree_amd import HipProfileOps, synth
    from veryfasttree_amd.backend import nj_run
    codes = synth.random_descent_codes(5000, 150, 4, 0.04, 0.02, seed=78)
    codes = codes[np.sort(np.unique(codes, axis=0, return_index=True)[1])]
    for fastest, second in ((False, False), (True, True)):
        runs = []
        for flags in (0, 256):
            ops = HipProfileOps(codes.shape[0], codes.shape[1], 4, np.float32)
            runs.append(nj_run(ops, codes, fastest=fastest, second_level=second, debug_flags=flags))
            ops.close()
        assert np.array_equal(runs[0][0], runs[1][0])
        assert np.array_equal(runs[0][1], runs[1][1])
