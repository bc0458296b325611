"""GPU: the packed tile streams (vft_layout.h) survive writes in any order.

Profiles are uploaded into random lanes of several tiles in random order, some of them several times with a different
vector / explicit-weight pattern, and every node must read back exactly as last written; averages of the uploaded nodes
must equal the CPU oracle's.  Exercises k_tile_commit's insert / replace / remove paths, which the NJ join loop (append
only) never takes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NOCODE = 127


def random_profile(rng, n_pos, n_codes, dt, p_vec, p_gap, p_oddw):
    """A profile in the reference's terms: code or NOCODE per column; a vector where NOCODE and weight > 0."""
    codes = rng.integers(0, n_codes, n_pos).astype(np.uint8)
    kind = rng.random(n_pos)
    w = np.ones(n_pos, dt)
    f = np.zeros((n_pos, n_codes), dt)
    vec = kind < p_vec
    gap = (kind >= p_vec) & (kind < p_vec + p_gap)
    codes[vec | gap] = NOCODE
    w[gap] = 0
    fv = rng.random((n_pos, n_codes)).astype(dt)
    fv /= fv.sum(1, keepdims=True)
    f[vec] = fv[vec]
    odd = rng.random(n_pos) < p_oddw          # explicit weights, on code and vector columns alike
    w[odd & ~gap] = rng.random(int((odd & ~gap).sum())).astype(dt) * 0.9 + 0.05
    return w, codes, f


@pytest.mark.parametrize("n_codes,dt", [(4, np.float32), (4, np.float64), (20, np.float32)])
def test_random_order_writes_read_back(n_codes, dt):
    from veryfasttree_amd import HipProfileOps, synth
    rng = np.random.default_rng(11 + n_codes)
    n_seqs, n_pos = 100, 77                     # nSeqs % 64 != 0: the first profile tile is shared with leaves
    codes = synth.random_descent_codes(n_seqs, n_pos, n_codes, 0.1, 0.05, seed=3)
    ops = HipProfileOps(n_seqs, n_pos, n_codes, dt)
    ops.upload_leaves(codes)
    ids = np.arange(n_seqs, 2 * n_seqs)
    ops.set_max_node(2 * n_seqs)
    last = {}
    order = np.concatenate([rng.permutation(ids), rng.permutation(ids)[:60], rng.permutation(ids)[:30]])
    for step, v in enumerate(order):
        # later rounds change the sparsity pattern: vectors and explicit weights appear and disappear
        p_vec = [0.3, 0.05, 0.6][min(step // len(ids), 2)]
        prof = random_profile(rng, n_pos, n_codes, dt, p_vec, 0.1, 0.2)
        ops.profile_upload(int(v), prof)
        last[int(v)] = prof
    for v in ids:
        w, c, f = ops.profile_download(int(v))
        ew, ec, ef = last[int(v)]
        vec = (ec == NOCODE) & (ew > 0)
        assert np.array_equal(c, ec), v
        assert np.array_equal(w, ew), v
        assert np.array_equal(f[vec], ef[vec]), v
        assert ops.profile_nvectors(int(v), 1)[0] == int(vec.sum())


def test_batched_average_into_scattered_tiles_matches_oracle():
    from veryfasttree_amd import HipProfileOps, synth
    from oracle import Oracle
    rng = np.random.default_rng(5)
    n_seqs, n_pos, n_codes, dt = 300, 90, 4, np.float32
    codes = synth.random_descent_codes(n_seqs, n_pos, n_codes, 0.2, 0.1, seed=9)
    ops = HipProfileOps(n_seqs, n_pos, n_codes, dt)
    ops.upload_leaves(codes)
    ops.set_max_node(2 * n_seqs)
    orc = Oracle(dt)
    # one launch writes nodes scattered over five tiles in shuffled order; a second rewrites a third of them
    out = rng.permutation(np.arange(n_seqs, 2 * n_seqs))[:200]
    a = rng.integers(0, n_seqs, len(out))
    b = rng.integers(0, n_seqs, len(out))
    ops.averageProfile(out, a, b)
    redo = out[::3]
    a2 = rng.integers(0, n_seqs, len(redo))
    ops.averageProfile(redo, a2, out[1::3][:len(redo)])   # children: leaves and already written internal nodes
    expect = {}
    leaf = [orc.leaf_profile(codes[i], n_codes) for i in range(n_seqs)]
    for k, v in enumerate(out):
        expect[int(v)] = orc.average_profile(leaf[a[k]], leaf[b[k]])
    snapshot = dict(expect)
    for k, v in enumerate(redo):
        expect[int(v)] = orc.average_profile(leaf[a2[k]], snapshot[int(out[1::3][k])])
    for v, (ew, ec, ef) in expect.items():
        w, c, f = ops.profile_download(v)
        vec = (ec == NOCODE) & (ew > 0)
        assert np.array_equal(c, ec) and np.array_equal(w, ew) and np.array_equal(f[vec], ef[vec]), v


def test_join_nodes_equals_the_separate_state_calls():
    """vft_join_nodes = set_max_node + 2 x set_parents + set_node_scalars(diameter) + set_out_distances for one join:
    the same sweep results (sentinels for the joined children, criterion inputs of the new node) either way."""
    from veryfasttree_amd import HipProfileOps, synth
    n_seqs, n_pos = 200, 64
    codes = synth.random_descent_codes(n_seqs, n_pos, 4, 0.1, 0.05, seed=21)
    results = []
    for fused in (False, True):
        ops = HipProfileOps(n_seqs, n_pos, 4, np.float32)
        ops.upload_leaves(codes)
        ops.set_node_scalars(0, np.zeros(n_seqs, np.float32), (codes != NOCODE).sum(1).astype(np.float32),
                             np.zeros(n_seqs, np.float32))
        ops.set_max_node(n_seqs)
        ops.outProfile(np.arange(n_seqs))
        ops.set_out_distances(0, np.zeros(n_seqs, np.float32), np.full(n_seqs, 10 * n_seqs, np.int64))
        ops.setOutDistance(None, n_seqs, 0.0)
        new, i, j, diam = n_seqs, 17, 101, np.float32(0.03125)
        if fused:
            ops.join_nodes(i, j, new, float(diam), 10 * n_seqs)
        else:
            ops.set_max_node(new + 1)
            ops.set_parents(i, [new])
            ops.set_parents(j, [new])
            ops.set_node_scalars(new, diameter=np.array([diam], np.float32))
            ops.set_out_distances(new, np.zeros(1, np.float32), np.array([10 * n_seqs], np.int64))
        ops.averageProfile([new], [i], [j])
        ops.updateOutProfile(i, j, new, n_seqs)
        hits, best = ops.setBestHit(new, n_seqs - 1, 1, float(diam), 40)
        d, w, c = ops.sweep_results(0, new + 1)
        od, na = ops.get_out_distances(0, new + 1)
        results.append((hits.copy(), best, d, w, c, od, na, ops.get_node_scalars(0, new + 1)))
    a, b = results
    assert a[1] == b[1] and np.array_equal(a[0], b[0])
    for x, y in zip(a[2:7], b[2:7]):
        assert np.array_equal(x, y)
    for x, y in zip(a[7], b[7]):
        assert np.array_equal(x, y)
    assert a[2][17] == np.float32(1e20) and a[2][101] == np.float32(1e20)      # joined children: sentinel


@pytest.mark.parametrize("n_pos,gap", [(2600, 0.02), (6000, 0.02), (5, 0.3), (16, 0.4), (17, 0.6)])
def test_long_short_and_gappy_alignments_sweep_equals_pair_list_and_oracle(n_pos, gap):
    """Alignments longer than 2560 columns: the wave-per-item kernels run with fewer waves per workgroup (their LDS
    staging grows with the length).  Very short and very gappy ones: rows that are almost or entirely gaps, columns
    that are all gaps, fewer columns than one 16-column chunk.  The sweep (lane per target, no LDS) and the pair list (wave / workgroup per pair)
    are independent implementations and must agree bit for bit; a few pairs are checked against the CPU oracle."""
    from veryfasttree_amd import HipProfileOps, synth
    from veryfasttree_amd.workload import TopHitsState
    from oracle import Oracle
    n = 192
    codes = synth.random_descent_codes(n, n_pos, 4, 0.05, gap, seed=31)
    if gap > 0.1:
        codes[5] = NOCODE          # an empty sequence
        codes[:, 0] = NOCODE       # an empty column
    ops = HipProfileOps(n, n_pos, 4, np.float32)
    st = TopHitsState(ops, codes, 48)
    q = int(st.active[st.active >= n][5])
    ops.setBestHit(q, st.n_active, st.n_diff_allow, st.totdiam, 16)
    d, w, c = ops.sweep_results(0, st.maxnode)
    others = st.active[st.active != q]
    pd, pw, pc = ops.setDistCriterion(np.full(len(others), q), others, st.n_active, st.n_diff_allow, st.totdiam)
    assert np.array_equal(pd, d[others]) and np.array_equal(pw, w[others]) and np.array_equal(pc, c[others])
    rd, rw = ops.profileDist(np.full(4, q), others[[0, 7, 100, 120]])
    orc = Oracle(np.float32)
    pq = ops.profile_download(q)
    for k, j in enumerate(others[[0, 7, 100, 120]]):
        ed, ew = orc.profiledist(pq, ops.profile_download(int(j)))
        assert rd[k] == ed and rw[k] == ew


@pytest.mark.parametrize("n_codes,dt", [(4, np.float32), (4, np.float64), (20, np.float32)])
def test_sweep_batch_equals_single_sweeps(n_codes, dt):
    """vft_sweep_batch = the same sweeps seed after seed, one batched top-k selection: identical records and bestjoin,
    for leaf and internal seeds mixed, with stale out-distances in play."""
    from veryfasttree_amd import HipProfileOps, synth
    from veryfasttree_amd.workload import TopHitsState
    n, L = 3000, 120
    codes = synth.random_descent_codes(n, L, n_codes, 0.05, 0.03, seed=41)

    def fresh():
        ops = HipProfileOps(n, L, n_codes, dt)
        if n_codes == 20:
            dm = G_load("wb_aa_f32")
            ops.set_distance_matrix(dm["dmat.distances"], dm["dmat.codefreq"], dm["dmat.eigenval"], dm["dmat.eigentot"])
        st = TopHitsState(ops, codes, 700)
        # make a third of the nodes stale beyond the allowance, so that the lazy refresh has work to do
        od, na = ops.get_out_distances(0, st.maxnode)
        na[::3] = st.n_active + st.n_diff_allow + 50
        ops.set_out_distances(0, od, na)
        return ops, st

    ops1, st = fresh()
    rng = np.random.default_rng(2)
    seeds = np.concatenate([rng.choice(st.active[st.active < n], 5, replace=False),
                            rng.choice(st.active[st.active >= n], 5, replace=False)])
    rng.shuffle(seeds)
    k = 64
    single = [ops1.setBestHit(int(q), st.n_active, st.n_diff_allow, st.totdiam, k) for q in seeds]
    ops2, _ = fresh()
    hits, best = ops2.setBestHitBatch(seeds, st.n_active, st.n_diff_allow, st.totdiam, k)
    for s, (h1, b1) in enumerate(single):
        assert b1 == best[s]
        assert np.array_equal(h1, hits[s])
    # and the refreshed out-distances ended up the same
    a, b = ops1.get_out_distances(0, st.maxnode), ops2.get_out_distances(0, st.maxnode)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # sharded: every "rank" sweeps its target range for the whole batch, the [ranks][seeds][k] block is merged in one
    # launch (what bench.py does after its one all-gather per step) and equals the unsharded lists
    from veryfasttree_amd.workload import shard_range
    parts = []
    for r in range(3):
        lo, hi = shard_range(st.maxnode, r, 3)
        ops2.set_shard(lo, hi)
        parts.append(ops2.setBestHitBatch(seeds, st.n_active, st.n_diff_allow, st.totdiam, k)[0])
    ops2.set_shard(0, st.maxnode)
    d_all = ops2.device_buffer(np.stack(parts))
    merged = ops2.merge_hits_batch(d_all, 3, len(seeds), k)
    ops2.device_free(d_all)
    assert np.array_equal(merged, hits)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_seeds_sharing_a_pass_over_the_targets_equal_single_sweeps(dt):
    """vft_sweep_batch takes its leaf seeds four (two) per pass over the targets (k_sweep_nt_leafq_multi) and its profile seeds likewise
    (k_sweep_nt_profq_multi): every record equals the one-launch-per-seed sweeps' (VFT_DEBUG_NO_MULTI_SWEEP) and the single vft_sweep
    calls', leaf and internal seeds mixed, a gappy alignment, every criterion of the first seed of a group (leaf, then profile), and a
    target range that starts inside the leaves."""
    import ctypes
    from veryfasttree_amd import HipProfileOps, synth
    from veryfasttree_amd.workload import TopHitsState
    n, L = 5000, 173
    codes = synth.random_descent_codes(n, L, 4, 0.05, 0.08, seed=17)

    def fresh(no_multi):
        ops = HipProfileOps(n, L, 4, dt)
        assert ops.lib.vft_debug_option(ops.ctx, ctypes.c_int32(12), ctypes.c_int64(1 if no_multi else 0)) == 0
        return ops, TopHitsState(ops, codes, 1800)

    ops1, st = fresh(True)
    ops2, _ = fresh(False)
    rng = np.random.default_rng(5)
    leaves = rng.choice(st.active[st.active < n], 18, replace=False)
    inner = rng.choice(st.active[st.active >= n], 7, replace=False)
    # 18 leaf seeds (4 + 4 + 4 + 4 + 2 per pass) with 7 profile seeds (4 + 2 per launch of k_sweep_nt_profq_multi, the last one alone) in between
    seeds = np.concatenate([leaves[:6], inner[:1], leaves[6:7], inner[1:2], leaves[7:11], inner[2:3], leaves[11:13], inner[3:7], leaves[13:18]])
    k = 300
    single = [ops1.setBestHit(int(q), st.n_active, st.n_diff_allow, st.totdiam, k) for q in seeds]
    h1, b1 = ops1.setBestHitBatch(seeds, st.n_active, st.n_diff_allow, st.totdiam, k)
    h2, b2 = ops2.setBestHitBatch(seeds, st.n_active, st.n_diff_allow, st.totdiam, k)
    assert np.array_equal(b1, b2) and np.array_equal(h1, h2)
    for s, (h, b) in enumerate(single):
        assert b == b2[s] and np.array_equal(h, h2[s])
    # all criteria of a group's first seed (slot 0 of the batch) against the single sweep's
    ops1.setBestHit(int(seeds[0]), st.n_active, st.n_diff_allow, st.totdiam, k)
    ops2.setBestHitBatch(seeds[:4], st.n_active, st.n_diff_allow, st.totdiam, k)
    r1, r2 = ops1.sweep_results(0, st.maxnode), ops2.sweep_results(0, st.maxnode)
    for a, b in zip(r1, r2):
        assert np.array_equal(a, b)
    ops1.setBestHit(int(inner[0]), st.n_active, st.n_diff_allow, st.totdiam, k)
    ops2.setBestHitBatch(inner[:4], st.n_active, st.n_diff_allow, st.totdiam, k)
    r1, r2 = ops1.sweep_results(0, st.maxnode), ops2.sweep_results(0, st.maxnode)
    for a, b in zip(r1, r2):
        assert np.array_equal(a, b)
    lo, hi = 1984, st.maxnode - 77
    lo -= lo % 64
    ops1.set_shard(lo, hi)
    ops2.set_shard(lo, hi)
    p1 = ops1.setBestHitBatch(seeds, st.n_active, st.n_diff_allow, st.totdiam, k)
    p2 = ops2.setBestHitBatch(seeds, st.n_active, st.n_diff_allow, st.totdiam, k)
    assert np.array_equal(p1[0], p2[0]) and np.array_equal(p1[1], p2[1])


def G_load(name):
    import golden_util as G
    return G.load(name)


def test_leaf_block_distances_equal_the_pair_list():
    """vft_leaf_block_distances (k_leaf_block: the close-neighbour transfers of setAllLeafTopHits as one cross-product
    call) against vft_pair_distances on the same pairs, bit for bit - f32 and f64, list lengths that do not fill the
    kernel's 64 x 64 tiles, negative ids in the candidate list, a stale out-distance in play."""
    from veryfasttree_amd import HipProfileOps, synth
    n, L = 3000, 333
    codes = synth.random_descent_codes(n, L, 4, 0.05, 0.08, seed=91)
    for dt in (np.float32, np.float64):
        ops = HipProfileOps(n, L, 4, dt)
        ops.upload_leaves(codes)
        ops.set_node_scalars(0, np.zeros(n, dt), (codes != 127).sum(1).astype(dt), np.zeros(n, dt))
        ops.outProfile(np.arange(n))
        stamps = np.full(n, n, np.int64)
        stamps[[5, 77, 1500]] = n + 200          # staler than allowed: refreshed inside the call
        rng = np.random.default_rng(4)
        ops.set_out_distances(0, rng.uniform(0, 50, n).astype(dt), stamps)
        a = rng.choice(n, 83, replace=False)
        b = rng.choice(n, 150, replace=False).astype(np.int64)
        b[[3, 40]] = -1
        b[10] = a[7]                                # a pair of a leaf with itself
        d, w, c = ops.leafBlockDistances(a, b, n, 30, 12.5)
        ok = b >= 0
        pi, pj = np.repeat(a, ok.sum()), np.tile(b[ok], len(a))
        d2, w2, c2 = ops.setDistCriterion(pi, pj, n, 30, 12.5)
        assert np.array_equal(d[:, ok].ravel(), d2) and np.array_equal(w[:, ok].ravel(), w2) and np.array_equal(c[:, ok].ravel(), c2)
        assert np.all(d[:, ~ok] == dt(1e20)) and np.all(w[:, ~ok] == 0)
        ops.close()


@pytest.mark.parametrize("ncodes", [4, 20])
def test_fused_join_and_block_distances_equal_the_separate_calls(ncodes):
    """vft_join_fused (tree arrays + average + self-distance + out-profile update in one launch, tile streams rebuilt
    lazily) against vft_join_nodes + vft_average_profiles + vft_out_profile_update on a second context: profiles, node
    scalars, out-profile and a sweep over the joined tree, bit for bit.  Then vft_block_distances over a mix of leaves and
    internal nodes (negative ids, a pair of a node with itself, stale out-distances) against vft_pair_distances."""
    from veryfasttree_amd import HipProfileOps, synth
    n, L, nj = 300, 211, 150
    codes = synth.random_descent_codes(n, L, ncodes, 0.05, 0.08, seed=17)
    rng = np.random.default_rng(3)
    for dt in (np.float32, np.float64):
        both = []
        for fused in (False, True):
            ops = HipProfileOps(n, L, ncodes, dt, max_nodes=2 * n)
            ops.upload_leaves(codes)
            ops.set_node_scalars(0, np.zeros(n, dt), (codes != 127).sum(1).astype(dt), np.zeros(n, dt))
            ops.outProfile(np.arange(n))
            active = list(range(n))
            order = np.random.default_rng(5)
            for k in range(nj):
                i, j = (int(x) for x in order.choice(len(active), 2, replace=False))
                i, j = active[i], active[j]
                new, diam = n + k, 0.01 * (k % 7)
                if fused:
                    ops.join_fused(i, j, new, diam, 10 * n, len(active), True)
                else:
                    ops.join_nodes(i, j, new, diam, 10 * n)
                    ops.averageProfile([new], [i], [j])
                    ops.updateOutProfile(i, j, new, len(active))
                active = [v for v in active if v not in (i, j)] + [new]
            both.append((ops, active))
        (o1, act), (o2, _) = both
        for v in range(n, n + nj):
            p1, p2 = o1.profile_download(v), o2.profile_download(v)
            assert all(np.array_equal(x, y) for x, y in zip(p1, p2)), v
        s1, s2 = o1.get_node_scalars(n, nj), o2.get_node_scalars(n, nj)
        assert all(np.array_equal(x, y) for x, y in zip(s1, s2))
        assert all(np.array_equal(x, y) for x, y in zip(o1.out_profile_download()[0], o2.out_profile_download()[0]))
        nact = len(act)
        h1 = o1.setBestHit(act[-1], nact, 5, 3.0, 40)
        h2 = o2.setBestHit(act[-1], nact, 5, 3.0, 40)     # flushes the pending tile rebuild first
        assert np.array_equal(h1[0], h2[0]) and h1[1] == h2[1]
        a = np.array(rng.choice(act, 37, replace=False), np.int64)
        b = np.array(rng.choice(act, 90, replace=False), np.int64)
        b[[2, 11]] = -1
        b[5] = a[3]
        d = o2.blockDistances(a, b, nact, 5, 3.0)
        ok = np.ones((len(a), len(b)), bool)
        ok[:, b < 0] = False
        ok &= a[:, None] != b[None, :]          # (b[5] == a[3] for sure; the random draws may share more nodes)
        assert not ok[3, 5]
        pi, pj = np.repeat(a, len(b)).reshape(len(a), -1)[ok], np.tile(b, len(a)).reshape(len(a), -1)[ok]
        d2, w2, c2 = o1.setDistCriterion(pi, pj, nact, 5, 3.0)
        assert np.array_equal(d[ok], d2)
        # both calls refreshed the same stale out-distances
        assert all(np.array_equal(x, y) for x, y in zip(o1.get_out_distances(0, n + nj), o2.get_out_distances(0, n + nj)))
        o1.close()
        o2.close()


@pytest.mark.parametrize("ncodes,matrix", [(4, False), (20, False), (20, True)])
def test_out_profile_chain_kernel_equals_the_column_walk(ncodes, matrix):
    """vft_out_profile_full over all active nodes (k_leaf_hist + vft_iterate_add for matrix-free nucleotide leaves,
    k_outprofile_chain with LDS-staged addends for the rest) against the one-thread-per-column walk over the same list
    (k_outprofile_full, forced through vft_debug_option), bit for bit: 5000 leaves with gaps, 3000 joins,
    f32 and f64, with and without a distance matrix."""
    import os
    from veryfasttree_amd import HipProfileOps, synth
    n, L, nj = 5000, 157, 3000
    codes = synth.random_descent_codes(n, L, ncodes, 0.05, 0.15, seed=23)
    for dt in (np.float32, np.float64):
        ops = HipProfileOps(n, L, ncodes, dt, max_nodes=2 * n)
        if matrix:
            from veryfasttree_amd.backend import distance_tables
            t = distance_tables(None, dt)
            ops.set_distance_matrix(t["distances"], t["codefreq"], t["eigenval"], t["eigentot"])
        ops.upload_leaves(codes)
        ops.set_node_scalars(0, np.zeros(n, dt), (codes != 127).sum(1).astype(dt), np.zeros(n, dt))
        ops.outProfile(np.arange(n))
        first = ops.out_profile_download(matrix)
        ops.debug_option(4, 1)
        try:
            ops.outProfile(np.arange(n))
            ref = ops.out_profile_download(matrix)
        finally:
            ops.debug_option(4, 0)
        assert all(np.array_equal(x, y) for x, y in zip(first[0], ref[0]))
        assert first[1] is None or np.array_equal(first[1], ref[1])
        active = list(range(n))
        order = np.random.default_rng(8)
        for k in range(nj):
            i, j = (int(x) for x in order.choice(len(active), 2, replace=False))
            i, j = active[i], active[j]
            ops.join_fused(i, j, n + k, 0.0, 10 * n, len(active), True)
            active = [v for v in active if v != i and v != j] + [n + k]
            if k in (40, 700, nj - 1):
                ids = np.array(sorted(active))
                ops.outProfile(ids)
                got = ops.out_profile_download(matrix)
                ops.debug_option(4, 1)
                try:
                    ops.outProfile(ids)
                    ref = ops.out_profile_download(matrix)
                finally:
                    ops.debug_option(4, 0)
                assert all(np.array_equal(x, y) for x, y in zip(got[0], ref[0])), (k, dt)
                assert got[1] is None or np.array_equal(got[1], ref[1])
        ops.close()


@pytest.mark.parametrize("fused", [True, False])
def test_pair_list_with_refreshes_in_one_call(fused, monkeypatch):
    """vft_pair_distances_refresh (forced + lazy out-distance refreshes travelling with the pair list; with `fused` the
    single-launch kernel whose pair workgroups wait for the refresh workgroups, otherwise two launches) against the
    separate calls on a second context: vft_out_distances(forced ids), then vft_pair_distances - distances, criteria and
    every out-distance / stamp afterwards, bit for bit; leaves and internal nodes, nodes named many times, a forced node
    that is also a pair end, a forced node whose stamp is already current."""
    from veryfasttree_amd import HipProfileOps, synth
    n, L, nj = 400, 190, 200
    codes = synth.random_descent_codes(n, L, 4, 0.05, 0.05, seed=31)
    for dt in (np.float32, np.float64):
        ctxs = []
        for _ in range(2):
            rng = np.random.default_rng(12)       # the same draws for both contexts
            ops = HipProfileOps(n, L, 4, dt, max_nodes=2 * n)
            if not fused:
                ops.debug_option(1, 1)   # VFT_DEBUG_NO_FUSED_REFRESH
            ops.upload_leaves(codes)
            ops.set_node_scalars(0, np.zeros(n, dt), (codes != 127).sum(1).astype(dt), np.zeros(n, dt))
            ops.outProfile(np.arange(n))
            active = list(range(n))
            order = np.random.default_rng(6)
            for k in range(nj):
                i, j = (int(x) for x in order.choice(len(active), 2, replace=False))
                i, j = active[i], active[j]
                ops.join_fused(i, j, n + k, 0.02 * (k % 5), 10 * n, len(active), True)
                active = [v for v in active if v != i and v != j] + [n + k]
            nact = len(active)
            stamps = np.full(n + nj, nact, np.int64)
            stale = rng.choice(active, 60, replace=False)
            stamps[stale] = nact + 50             # staler than allowed
            stamps[active[3]] = nact + 2          # fresh enough for the lazy rule, not for a forced refresh
            ops.set_out_distances(0, np.random.default_rng(2).uniform(0, 30, n + nj).astype(dt), stamps)
            ctxs.append((ops, active, nact, stale))
        (o1, active, nact, stale), (o2, _, _, _) = ctxs
        rng = np.random.default_rng(13)
        hub = active[-1]                           # the newest node against many partners, as after a join
        partners = np.array([v for v in rng.choice(active, 150, replace=False) if v != hub], np.int64)
        pi = np.concatenate([np.full(len(partners), hub, np.int64), rng.choice(active, 40)])
        pj = np.concatenate([partners, rng.choice(active, 40)])
        keep = pi != pj
        pi, pj = pi[keep], pj[keep]
        forced = np.array([hub, active[3], active[5], int(stale[0])], np.int64)   # active[5]: stamp already current
        d1, w1, c1 = o1.setDistCriterionRefresh(pi, pj, forced, nact, 5, 2.5)
        o2.setOutDistance(forced, nact, 2.5)
        d2, w2, c2 = o2.setDistCriterion(pi, pj, nact, 5, 2.5)
        assert np.array_equal(d1, d2) and np.array_equal(w1, w2) and np.array_equal(c1, c2)
        assert all(np.array_equal(x, y) for x, y in zip(o1.get_out_distances(0, n + nj), o2.get_out_distances(0, n + nj)))
        # one pair and many forced nodes that are no end of it: nobody waits for those refresh workgroups but the publishing
        # one; the host-mapped mirrors must hold every forced refresh the moment the call returns (no synchronisation here)
        for rep in range(20):
            far = np.array([v for v in active if v not in (active[0], active[1])][rep:rep + 120], np.int64)
            o1.set_out_distances(0, np.zeros(n + nj, dt), np.full(n + nj, nact + 7, np.int64))
            o1.setDistCriterionRefresh(np.array([active[0]], np.int64), np.array([active[1]], np.int64), far, nact, 50, 2.5)
            mo, mn = o1.out_distance_mirror()
            assert np.all(mn[far] == nact), rep
            o2.set_out_distances(0, np.zeros(n + nj, dt), np.full(n + nj, nact + 7, np.int64))
            o2.setOutDistance(far, nact, 2.5)
            assert np.array_equal(mo[far], o2.get_out_distances(0, n + nj)[0][far]), rep
        o1.close()
        o2.close()


@pytest.mark.gpu
def test_a_selection_that_overflows_does_not_look_at_what_it_never_stored():
    """More candidates below the threshold digit than the candidate buffer takes (16 000 sequences one substitution away from a common
    ancestor among 24 000 leaves of 800 columns: their criteria crowd into a sliver of the key range): the collection overflows, the host narrows the key range and repeats.  The
    overflowed round counted candidates it never stored; ranking them read ids from whatever the buffer held (round 5: a GPU memory fault
    in the second tree of a process, when the allocator handed out recycled blocks).  With the buffers poisoned the selection must still
    return the k smallest records in the reference's order, one launch per seed and four seeds per pass alike."""
    from veryfasttree_amd import HipProfileOps, synth
    from veryfasttree_amd.workload import TopHitsState
    n, L, n_copies = 24000, 800, 16000
    rng = np.random.default_rng(12)
    codes = synth.random_descent_codes(n, L, 4, 0.3, 0.0, seed=3)
    base = codes[0].copy()
    for i in range(n_copies):   # near-copies of one sequence: one substitution each
        row = base.copy()
        pos = rng.choice(L, 1, replace=False)
        row[pos] = (row[pos] + rng.integers(1, 4, len(pos))) % 4
        codes[i] = row
    ops = HipProfileOps(n, L, 4, np.float32)
    st = TopHitsState(ops, codes, 64)
    k = 2000
    seeds = st.active[st.active < n_copies][:4]
    ops.setBestHitBatch(seeds, st.n_active, st.n_diff_allow, st.totdiam, k)   # (the slots exist now)
    ops.debug_option(14, 1)
    hits, best = ops.setBestHitBatch(seeds, st.n_active, st.n_diff_allow, st.totdiam, k)
    ops.debug_option(14, 1)
    h0, b0 = ops.setBestHit(int(seeds[0]), st.n_active, st.n_diff_allow, st.totdiam, k)
    assert ops.sweep_info()[1] > 0, "the key range was not narrowed: this alignment no longer overflows the candidate buffer"
    dist, weight, crit = ops.sweep_results(0, st.maxnode)
    ids = np.nonzero(st.parent < 0)[0]
    order = ids[np.lexsort((-ids, crit[ids]))][:k]
    assert np.array_equal(h0["j"], order) and np.array_equal(h0["criterion"], crit[order])
    assert b0 == best[0] and np.array_equal(h0, hits[0])

