"""The entry points round 4 added for batched refinement steps, each against the calls it replaces - bit for bit:
vft_average_chains / vft_posterior_chains_blen (several independent chains in one launch) against one vft_*_chain call per chain,
vft_walk_step (queued averages + the six distances of a quartet, through the walk server) against vft_average_chain + vft_profile_distances,
vft_profiles_differ against a host comparison of the downloaded profiles."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

I64, I32, P = C.c_int64, C.c_int32, C.c_void_p


def ptr(a):
    return a.ctypes.data_as(P)


def make_state(dt, n_codes, n=48, L=137, seed=5):
    """48 leaves, 47 internal profiles averaged from random earlier nodes, everything as plain rows; ids 95.. are free"""
    from veryfasttree_amd import HipProfileOps, synth
    rng = np.random.default_rng(seed)
    codes = synth.random_descent_codes(n, L, n_codes, 0.15, 0.05, seed=seed)
    ops = HipProfileOps(n, L, n_codes, dt, max_nodes=8 * n)
    if n_codes == 20:   # the BLOSUM45-derived tables of the reference's protein NJ / ME phase
        from veryfasttree_amd.backend import distance_tables
        t = distance_tables(None, dt)
        ops.set_distance_matrix(t["distances"], t["codefreq"], t["eigenval"], t["eigentot"])
    ops.upload_leaves(codes)
    for v in range(n, 2 * n - 1):
        a, b = rng.choice(v, 2, replace=False)
        ops.averageProfile([v], [int(a)], [int(b)])
    ops.set_max_node(8 * n)
    assert ops.lib.vft_set_profile_rows(ops.ctx, I32(1)) == 0
    return ops, rng, 2 * n - 1


def same(p, q):
    (w1, c1, f1), (w2, c2, f2) = p, q
    vec = (w1 > 0) & (c1 == 127)
    return np.array_equal(w1.view(np.uint8), w2.view(np.uint8)) and np.array_equal(c1, c2) and np.array_equal(f1[vec].view(np.uint8), f2[vec].view(np.uint8))


@pytest.mark.parametrize("dt,nc", [(np.float32, 4), (np.float64, 4), (np.float64, 20)])
def test_average_chains_equal_single_chains(dt, nc):
    ops, rng, free = make_state(dt, nc)
    # three chains with inner dependencies, written to two disjoint sets of ids
    chains = []
    for ch in range(3):
        ops_list = []
        prev = None
        for k in range([3, 1, 6][ch]):
            a = int(rng.integers(0, free)) if prev is None or k % 2 else prev
            b = int(rng.integers(0, free))
            ops_list.append((a, b))
            prev = -1 - len(ops_list)   # placeholder: "the op before"
        chains.append(ops_list)

    def ids(base):
        out, a, b, off = [], [], [], [0]
        nxt = base
        for ops_list in chains:
            first = nxt
            for k, (x, y) in enumerate(ops_list):
                out.append(nxt)
                a.append(first + (-1 - x) - 1 if x < 0 else x)
                b.append(y)
                nxt += 1
            off.append(len(out))
        return np.array(out, np.int64), np.array(a, np.int64), np.array(b, np.int64), np.array(off, np.int32)

    o1, a1, b1, off = ids(free)
    assert ops.lib.vft_average_chains(ops.ctx, I32(len(off) - 1), ptr(off), ptr(o1), ptr(a1), ptr(b1)) == 0
    o2, a2, b2, _ = ids(free + 40)
    for ch in range(len(off) - 1):
        s, e = int(off[ch]), int(off[ch + 1])
        assert ops.lib.vft_average_chain(ops.ctx, I32(e - s), ptr(o2[s:e].copy()), ptr(a2[s:e].copy()), ptr(b2[s:e].copy())) == 0
    for x, y in zip(o1, o2):
        assert same(ops.profile_download(int(x)), ops.profile_download(int(y))), (x, y)
    differ = np.zeros(len(o1), np.int32)
    assert ops.lib.vft_profiles_differ(ops.ctx, I64(len(o1)), ptr(o1), ptr(o2), ptr(differ)) == 0
    assert not differ.any()
    other = np.roll(o2, 1)
    assert ops.lib.vft_profiles_differ(ops.ctx, I64(len(o1)), ptr(o1), ptr(other), ptr(differ)) == 0
    want = np.array([0 if same(ops.profile_download(int(x)), ops.profile_download(int(y))) else 1 for x, y in zip(o1, other)], np.int32)
    assert np.array_equal(differ, want) and want.any()
    ops.close()


@pytest.mark.parametrize("dt,nc", [(np.float32, 4), (np.float64, 4), (np.float32, 20), (np.float64, 20)])
def test_walk_step_equals_chain_plus_distances(dt, nc):
    """vft_walk_step hands the step to the walk server (tests/test_gpu_walk_server.py has the long runs); without a server it answers
    VFT_ERR_STATE and the caller makes the two calls"""
    ops, rng, free = make_state(dt, nc)
    o0 = np.array([free], np.int64)
    assert ops.lib.vft_walk_step(ops.ctx, I32(1), ptr(o0), ptr(o0), ptr(o0), ptr(np.zeros(4, np.int64)), ptr(np.zeros(6, dt))) == 3
    assert ops.lib.vft_walk_server_start(ops.ctx) == 0
    for trial in range(7):
        n = [0, 1, 2, 5, 9, 3, 19][trial]
        base1, base2 = free + 20 * trial, free + 140 + 20 * trial
        a = rng.integers(0, free, n).astype(np.int64)
        b = rng.integers(0, free, n).astype(np.int64)

        def build(base):
            aa, bb = a.copy(), b.copy()
            for k in range(1, n):   # every second op reads the one before it, one reads an older output
                if k % 2:
                    aa[k] = base + k - 1
                elif k >= 4:
                    bb[k] = base + k - 3
            return np.arange(base, base + n, dtype=np.int64), aa, bb

        o1, a1, b1 = build(base1)
        o2, a2, b2 = build(base2)
        # the quartet: a leaf, an internal node, and (when there are any) two of the chain's outputs
        q = np.array([int(rng.integers(0, 48)), int(rng.integers(48, free)), int(rng.integers(0, free)), int(rng.integers(0, free))], np.int64)
        q1, q2 = q.copy(), q.copy()
        if n >= 2:
            q1[2], q1[3], q2[2], q2[3] = o1[n - 1], o1[0], o2[n - 1], o2[0]
        if trial == 5:
            q1[1] = q2[1] = int(rng.integers(0, 48))   # a leaf x leaf pair as well
        d1 = np.zeros(6, dt)
        assert ops.lib.vft_walk_server_start(ops.ctx) == 0   # (the plain calls below retire it every time)
        assert ops.lib.vft_walk_step(ops.ctx, I32(n), ptr(o1), ptr(a1), ptr(b1), ptr(q1), ptr(d1)) == 0
        if n:
            assert ops.lib.vft_average_chain(ops.ctx, I32(n), ptr(o2), ptr(a2), ptr(b2)) == 0
        pi = np.array([q2[0], q2[0], q2[0], q2[1], q2[1], q2[2]], np.int64)
        pj = np.array([q2[1], q2[2], q2[3], q2[2], q2[3], q2[3]], np.int64)
        d2, w2 = ops.profileDist(pi, pj)
        assert np.array_equal(np.asarray(d1).view(np.uint8), np.asarray(d2, dt).view(np.uint8)), (trial, d1, d2)
        for x, y in zip(o1, o2):
            assert same(ops.profile_download(int(x)), ops.profile_download(int(y)))
    ops.close()


@pytest.mark.parametrize("dt,nc", [(np.float32, 4), (np.float64, 20)])
def test_walk_step_that_rewrites_nodes(dt, nc):
    """Three steps in ten of an SPR round write a node that the step read or wrote earlier (an up-profile slot re-used, a node
    recomputed after its old profile went into another average).  Round 4's step kernel ran the chain in each of its six workgroups
    and raced on such steps; in the walk server every column has one owner.  Against the plain sequence on an identical state, twenty
    times over (the race needed a dozen runs of a pipeline to show)."""
    a_ops, rng, free = make_state(dt, nc, seed=11)
    b_ops, _, _ = make_state(dt, nc, seed=11)
    assert a_ops.lib.vft_walk_server_start(a_ops.ctx) == 0
    X, Y, Z = free, free + 1, free + 2
    for trial in range(20):
        e = [int(v) for v in rng.integers(0, free, 6)]
        # X written, read, written again; e[2] (an old node) read and then overwritten; Y written twice without a read in between
        out = np.array([X, Y, X, Y, Z, e[2] if e[2] >= 48 else 60, Z], np.int64)
        a = np.array([e[0], X, Y, e[3], e[2], X, Z], np.int64)
        b = np.array([e[1], e[2], e[4], X, Y, e[5], e[0]], np.int64)
        q = np.array([X, Z, int(out[5]), e[1]], np.int64)
        d1 = np.zeros(6, dt)
        assert a_ops.lib.vft_walk_step(a_ops.ctx, I32(len(out)), ptr(out), ptr(a), ptr(b), ptr(q), ptr(d1)) == 0
        assert b_ops.lib.vft_average_chain(b_ops.ctx, I32(len(out)), ptr(out), ptr(a), ptr(b)) == 0
        pi = np.array([q[0], q[0], q[0], q[1], q[1], q[2]], np.int64)
        pj = np.array([q[1], q[2], q[3], q[2], q[3], q[3]], np.int64)
        d2, _ = b_ops.profileDist(pi, pj)
        assert np.array_equal(np.asarray(d1).view(np.uint8), np.asarray(d2, dt).view(np.uint8)), (trial, d1, d2)
    assert a_ops.lib.vft_walk_server_stop(a_ops.ctx) == 0
    for x in (X, Y, Z):
        assert same(a_ops.profile_download(x), b_ops.profile_download(x)), x
    a_ops.close()
    b_ops.close()


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_posterior_chains_equal_single_chains(dt):
    ops, rng, free = make_state(dt, 4)
    n_nodes = 8 * 48
    ops.set_rates(np.array([0.5, 1.0, 2.0], dt), rng.integers(0, 3, 137))
    ops.set_ml_limits(5e-4 if dt == np.float32 else 5e-9, 2.5e-4 if dt == np.float32 else 2.5e-9, 1e-10 if dt == np.float32 else 1e-20)
    ops.set_transition_matrix()
    ops.branch_lengths_set(0, rng.uniform(0.001, 0.4, n_nodes).astype(dt))
    off = np.array([0, 2, 2, 6], np.int32)   # (an empty chain in the middle)

    def ids(base):
        out = np.arange(base, base + 6, dtype=np.int64)
        a = np.array([3, base, 50, base + 2, 7, base + 4], np.int64)
        b = np.array([60, 9, 11, 70, base + 3, 52], np.int64)
        return out, a, b, rng.integers(0, 95, 6).astype(np.int64), rng.integers(0, 95, 6).astype(np.int64)

    o1, a1, b1, la, lb = ids(free)
    assert ops.lib.vft_posterior_chains_blen(ops.ctx, I32(3), ptr(off), ptr(o1), ptr(a1), ptr(b1), ptr(la), ptr(lb)) == 0
    o2, a2, b2, _, _ = ids(free + 40)
    for ch in range(3):
        s, e = int(off[ch]), int(off[ch + 1])
        if e > s:
            assert ops.lib.vft_posterior_chain_blen(ops.ctx, I32(e - s), ptr(o2[s:e].copy()), ptr(a2[s:e].copy()), ptr(b2[s:e].copy()), ptr(la[s:e].copy()),
                                                    ptr(lb[s:e].copy())) == 0
    for x, y in zip(o1, o2):
        assert same(ops.profile_download(int(x)), ops.profile_download(int(y))), (x, y)
    ops.close()
