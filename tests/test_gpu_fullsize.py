"""Full-size GPU checks at BASELINE.json's configurations through size-independent properties, plus spot checks
against the oracle on sampled pairs (the oracle cannot sweep 10^5..10^6 targets in seconds)."""
import numpy as np
import pytest

from conftest import free_port, heavy
from oracle import Oracle

pytestmark = pytest.mark.gpu


def _state(n, L, seed, mu=0.03, gap=0.01, nc=4, dt=np.float32):
    from veryfasttree_amd import HipProfileOps, synth
    from veryfasttree_amd.backend import distance_tables
    from veryfasttree_amd.workload import TopHitsState
    codes = synth.random_descent_codes(n, L, nc, mu, gap, seed=seed)
    ops = HipProfileOps(n, L, nc, dt)
    tables = None
    if nc == 20:   # proteins: the BLOSUM45-derived distance matrix (host/AAModels.h), as the driver installs it
        tables = {k: v.astype(dt) for k, v in distance_tables(None, dt).items()}
        ops.set_distance_matrix(tables["distances"], tables["codefreq"], tables["eigenval"], tables["eigentot"])
    return codes, ops, TopHitsState(ops, codes, n // 4), tables


# BASELINE configs C3 and C4 (nucleotides, float32) and C5 (50k proteins x 300, double precision, distance matrix)
@pytest.mark.parametrize("n,L,seed,nc,dt,mu,gap", [(100000, 500, 3, 4, np.float32, 0.03, 0.01), (1000000, 200, 4, 4, np.float32, 0.02, 0.01),
                                                   (50000, 300, 5, 20, np.float64, 0.08, 0.02)])
def test_sweep_properties_at_full_size(n, L, seed, nc, dt, mu, gap):
    from veryfasttree_amd.workload import merge_hits, shard_range
    codes, ops, st, tables = _state(n, L, seed, mu, gap, nc, dt)
    orc = Oracle(dt)
    dm = orc.dmat(tables["distances"], tables["codefreq"], tables["eigenval"], tables["eigentot"]) if tables else None
    big = dt(1e20)
    m = int(0.5 + np.sqrt(n))
    k = 2 * m
    rng = np.random.default_rng(seed)
    leaf_q = int(st.active[st.active < n][rng.integers(0, 1000)])
    int_q = int(st.active[st.active >= n][rng.integers(0, 1000)])
    for q in (leaf_q, int_q):
        hits, best = ops.setBestHit(q, st.n_active, st.n_diff_allow, st.totdiam, k)
        dist, weight, crit = ops.sweep_results(0, st.maxnode)
        # 1. inactive targets carry the sentinel, active ones do not
        inactive = st.parent < 0
        assert np.all(crit[~inactive] == big) and np.all(crit[inactive] < big)
        # 2. the k hits are exactly the k smallest under (criterion asc, id desc), in that order
        ids = np.nonzero(inactive)[0]
        order = ids[np.lexsort((-ids, crit[ids]))][:k]
        assert np.array_equal(hits["j"], order)
        assert np.array_equal(hits["criterion"], crit[order]) and np.array_equal(hits["dist"], dist[order])
        # 3. bestjoin = smallest id among the minimal criteria, the query excluded
        others = ids[ids != q]
        cmin = crit[others].min()
        assert best == others[crit[others] == cmin].min()
        # 4. the pair-list path (transferBestHits) gives the same numbers as the sweep for the same pairs
        sample = rng.choice(ids, 4096, replace=False)
        d2, w2, c2 = ops.setDistCriterion(np.full(len(sample), q), sample, st.n_active, st.n_diff_allow, st.totdiam)
        assert np.array_equal(d2, dist[sample]) and np.array_equal(w2, weight[sample]) and np.array_equal(c2, crit[sample])
        # 5. symmetry: dist(q, j) == dist(j, q)
        d3, w3, _ = ops.setDistCriterion(sample, np.full(len(sample), q), st.n_active, st.n_diff_allow, st.totdiam)
        assert np.array_equal(d3, d2) and np.array_equal(w3, w2)
        # 6. oracle spot check on 64 sampled targets (leaf and internal)
        pq = ops.profile_download(q)
        diam = ops.get_node_scalars(0, st.maxnode)[0]
        for j in sample[:64]:
            j = int(j)
            if q < n and j < n:
                od, ow = orc.seqdist(codes[q], codes[j], nc, tables["distances"] if tables else None)
            else:
                od, ow = orc.profiledist(pq, ops.profile_download(j), dm=dm)
                od = dt(od - dt(diam[q] + diam[j]))
            assert od == dist[j] and ow == weight[j], (q, j)
        # 7. sharding: per-shard top-k merged == unsharded top-k (what the multi-GPU path relies on)
        parts = []
        for r in range(4):
            lo, hi = shard_range(st.maxnode, r, 4)
            ops.set_shard(lo, hi)
            h, _ = ops.setBestHit(q, st.n_active, st.n_diff_allow, st.totdiam, k, want_best=False)
            parts.append(h)
        ops.set_shard(0, st.maxnode)
        merged = merge_hits(parts, k)
        assert np.array_equal(merged["j"], hits["j"]) and np.array_equal(merged["criterion"], hits["criterion"])
        # the same merge on the device (what bench.py does after the RCCL all-gather)
        d_all = ops.device_buffer(np.concatenate(parts))
        dev = ops.merge_hits(d_all, 4, k)
        ops.device_free(d_all)
        assert np.array_equal(dev, hits)
    # 8. self distance of a leaf is 0 with weight = ungapped columns
    some = st.active[st.active < n][:256]
    d, w, _ = ops.setDistCriterion(some, some, st.n_active, st.n_diff_allow, st.totdiam)
    assert np.all(d == 0) and np.array_equal(w, (codes[some] != 127).sum(1).astype(dt))
    ops.close()


def test_join_then_sweep_round_trip_at_c3_size():
    """averageProfile -> out-profile update -> sweep from the new node: idempotent under re-evaluation, and the
    joined children disappear from the hit list."""
    n, L = 100000, 500
    codes, ops, st, _ = _state(n, L, 33)
    a, b = int(st.active[10]), int(st.active[11])
    new = st.maxnode
    ops.set_max_node(new + 1)
    ops.averageProfile([new], [a], [b])
    ops.set_parents(a, [new]); ops.set_parents(b, [new])
    ops.updateOutProfile(a, b, new, st.n_active)
    ops.set_out_distances(new, np.zeros(1, np.float32), [10 * n])
    n_active = st.n_active - 1
    h1, b1 = ops.setBestHit(new, n_active, int(n_active * 0.01), st.totdiam, 600)
    h2, b2 = ops.setBestHit(new, n_active, int(n_active * 0.01), st.totdiam, 600)
    assert np.array_equal(h1, h2) and b1 == b2
    assert a not in h1["j"] and b not in h1["j"] and new in h1["j"]
    assert np.all(np.diff(h1["criterion"]) >= 0)
    ops.close()


def test_bench_two_ranks_on_one_device_agree_with_one_rank():
    """bench.py's multi-rank flow (target-range shards, one batched all-gather per step, device merge) with two ranks
    sharing the one GPU of the test box over gloo (VFT_BENCH_SAME_DEVICE / VFT_BENCH_BACKEND hooks): it must run, and
    report the same number of profile-ops per step as the single-rank run of the same reduced workload."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VFT_BENCH_SAME_DEVICE="1", VFT_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = ["--steps", "2", "--warmup", "1", "--n-seqs", "200000", "--n-pos", "200", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, check=True, stdout=subprocess.PIPE,
                         timeout=600).stdout.decode()
    # no launcher around it: `bench.py --gpus 2` starts its two ranks itself, as the round driver calls it
    two = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"] + args,
                         check=True, stdout=subprocess.PIPE, env=env, timeout=600).stdout.decode()
    a = json.loads([l for l in one.splitlines() if l.startswith("{")][-1])
    b = json.loads([l for l in two.splitlines() if l.startswith("{")][-1])
    assert b["n_gpus"] == 2 and a["n_gpus"] == 1 and b["world"] == 2
    assert len(b["shards"]) == 2 and b["shards"][0][1] == b["shards"][1][0] and b["allgather_bytes_per_step"] > 0
    assert a["config"]["active_nodes"] == b["config"]["active_nodes"] and a["config"]["top_k"] == b["config"]["top_k"]
    assert b["value"] > 0
    assert a["hits_crc"] == b["hits_crc"]   # the merged lists of the two shards are the single-rank lists


def test_c3_tree_equals_the_reference_tree():
    """BASELINE config C3 end to end at full size and with its exact flags: 100 000 nt x 500, `-nt -fastest` (top hits
    WITH the second-level lists, as the reference runs -fastest at one thread), NJ phase, root, minimum-evolution lengths,
    Newick - the 2.5 MB tree must be the reference's byte for byte (CRC-32 and length of the reference's own output,
    tests/golden/bb_c3_crc.npz; the reference needs 640 s of one core for it).  This is what bench.py reports as `e2e`."""
    import zlib
    import golden_util as G
    from veryfasttree_amd import HipProfileOps, synth
    from veryfasttree_amd.backend import nj_newick
    ref = G.load("bb_c3_crc")
    n, L = 100000, 500
    codes = synth.random_descent_codes(n, L, 4, 0.03, 0.01, seed=3)
    names = ["s%d" % k for k in range(n)]
    tree = nj_newick(lambda m, Lp: HipProfileOps(m, Lp, 4, np.float32, max_nodes=3 * m), codes, names, fastest=True, me_lengths=True)
    assert len(tree) == int(ref["newick_bytes"])
    assert zlib.crc32(tree.encode()) == int(ref["newick_crc"])


@pytest.mark.parametrize("n", [200000, 400000])
def test_c4_generator_tree_equals_the_reference_tree(n):
    """Config C4's alignment generator and flags (`-nt`, default top hits) at the sizes the one-thread reference finishes in an hour
    or two: NJ phase, root, minimum-evolution lengths, Newick - byte for byte the reference's tree (CRC-32 and length of
    oracle/_ref/VeryFastTree's own output, oracle/gen_fixtures.py c4_scaled).  The million-sequence run itself is pinned by its join
    order as far as the one-thread reference got (bench.py e2e_c4, bb_c4_prefix.npz)."""
    import os
    import zlib
    import golden_util as G
    from veryfasttree_amd import HipProfileOps, synth
    from veryfasttree_amd.backend import nj_newick
    name = "bb_c4_%dk_crc" % (n // 1000)
    if not os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz")):
        pytest.skip("no fixture at this size")
    ref = G.load(name)
    codes = synth.random_descent_codes(n, 200, 4, 0.02, 0.01, seed=4)
    names = ["s%d" % k for k in range(n)]
    tree = nj_newick(lambda m, Lp: HipProfileOps(m, Lp, 4, np.float32, max_nodes=3 * m), codes, names, fastest=False, me_lengths=True)
    assert len(tree) == int(ref["newick_bytes"])
    assert zlib.crc32(tree.encode()) == int(ref["newick_crc"])


@pytest.mark.parametrize("mode", [[], ["fastest"], ["fastest", "second"]])
def test_nj_driver_join_order_is_rank_count_independent(mode):
    """The C++ NJ driver with its sweeps and leaf blocks split over two ranks (vft_comm over torch.distributed; both ranks
    on this box's one GPU, gloo): every rank must produce the join order of the single-rank run - the whole NJ state is
    replicated, only distance computations are divided, and merged top-k lists equal unsharded ones."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "tests", "run_nj_ranks.py")
    args = ["3000", "150"] + mode
    one = subprocess.run([sys.executable, script] + args, check=True, stdout=subprocess.PIPE, timeout=600).stdout.decode()
    env = dict(os.environ, VFT_SAME_DEVICE="1", VFT_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(free_port()), script] + args, check=True, stdout=subprocess.PIPE, env=env,
                         timeout=900).stdout.decode()
    want = re.search(r"rank 0 crc (\d+) joins (\d+)", one).groups()
    got = re.findall(r"rank (\d) crc (\d+) joins (\d+) allgathers (\d+)", two)
    assert len(got) == 2
    for r, crc, nj, calls in got:
        assert (crc, nj) == want, (r, crc, nj, want)
        assert int(calls) > 10          # the exchange really ran


@pytest.mark.gpu
def test_c4_join_order_equals_the_reference_trace():
    """Config C4 (1 000 000 x 200 nt, default settings): EVERY join of the NJ phase against the reference's own `Join` trace at one
    thread (`-verbose 3`; tests/golden/bb_c4_prefix.npz: CRC-32 of every 10 000 joins and of the last, shorter chunk of the finished
    trace, oracle/gen_fixtures.py c4_prefix), and the tree `VeryFastTree -nt -noml -nome -nosupport -threads 1` printed for it
    (bb_c4_crc.npz: its run ended in round 6, 43 695 s).  The top-hit lists, setAllLeafTopHits over 10^6 leaves, the join engine with
    m = 1 000, the top-hits refreshes and the late join loop (top-visible resets, refreshes at small nActive) all run at their full
    size here: two minutes."""
    import os, zlib
    from veryfasttree_amd import HipProfileOps, synth
    from veryfasttree_amd.backend import nj_newick, last_join_crcs
    gold = os.path.join(os.path.dirname(__file__), "golden")
    g = np.load(os.path.join(gold, "bb_c4_prefix.npz"))
    chunk, want = int(g["join_chunk"]), g["join_chunk_crc"]
    assert bytes(g["alignment"]).decode() == "random_descent_codes(1000000, 200, 4, 0.02, 0.01, seed=4)"
    codes = synth.random_descent_codes(1000000, 200, 4, 0.02, 0.01, seed=4)
    names = ["s%d" % k for k in range(len(codes))]
    tree = nj_newick(lambda m, Lp: HipProfileOps(m, Lp, 4, np.float32, max_nodes=3 * m), codes, names, me_lengths=True)
    got_chunk, n_joins, crcs = last_join_crcs()
    assert got_chunk == chunk
    n_cmp = min(len(want), len(crcs))
    if int(g["complete"]) if "complete" in g else False:   # the finished trace: to the last join
        assert n_joins == int(g["n_joins"]) and len(crcs) == len(want)
    else:                                                      # (a prefix: complete chunks only; this run's last entry is its shorter tail)
        n_cmp = min(len(want), len(crcs) - (1 if n_joins % chunk else 0))
    assert n_cmp >= 62
    bad = [k for k in range(n_cmp) if int(crcs[k]) != int(want[k])]
    assert not bad, "joins %d..%d differ from the reference's" % (bad[0] * chunk, (bad[0] + 1) * chunk)
    r = np.load(os.path.join(gold, "bb_c4_crc.npz"))   # the tree the one-thread reference printed after its 12 hours
    assert len(tree) == int(r["newick_bytes"]) and zlib.crc32(tree.encode()) == int(r["newick_crc"])


@heavy
def test_c3_complete_pipeline_equals_the_reference_tree():
    """Config C3's alignment (100 000 x 500 nt) through the complete default pipeline with its own flag, `VeryFastTree -nt -fastest` at one
    thread: the NJ phase with the second-level top-hit lists, ME NNIs + SPRs, ML NNIs under Jukes-Cantor + CAT, SH-like supports
    (oracle/gen_fixtures.py c3_full -> bb_c3_full_crc.npz).  Every TreeLogLk line within 1e-4 relative, the tree byte for byte."""
    import os, zlib
    from veryfasttree_amd import HipProfileOps, synth
    from veryfasttree_amd.backend import nj_newick
    ref = os.path.join(os.path.dirname(__file__), "golden", "bb_c3_full_crc.npz")
    if not os.path.exists(ref):
        pytest.skip("fixture not generated")
    g = np.load(ref)
    assert bytes(g["alignment"]).decode() == "random_descent_codes(100000, 500, 4, 0.03, 0.01, seed=3)"
    codes = synth.random_descent_codes(100000, 500, 4, 0.03, 0.01, seed=3)
    names = ["s%d" % k for k in range(len(codes))]
    tree, loglk = nj_newick(lambda m, Lp: HipProfileOps(m, Lp, 4, np.float32, max_nodes=3 * m), codes, names, fastest=True, me_lengths=True,
                            me_nni=True, spr=2, ml_nni=20, n_bootstrap=1000, return_loglk=True)
    want = g["loglk"]
    print("TreeLogLk", list(loglk), "reference", list(want))
    assert len(loglk) == len(want) and np.allclose(loglk, want, rtol=1e-4, atol=0)
    assert len(tree) == int(g["newick_bytes"]) and zlib.crc32(tree.encode()) == int(g["newick_crc"])


def _c2_alignment():
    from veryfasttree_amd import synth
    return synth.random_descent_codes(10000, 1000, 4, 0.03, 0.01, seed=2)


def test_c2_tree_equals_the_reference_tree():
    """BASELINE config C2 at full size with its exact flags (10 000 nt x 1 000, `-nt -gtr`, float32, one-thread order): the complete
    default pipeline - NJ, ME NNIs + SPRs, ML NNIs with CAT and the fitted GTR model, SH-like supports - must print the reference's
    289 KB tree byte for byte (CRC-32 and length of the reference binary's own output, tests/golden/bb_c2_crc.npz, oracle/gen_fixtures.py
    c2; 221 s of one core there).  This is bench.py's `e2e_c2`."""
    import zlib
    import golden_util as G
    from veryfasttree_amd import HipProfileOps
    from veryfasttree_amd.backend import nj_newick
    ref = G.load("bb_c2_crc")
    codes = _c2_alignment()
    names = ["s%d" % k for k in range(len(codes))]
    tree, loglk = nj_newick(lambda m, Lp: HipProfileOps(m, Lp, 4, np.float32, max_nodes=3 * m), codes, names, me_lengths=True, me_nni=True,
                            spr=2, ml_nni=20, gtr=True, n_bootstrap=1000, return_loglk=True)
    assert abs(loglk[-1] - ref["loglk"][-1]) <= 1e-4 * abs(ref["loglk"][-1])   # the north star's bar
    assert len(tree) == int(ref["newick_bytes"])
    assert zlib.crc32(tree.encode()) == int(ref["newick_crc"])


@pytest.mark.parametrize("fixture,mu,gap,seed", [("bb_c5_20k_crc", 0.08, 0.02, 5), pytest.param("bb_c5mu03_20k_crc", 0.03, 0.01, 2, marks=heavy)])
def test_c5_generator_tree_equals_the_reference_tree(fixture, mu, gap, seed):
    """BASELINE config C5's generator (mu 0.08, gaps 0.02, seed 5) and flags at 20 000 sequences (amino acids x 300, `-lg
    -double-precision`, one-thread order - the reference's deterministic path): the complete default pipeline must print the reference
    binary's 749 KB tree byte for byte and end at its log-likelihood (tests/golden/bb_c5_20k_crc.npz, oracle/gen_fixtures.py c5:20000;
    384 s of one core there).  The configuration itself - 50 000 sequences, bb_c5_crc.npz, 1 069 s of one core - is bench.py's `e2e_c5`.
    Second case: rounds 4-5's fixture, made on C2's generator parameters by mistake (less divergence, half the gaps) - kept."""
    import zlib
    import golden_util as G
    from veryfasttree_amd import HipProfileOps, synth
    from veryfasttree_amd.backend import nj_newick
    ref = G.load(fixture)
    assert bytes(ref["alignment"]).decode() == "random_descent_codes(20000, 300, 20, %g, %g, seed=%d)" % (mu, gap, seed)
    codes = synth.random_descent_codes(20000, 300, 20, mu, gap, seed=seed)
    names = ["s%d" % k for k in range(len(codes))]
    tree, loglk = nj_newick(lambda m, Lp: HipProfileOps(m, Lp, 20, np.float64, max_nodes=3 * m), codes, names, dtype=np.float64, aa_model="lg",
                            me_lengths=True, me_nni=True, spr=2, ml_nni=20, n_bootstrap=1000, return_loglk=True)
    assert abs(loglk[-1] - ref["loglk"][-1]) <= 1e-4 * abs(ref["loglk"][-1])   # the north star's bar
    assert len(tree) == int(ref["newick_bytes"])
    assert zlib.crc32(tree.encode()) == int(ref["newick_crc"])


def test_c2_shape_on_the_64_thread_schedule_equals_the_reference_run():
    """10 000 nt x 1 000 under Jukes-Cantor (config C2's shape; `-gtr` is not reproducible in the reference at T > 1) on the schedule of
    a 64-thread run: this backend (threads = 64: the walks of the reference's tree partitions in lockstep, host/MLLengths.h) against
    the compiled reference run HERE with `-threads 64` - the printed trees, SH supports included, byte for byte.  Needs oracle/_ref
    (travels with the snapshot) and a box with a few dozen cores; about 1.5 minutes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "oracle", "_ref", "VeryFastTree")):
        pytest.skip("the compiled reference is not on this box")
    if (os.cpu_count() or 1) < 16:
        pytest.skip("too few cores for a 64-thread run of the reference")
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "compare_with_reference_run.py"), "10000", "1000", "--threads", "64"],
                         check=True, stdout=subprocess.PIPE, timeout=1200).stdout.decode()
    print(out)
    assert "byte-identical output: YES" in out


@pytest.mark.parametrize("n,L,seed,mu,gap", [(100000, 500, 3, 0.03, 0.01), (1000000, 200, 4, 0.02, 0.01)])
def test_seeds_sharing_passes_at_full_size(n, L, seed, mu, gap):
    """BASELINE configs C3 and C4: a batch of 5 leaf + 5 profile seeds through the multi-seed passes (k_sweep_nt_leafq_multi,
    k_sweep_nt_profq_multi: 4 + 1 of each kind) gives the records of one launch per seed (VFT_DEBUG_NO_MULTI_SWEEP), every criterion of a
    seed of each kind equals the single sweep's, and every list is sorted under (criterion asc, id desc) without duplicates."""
    import ctypes
    codes, ops, st, _ = _state(n, L, seed, mu, gap)
    rng = np.random.default_rng(seed + 1)
    seeds = np.concatenate([rng.choice(st.active[st.active < n], 5, replace=False), rng.choice(st.active[st.active >= n], 5, replace=False)])
    rng.shuffle(seeds)
    k = 2 * int(0.5 + np.sqrt(n))
    hm, bm = ops.setBestHitBatch(seeds, st.n_active, st.n_diff_allow, st.totdiam, k)
    crit_first = ops.sweep_results(0, st.maxnode)
    assert ops.lib.vft_debug_option(ops.ctx, ctypes.c_int32(12), ctypes.c_int64(1)) == 0
    hs, bs = ops.setBestHitBatch(seeds, st.n_active, st.n_diff_allow, st.totdiam, k)
    assert np.array_equal(bm, bs) and np.array_equal(hm, hs)
    for a, b in zip(crit_first, ops.sweep_results(0, st.maxnode)):
        assert np.array_equal(a, b)
    for h in hm:
        key = np.lexsort((-h["j"].astype(np.int64), h["criterion"]))
        assert np.array_equal(key, np.arange(k)) and len(np.unique(h["j"])) == k
