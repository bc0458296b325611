"""The subtree schedule (veryfasttree_amd/host/MLLengths.h: doNNIThreaded, optimizeRoundThreaded) against whole runs of the
reference with `-threads T` (oracle/gen_fixtures.py threads): the reference partitions the tree (treePartitioning,
NJ.tcc:5540-5750) and lets T OpenMP threads walk the subtrees; this backend advances all the walks in lockstep as batches of
quartets / splits on one GPU.  Same schedule, same arithmetic: the trees must come out byte for byte."""
import re

import numpy as np
import pytest

import golden_util as G
from conftest import free_port, heavy

pytestmark = pytest.mark.gpu

AA = {"-lg": "lg", "-wag": "wag"}


def run_case(name, n_bootstrap=0):
    from veryfasttree_amd import HipProfileOps
    from veryfasttree_amd.backend import nj_newick
    d = G.load(name)
    flags = bytes(d["flags"]).decode().split()
    codes_all = d["codes"]
    nt = "-nt" in flags
    dt = np.float64 if "-double-precision" in flags else np.float32
    names = ["s%d" % k for k in range(len(codes_all))]
    make = lambda n, L: HipProfileOps(n, L, 4 if nt else 20, dt, max_nodes=3 * n)
    kw = dict(dtype=dt, me_lengths=True, threads=int(d["threads"]), n_bootstrap=n_bootstrap, return_loglk=True)
    if not nt:
        kw["aa_model"] = next((AA[f] for f in flags if f in AA), "jtt")
    if "-noml" in flags:
        kw.update(me_nni=True, spr=2)
    elif "-mllen" in flags:
        kw.update(mllen=20)
    else:
        kw.update(me_nni="-nome" not in flags, spr=0 if "-nome" in flags else 2, ml_nni=20, gtr="-gtr" in flags)
    if "-noml" in flags:
        kw.pop("return_loglk")
        return d, nj_newick(make, codes_all, names, **kw), []
    tree, loglk = nj_newick(make, codes_all, names, **kw)
    return d, tree, loglk


@pytest.mark.parametrize("name", ["thr_menni_nt_400_t4", "thr_mllen_nt_300_t4", "thr_full_nt_400_t4", "thr_full_nt_600_t8",
                                  "thr_full_nt_300_double_t3", "thr_full_nt_500_t16", "thr_full_nt_1500_t32"])
def test_trees_match_the_reference_at_T_threads(name):
    """`VeryFastTree [-nt] [flags] -threads T -nosupport`: TreeLogLk after every stage within the north star's 1e-4 relative
    (observed: to the printed digit), topology identical, the printed tree byte-identical."""
    d, tree, loglk = run_case(name)
    want = d["loglk"]
    ref = bytes(d["newick"]).decode().strip()
    print(name, "T =", int(d["threads"]), "TreeLogLk", list(loglk), "reference", list(want),
          "(the reference's tree at T threads", "differs from" if int(d["differs_from_one_thread"]) else "equals", "its one-thread tree)")
    if len(want):
        assert len(loglk) == len(want)
        assert np.allclose(loglk, want, rtol=1e-4, atol=0)
    strip = lambda t: re.sub(r":[0-9.eE+-]+", ":", t)
    assert strip(tree) == strip(ref), "topology differs"
    assert tree == ref


@pytest.mark.parametrize("name", ["thr_full_nt_400_t4", "thr_full_nt_600_t8"])
def test_supports_match_the_reference_at_T_threads(name):
    """the same runs with the default 1000 resamples: SH-like supports included, byte for byte"""
    d, tree, _ = run_case(name, n_bootstrap=1000)
    assert tree == bytes(d["newick_support"]).decode().strip()


@pytest.mark.parametrize("name", ["thrx_full_aa_300_lg_double_t8", "thrx_full_nt_500_gtr_t16"])
def test_matrix_models_stay_within_the_references_own_spread(name):
    """Matrix models with threads > 1: the reference itself is not reproducible (oracle/gen_fixtures.py THREADS_CASES - three runs,
    three trees), so there is no tree to pin.  The schedule is pinned by the Jukes-Cantor cases above, the matrix arithmetic by the
    one-thread fixtures (test_gpu_nni.py, test_gpu_aa.py); here: the first two and the last TreeLogLk within the north star's 1e-4
    relative of the span of the reference's three runs, and two runs of this backend identical."""
    d, tree, loglk = run_case(name)
    runs = d["loglk_runs"]
    print(name, "T =", int(d["threads"]), "TreeLogLk", list(loglk), "reference runs (final)", list(runs[:, -1]), "distinct reference trees", int(d["distinct_trees"]))
    n = min(len(loglk), runs.shape[1])
    assert n >= 2
    # (the three GTR runs of the reference end 1.8e-4 apart: the envelope of its runs, widened by the bar)
    for k in (0, 1, -1):
        lo, hi = runs[:, k].min(), runs[:, k].max()
        assert lo - 1e-4 * abs(lo) <= loglk[k] <= hi + 1e-4 * abs(hi), (k, loglk[k], lo, hi)
    _, again, _ = run_case(name)
    assert again == tree


@pytest.mark.parametrize("fixture,n,identical", [pytest.param("thr_c4_200k_t64_crc", 200000, True, marks=heavy), ("thr_c4_100k_t64_crc", 100000, True),
                                                 pytest.param("thr_c4_100k_t1_crc", 100000, True, marks=heavy),
                                                 pytest.param("thr_c4_200k_t1_crc", 200000, True, marks=heavy),
                                                 pytest.param("thr_c4_400k_t64_crc", 400000, False, marks=heavy)])
def test_the_64_thread_schedule_far_beyond_the_toy_fixtures(fixture, n, identical):
    """The complete default pipeline far beyond the toy fixtures: config C4's generator at n sequences, `VeryFastTree -nt -threads T -seed 1`
    (NJ, ME NNIs + SPRs, ML NNIs under Jukes-Cantor + CAT, SH-like supports; oracle/gen_fixtures.py thrbig:<n>:<T> ->
    thr_c4_<n/1000>k_t<T>_crc.npz: CRC-32 and length of the tree, every TreeLogLk line; 985 s on the build container's eight cores at
    100 000 / T = 64, 2 273 s at 200 000 / T = 64, 950 s at 100 000 / T = 1).  Every TreeLogLk line within 1e-4 relative (observed: to
    the printed digit) and the tree BYTE FOR BYTE - on the 64-thread schedule and in the one-thread order.  Round 6's history: with the
    per-thread Jukes-Cantor sums of rounds 1-5 these trees had the reference's length and likelihoods and differed in a few dozen leaf
    placements on zero-length branches (exact ties of an ML NNI: 24 places at 100 000, 44 splits at 200 000 - at first put down to the
    reference's threaded NJ phase, which at 200 000 sequences and 64 threads differs from its one-thread NJ tree in one split); with the
    totals as the reference's ordered product (vft_set_jc_exact, the default) all of them are identical - at 200 000 too: the one
    different split of the reference's NJ tree does not survive its own ME NNIs / SPRs.  At 400 000 sequences and 64 threads (5 502 s for the
    reference, 177 s here) the two runs DO part: same length, all eight TreeLogLk lines within 4e-8 relative (the first one -7277475.29
    here, -7277475.01 there), different trees (identical = False: likelihoods and length only).  The reference's threaded NJ phase
    leaves its own one-thread order from 200 000 sequences on (one split there) while this backend's NJ phase IS the one-thread order; at
    400 000 sequences its NJ-only trees at 64 threads and at one thread differ from character 890 573 of 10 502 875 on (5 115 s run, the
    fixture's reference_nj_equals_its_one_thread_nj = 0) - far more than its ME NNIs / SPRs wash out."""
    import os
    import zlib
    from veryfasttree_amd import HipProfileOps, synth
    from veryfasttree_amd.backend import nj_newick
    if not os.path.exists(os.path.join(os.path.dirname(__file__), "golden", fixture + ".npz")):
        pytest.skip("fixture not generated")
    d = G.load(fixture)
    assert bytes(d["alignment"]).decode() == "random_descent_codes(%d, 200, 4, 0.02, 0.01, seed=4)" % n
    codes = synth.random_descent_codes(n, 200, 4, 0.02, 0.01, seed=4)
    names = ["s%d" % k for k in range(len(codes))]
    tree, loglk = nj_newick(lambda m, Lp: HipProfileOps(m, Lp, 4, np.float32, max_nodes=3 * m), codes, names, me_lengths=True, me_nni=True, spr=2,
                            ml_nni=20, n_bootstrap=1000, return_loglk=True, threads=int(d["threads"]))
    want = d["loglk"]
    same = zlib.crc32(tree.encode()) == int(d["newick_crc"])
    print("TreeLogLk", list(loglk), "reference", list(want), "byte-identical" if same else "NOT byte-identical",
          "(reference NJ at T threads == its one-thread NJ: %d)" % int(d["reference_nj_equals_its_one_thread_nj"]))
    assert len(loglk) == len(want) and np.allclose(loglk, want, rtol=1e-4, atol=0)
    assert len(tree) == int(d["newick_bytes"])
    assert same or not identical


def test_one_thread_is_untouched_by_the_option():
    """threads = 1 is the one-thread order: the fixture of the sequential walk, through the same entry point"""
    from veryfasttree_amd import HipProfileOps
    from veryfasttree_amd.backend import nj_newick
    d = G.load("full_nt_200")
    codes_all = d["codes"]
    names = ["s%d" % k for k in range(len(codes_all))]
    make = lambda n, L: HipProfileOps(n, L, 4, np.float32, max_nodes=3 * n)
    tree = nj_newick(make, codes_all, names, me_lengths=True, me_nni=True, spr=2, ml_nni=20, threads=1)
    assert tree == bytes(d["newick"]).decode().strip()


@pytest.mark.parametrize("name,dt,aa", [("spr_nt_500", np.float32, None), ("spr_nt_300_double", np.float64, None), ("nni_aa_150", np.float32, "jtt")])
def test_spr_walk_without_the_walk_server_gives_the_reference_tree(name, dt, aa):
    """The SPR / NNI walks with the resident walk server switched off (vft_nj_options.debug_flags & VFT_NJ_DEBUG_NO_WALK_SERVER): every step
    is then the two plain calls (vft_average_chain + vft_profile_distances) - `VeryFastTree [-nt] -noml` with the default two SPR rounds,
    byte for byte, as through the server (the default, every other test of this file)."""
    from veryfasttree_amd import HipProfileOps
    from veryfasttree_amd.backend import nj_newick, DEBUG_NO_WALK_SERVER, last_stage_seconds
    d = G.load(name)
    codes_all = d["codes"]
    names = ["s%d" % k for k in range(len(codes_all))]
    make = lambda n, L: HipProfileOps(n, L, 20 if aa else 4, dt, max_nodes=3 * n)
    kw = dict(dtype=dt, me_lengths=True, me_nni=True, spr=2, debug_flags=DEBUG_NO_WALK_SERVER)
    if aa:
        kw["aa_model"] = aa
    tree = nj_newick(make, codes_all, names, **kw)
    st = last_stage_seconds()
    assert st["spr_steps"] > 0
    assert tree == bytes(d["newick"]).decode().strip()


@pytest.mark.parametrize("name,dt,aa", [("spr_nt_500", np.float32, None), ("spr_nt_300_double", np.float64, None), ("nni_aa_150", np.float32, "jtt")])
def test_spr_chains_with_both_continuations_on_the_device_give_the_reference_tree(name, dt, aa):
    """SPR chains hand BOTH continuations of a chain step to the walk server (vft_walk_submit_dual; host/MLLengths.h specContinuations): the
    resident workgroups compare the step's six distances themselves (logCorrect with glibc's log) and run the continuation that the
    host - making the same comparison when the answer arrives - moves its state to; a disagreement throws.  `VeryFastTree [-nt] -noml`
    with the default two SPR rounds, byte for byte, with the dual commands (the default: they must really have been taken) and with every
    step waiting for the host's verdict (VFT_NJ_DEBUG_NO_WALK_DUAL)."""
    from veryfasttree_amd import HipProfileOps
    from veryfasttree_amd.backend import nj_newick, DEBUG_NO_WALK_DUAL, last_stage_seconds
    d = G.load(name)
    codes_all = d["codes"]
    names = ["s%d" % k for k in range(len(codes_all))]
    make = lambda n, L: HipProfileOps(n, L, 20 if aa else 4, dt, max_nodes=3 * n)
    kw = dict(dtype=dt, me_lengths=True, me_nni=True, spr=2)
    if aa:
        kw["aa_model"] = aa
    ref = bytes(d["newick"]).decode().strip()
    tree = nj_newick(make, codes_all, names, **kw)
    st = last_stage_seconds()
    print(name, "SPR steps", st["spr_steps"], "dual commands", st["spr_dual_commands"], "continuations run by the device", st["spr_dual_continuations"])
    assert tree == ref
    assert st["spr_dual_commands"] > 100 and st["spr_dual_continuations"] > 50
    tree = nj_newick(make, codes_all, names, debug_flags=DEBUG_NO_WALK_DUAL, **kw)
    st = last_stage_seconds()
    assert tree == ref
    assert st["spr_dual_commands"] == 0


@pytest.mark.parametrize("name", ["thr_full_nt_1500_t32", "thr_menni_nt_400_t4", "thr_mllen_nt_300_t4"])
def test_lanes_across_two_ranks_give_the_single_rank_tree(name):
    """The lanes of the subtree schedule split over two ranks (host/MLLengths.h "lanes across ranks": every rank judges its share of a
    lockstep step's quartets / splits / distances, the verdicts and the branch lengths the kernels wrote are all-gathered, the others'
    lengths scattered; both ranks on this box's one GPU, gloo): every rank must print the reference's tree at T threads - the tree of
    the single-rank run - byte for byte, and the exchange must really have run."""
    import os
    import subprocess
    import sys
    import zlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "tests", "run_pipeline_ranks.py")
    d = G.load(name)
    ref = bytes(d["newick"]).decode().strip()
    env = dict(os.environ, VFT_SAME_DEVICE="1", VFT_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(free_port()), script, name], check=True, stdout=subprocess.PIPE, env=env,
                         timeout=900).stdout.decode()
    got = re.findall(r"rank (\d) crc (\d+) bytes (\d+) lane_allgathers (\d+) lane_bytes (\d+) loglk (\S*) end", two)
    assert len(got) == 2, two
    for r, crc, nb, calls, nbytes, ll in got:
        assert (int(crc), int(nb)) == (zlib.crc32(ref.encode()), len(ref)), (r, crc, nb)
        assert int(calls) > 10 and int(nbytes) > 1000          # the exchange really ran
    # treeLogLk with its pair likelihoods split over the ranks (MLLengths::treeLogLk): the same doubles on both ranks, the reference's lines
    assert got[0][5] == got[1][5]
    if got[0][5] and "loglk" in d:
        mine = [float.fromhex(x) for x in got[0][5].split(",")]
        assert len(mine) == len(d["loglk"]) and np.allclose(mine, d["loglk"], rtol=0, atol=1e-3), (mine, list(d["loglk"]))
