"""Join order of the host NJ driver running on the HIP backend, against the reference's `Join` lines
(integer join order bit-exact: the north star's NJ-phase parity bar)."""
import numpy as np
import pytest

import golden_util as G
from test_nj_driver_cpu import unique_codes
from nj_driver_py import NJDriver

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,fastest,limit", [("bb_nt_c1", True, None), ("bb_nt_200", False, None),
                                                ("bb_nt_600_fastest_no2nd", True, None),
                                                ("bb_nt_1500", False, None)])
def test_join_order_matches_reference_on_gpu(name, fastest, limit):
    from veryfasttree_amd import HipProfileOps
    d = G.load(name)
    codes = unique_codes(d["codes"])
    ops = HipProfileOps(codes.shape[0], codes.shape[1], 4, np.float32)
    drv = NJDriver(ops, codes, fastest=fastest)
    if fastest:
        drv.tophits_refresh = 0.5   # main.cpp:339-343: -fastest
    joins = drv.run(max_joins=limit)
    want = d["joins"][:len(joins)]
    got = np.array([(a, b, c) for a, b, c, _ in joins], dtype=np.int64)
    assert len(got) == len(d["joins"]) or limit is not None
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert len(bad) == 0, "first differing join %d: got %s want %s" % (bad[0], got[bad[0]], want[bad[0]])
    # criteria printed by the reference with %.6f
    crit = np.array([c for _, _, _, c in joins])
    assert np.allclose(crit, d["join_criterion"][:len(joins)], atol=1e-6)
    ops.close()


@pytest.mark.parametrize("name,fastest,second", [("bb_nt_c1", True, True), ("bb_nt_200", False, False),
                                                 ("bb_nt_600_fastest_no2nd", True, False),
                                                 ("bb_nt_600_fastest", True, True), ("bb_nt_1500", False, False),
                                                 ("bb_nt_300_double", False, False)])
def test_cpp_host_driver_join_order_on_gpu(name, fastest, second):
    """The C++ host driver (veryfasttree_amd/host/NJDriver.h through include/vft_host.h) over the C ABI."""
    from veryfasttree_amd import HipProfileOps
    from veryfasttree_amd.backend import nj_run
    d = G.load(name)
    codes = unique_codes(d["codes"])
    ops = HipProfileOps(codes.shape[0], codes.shape[1], 4, np.float64 if "double" in name else np.float32)
    joins, crit = nj_run(ops, codes, fastest=fastest, second_level=second)
    want = d["joins"]
    assert len(joins) == len(want)
    bad = np.nonzero((joins != want).any(axis=1))[0]
    assert len(bad) == 0, "first differing join %d: got %s want %s" % (bad[0], joins[bad[0]], want[bad[0]])
    assert np.allclose(crit, d["join_criterion"], atol=1e-6)
    ops.close()


@pytest.mark.parametrize("name,fastest,second,dt", [("bb_nt_c1", True, True, np.float32), ("bb_nt_200", False, False, np.float32),
                                                    ("bb_nt_600_fastest_no2nd", True, False, np.float32),
                                                    ("bb_nt_600_fastest", True, True, np.float32),
                                                    ("bb_nt_1500", False, False, np.float32),
                                                    ("bb_nt_300_double", False, False, np.float64)])
def test_nj_tree_string_equals_the_reference(name, fastest, second, dt):
    """End to end: duplicates included, fastNJ to the root, branch lengths, printNJ -- the exact "NJ\t<tree>" line of the
    reference's log (NJ.tcc:2706-2794, 3098-3120)."""
    from veryfasttree_amd import HipProfileOps
    from veryfasttree_amd.backend import nj_newick
    d = G.load(name)
    codes_all = d["codes"]
    names = ["s%d" % k for k in range(len(codes_all))]
    tree = nj_newick(lambda n, L: HipProfileOps(n, L, 4, dt), codes_all, names, fastest=fastest, second_level=second, dtype=dt)
    assert tree == bytes(d["nj_newick"]).decode()
    # ... and with the minimum-evolution branch lengths (updateBranchLengths, NJ.tcc:6514-6595): the reference's final
    # output for -noml -nome -nosupport
    final = nj_newick(lambda n, L: HipProfileOps(n, L, 4, dt, max_nodes=3 * n), codes_all, names, fastest=fastest,
                      second_level=second, dtype=dt, me_lengths=True)
    assert final == bytes(d["newick"]).decode().strip()
    # ... and with the default local-bootstrap supports (reliabilityNJ, 1000 resamples of Knuth's generator): the
    # reference's output for -noml -nome
    boot = nj_newick(lambda n, L: HipProfileOps(n, L, 4, dt, max_nodes=3 * n), codes_all, names, fastest=fastest,
                     second_level=second, dtype=dt, me_lengths=True, n_bootstrap=1000)
    assert boot == bytes(d["newick_support"]).decode().strip()


def test_fasta_to_newick_tool_matches_the_reference_output():
    """tools/nj_tree.py on the FASTA of bb_nt_200 prints what `VeryFastTree -nt -noml -nome [-nosupport]` printed."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fa = os.path.join(root, "tests", "golden", "bb_nt_200.fa")
    for flags, key in (([], "newick_support"), (["-nosupport"], "newick")):
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "nj_tree.py"), fa] + flags, check=True,
                             stdout=subprocess.PIPE, timeout=300).stdout.decode().strip()
        assert out == bytes(G.load("bb_nt_200")[key]).decode().strip()
    # the same alignment through `-nome -mllen` without and with the CAT approximation (ml_nt_200*: same sequences)
    for flags, name, key in ((["-mllen", "-nocat", "-nosupport"], "ml_nt_200", "newick"),
                             (["-mllen", "-nosupport"], "ml_nt_200_cat", "newick"),
                             (["-mllen"], "ml_nt_200_cat", "newick_support"),   # the reference's default for -nome -mllen
                             (["-full"], "full_nt_200", "newick_support")):      # plain `VeryFastTree -nt`
        res = subprocess.run([sys.executable, os.path.join(root, "tools", "nj_tree.py"), fa] + flags, check=True,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        d = G.load(name)
        assert res.stdout.decode().strip() == bytes(d[key]).decode().strip()
        got = [float(l.split("\t")[2]) for l in res.stderr.decode().splitlines() if l.startswith("TreeLogLk")]
        assert np.allclose(got, d["loglk"], rtol=0, atol=6e-5)   # (-full: the ML NNI rounds and the final length pass)


def test_amino_acid_alignment_with_distance_matrix_end_to_end():
    """Protein alignment, BLOSUM45-derived distance matrix (the reference's default for amino acids): join order, the
    NJ tree and the final -noml -nome -nosupport tree, through the generic (any alphabet / matrix) kernels."""
    from veryfasttree_amd import HipProfileOps
    from veryfasttree_amd.backend import nj_newick, nj_run
    d = G.load("bb_aa_300")
    dm = G.load("wb_aa_f32")

    def make(n, L):
        ops = HipProfileOps(n, L, 20, np.float32, max_nodes=3 * n)
        ops.set_distance_matrix(dm["dmat.distances"], dm["dmat.codefreq"], dm["dmat.eigenval"], dm["dmat.eigentot"])
        return ops

    codes = unique_codes(d["codes"])
    joins, crit = nj_run(make(*codes.shape), codes)
    assert np.array_equal(joins, d["joins"])
    names = ["s%d" % k for k in range(len(d["codes"]))]
    assert nj_newick(make, d["codes"], names, scoredist=True) == bytes(d["nj_newick"]).decode()
    assert nj_newick(make, d["codes"], names, scoredist=True, me_lengths=True) == bytes(d["newick"]).decode().strip()
    assert nj_newick(make, d["codes"], names, scoredist=True, me_lengths=True, n_bootstrap=1000) == \
        bytes(d["newick_support"]).decode().strip()


@pytest.mark.parametrize("name,fastest", [("bb_nt_10", False), ("bb_nt_5", False), ("bb_nt_12_fastest", True)])
def test_tiny_inputs_without_top_hits(name, fastest):
    """Fewer than 13 sequences: fastNJ runs without top hits (NJ.tcc:2827-2834) - every node keeps its best hit, found by
    one-vs-all sweeps, with hill climbing unless -fastest.  Join order and all three tree strings."""
    from veryfasttree_amd import HipProfileOps
    from veryfasttree_amd.backend import nj_newick, nj_run
    d = G.load(name)
    codes = unique_codes(d["codes"])
    assert len(codes) == len(d["codes"])
    joins, crit = nj_run(HipProfileOps(codes.shape[0], codes.shape[1], 4, np.float32), codes, fastest=fastest)
    assert np.array_equal(joins, d["joins"])
    names = ["s%d" % k for k in range(len(codes))]
    make = lambda n, L: HipProfileOps(n, L, 4, np.float32, max_nodes=3 * n)
    assert nj_newick(make, codes, names, fastest=fastest) == bytes(d["nj_newick"]).decode()
    assert nj_newick(make, codes, names, fastest=fastest, me_lengths=True) == bytes(d["newick"]).decode().strip()
    assert nj_newick(make, codes, names, fastest=fastest, me_lengths=True, n_bootstrap=1000) == \
        bytes(d["newick_support"]).decode().strip()


def test_no_top_hits_beyond_the_sorted_hit_buffer():
    """`-notop` on 4395 unique sequences: maxnode passes 8192 (the device's sorted-hit buffer) half-way through, so every
    node's best hit has to come from the full allhits[] arrays of the sweep (vft_sweep_results) - join order of all 4392
    joins and the NJ tree."""
    from veryfasttree_amd import HipProfileOps
    from veryfasttree_amd.backend import nj_newick, nj_run
    d = G.load("bb_nt_4400_notop")
    codes = unique_codes(d["codes"])
    assert 2 * len(codes) > 8192
    joins, crit = nj_run(HipProfileOps(codes.shape[0], codes.shape[1], 4, np.float32), codes, tophits_mult=-1.0)
    assert np.array_equal(joins, d["joins"])
