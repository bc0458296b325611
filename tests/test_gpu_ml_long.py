"""The ML line searches beyond the register-resident kernels (csrc/vft_kernels_ml_long.h: alignments of more than 2 048 columns keep the
posteriors of a search in a per-workgroup workspace) and Jukes-Cantor likelihoods to the last bit (vft_set_jc_exact).

Both are checked the strict way: the reference's own fixtures - complete pipelines of `oracle/_ref/VeryFastTree`, trees with SH-like supports -
must come out byte for byte when every line search is FORCED through the workspace kernels (VFT_DEBUG_ML_LONG: the same searches, the same
numbers, at any length), and when the Jukes-Cantor totals are the reference's ordered product instead of per-thread partial products."""
import ctypes
import re

import numpy as np
import pytest

import golden_util as G

pytestmark = pytest.mark.gpu
VFT_DEBUG_ML_LONG = 16
AA = {"-lg": "lg", "-wag": "wag"}


def run_fixture(name, long_kernels, jc_exact, n_bootstrap=1000):
    from veryfasttree_amd import HipProfileOps
    from veryfasttree_amd.backend import nj_newick
    d = G.load(name)
    flags = bytes(d["flags"]).decode().split()
    codes = d["codes"]
    nt = "-nt" in flags
    dt = np.float64 if "-double-precision" in flags else np.float32
    names = ["s%d" % k for k in range(len(codes))]

    def make(n, L):
        ops = HipProfileOps(n, L, 4 if nt else 20, dt, max_nodes=3 * n)
        if long_kernels:
            assert ops.lib.vft_debug_option(ops.ctx, ctypes.c_int32(VFT_DEBUG_ML_LONG), ctypes.c_int64(1)) == 0
        if jc_exact is not None:
            assert ops.lib.vft_set_jc_exact(ops.ctx, ctypes.c_int32(jc_exact)) == 0
        return ops

    kw = dict(dtype=dt, me_lengths=True, n_bootstrap=n_bootstrap, threads=int(d["threads"]) if "threads" in d else 1)
    if not nt:
        kw["aa_model"] = next((AA[f] for f in flags if f in AA), "jtt")
    if "-mllen" in flags or name.startswith("ml_"):   # (the protein `-nome -mllen` fixtures keep their kind in the name)
        kw.update(mllen=1 if "-nocat" in flags else 20)
    else:
        spr = int(flags[flags.index("-spr") + 1]) if "-spr" in flags else 2
        kw.update(me_nni="-nome" not in flags, spr=0 if "-nome" in flags else spr, ml_nni=1 if "-nocat" in flags else 20, gtr="-gtr" in flags)
    tree = nj_newick(make, codes, names, **kw)
    key = "newick_support" if n_bootstrap else "newick"
    return tree, bytes(d[key]).decode().strip()


# Jukes-Cantor (float, double, > 1 024 columns), GTR (float; double at 1 300 columns), LG / JTT proteins (the whole-column protein kernels'
# workspace twins: the quad-of-lanes kernels have none), `-mllen` alone (k_ml_node_lengths_long only), the 8-thread schedule (quartets in
# batches: k_ml_quartet_long's mode 2, a workgroup per pairing)
@pytest.mark.parametrize("name", ["full_nt_200", "full_nt_250_double", "full_nt_40_x1500", "full_nt_200_gtr", "full_nt_30_x1300_gtr_double",
                                  "full_aa_120_lg_double", "full_aa_100_jtt", "ml_aa_100_lg_double", "thr_full_nt_600_t8"])
def test_workspace_line_searches_print_the_references_tree(name):
    tree, ref = run_fixture(name, long_kernels=True, jc_exact=None)
    strip = lambda t: re.sub(r"\)[0-9.]+:", "):", re.sub(r":[0-9.eE+-]+", ":", t))
    assert strip(tree) == strip(ref), "topology differs"
    assert tree == ref


# every Jukes-Cantor pipeline fixture with the totals bit for bit the reference's: register-resident kernels, and the workspace kernels
@pytest.mark.parametrize("name,long_kernels", [("full_nt_200", False), ("full_nt_300", False), ("full_nt_250_double", False), ("full_nt_40_x1500", False),
                                               ("mlnni_nt_200_nocat", False), ("thr_full_nt_600_t8", False), ("thr_full_nt_1500_t32", False),
                                               ("full_nt_200", True), ("full_nt_250_double", True)])
def test_exact_jukes_cantor_totals_print_the_references_tree(name, long_kernels):
    tree, ref = run_fixture(name, long_kernels=long_kernels, jc_exact=1)
    assert tree == ref


# no option here: the driver's calls pick the workspace kernels by themselves once the alignment is longer than the register-resident
# instances hold (2 048 columns; the protein quad-of-lanes kernels stop at 512, the whole-column ones at 2 048)
@pytest.mark.parametrize("name", ["full_nt_40_x3000", "full_nt_30_x2500_gtr", "full_nt_24_x5000_double", "full_aa_30_x2200_lg_double"])
def test_alignments_beyond_2048_columns_print_the_references_tree(name):
    """`VeryFastTree [-nt] [-gtr | -lg] [-double-precision]` on 2 200 - 5 000 columns (oracle/gen_fixtures.py mlnni:<name> / aa:<name>): the
    complete default pipeline, tree and SH-like supports byte for byte."""
    tree, ref = run_fixture(name, long_kernels=False, jc_exact=None)
    strip = lambda t: re.sub(r"\)[0-9.]+:", "):", re.sub(r":[0-9.eE+-]+", ":", t))
    assert strip(tree) == strip(ref), "topology differs"
    assert tree == ref


def test_fast_jukes_cantor_totals_still_print_the_references_tree_where_they_did():
    """vft_set_jc_exact(ctx, 0): rounds 1-5's arithmetic (per-thread partial products) stays available and keeps its fixtures"""
    for name in ("full_nt_200", "thr_full_nt_600_t8"):
        tree, ref = run_fixture(name, long_kernels=False, jc_exact=0)
        assert tree == ref
