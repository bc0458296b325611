"""Nearest-neighbour interchanges on the GPU backend: DoNNI (NJ.tcc:5797-6200) driven by the C++ host
(veryfasttree_amd/host/MLLengths.h::doNNI) against whole reference runs."""
import re

import numpy as np
import pytest

import golden_util as G

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,dt,spr", [("nni_nt_12", np.float32, 0), ("nni_nt_200", np.float32, 0), ("nni_nt_500", np.float32, 0),
                                         ("nni_nt_300_double", np.float64, 0), ("spr_nt_12", np.float32, 2),
                                         ("spr_nt_200", np.float32, 2), ("spr_nt_500", np.float32, 2),
                                         ("spr_nt_300_double", np.float64, 2)])
def test_min_evolution_nnis_match_the_reference_run(name, dt, spr):
    """`VeryFastTree -nt -noml [-spr 0] [-nosupport]`: fastNJ, rounds of minimum-evolution NNIs with 0 or the default 2
    rounds of SPR moves in between, ME branch lengths, local-bootstrap supports - the final trees byte for byte
    (topology after 5 ... 361 NNIs and up to 9 SPRs, lengths, supports)."""
    from veryfasttree_amd import HipProfileOps
    from veryfasttree_amd.backend import nj_newick
    d = G.load(name)
    codes_all = d["codes"]
    names = ["s%d" % k for k in range(len(codes_all))]
    make = lambda n, L: HipProfileOps(n, L, 4, dt, max_nodes=3 * n)
    tree = nj_newick(make, codes_all, names, dtype=dt, me_lengths=True, me_nni=True, spr=spr)
    ref = bytes(d["newick"]).decode().strip()
    strip = lambda t: re.sub(r":[0-9.eE+-]+", ":", t)
    assert strip(tree) == strip(ref), "topology differs"
    assert tree == ref
    boot = nj_newick(make, codes_all, names, dtype=dt, me_lengths=True, me_nni=True, spr=spr, n_bootstrap=1000)
    assert boot == bytes(d["newick_support"]).decode().strip()


@pytest.mark.parametrize("name,dt,ncat,me,spr", [("mlnni_nt_20", np.float32, 20, True, 0),
                                                 ("mlnni_nt_200_nocat", np.float32, 1, False, 0),
                                                 ("mlnni_nt_200", np.float32, 20, False, 0),
                                                 ("mlnni_nt_150_double", np.float64, 20, False, 0),
                                                 ("mlnni_nt_300_spr0", np.float32, 20, True, 0),
                                                 ("full_nt_200", np.float32, 20, True, 2),
                                                 ("full_nt_300", np.float32, 20, True, 2),
                                                 ("full_nt_250_double", np.float64, 20, True, 2),
                                                 ("full_nt_40_x1500", np.float32, 20, True, 2)])   # > 1024 columns: 8 per thread
def test_max_likelihood_nnis_match_the_reference_run(name, dt, ncat, me, spr):
    """`VeryFastTree -nt [-nome | -spr 0] [-nocat] [-nosupport]` and plain `VeryFastTree -nt` (full_*: the complete
    default pipeline - NJ, ME NNIs + SPRs, ML NNIs, CAT rates, SH-like supports): ML NNI rounds (DoNNI with MLQuartetNNI per node on the
    device), CAT rates after the first round, final length pass, SH-like supports.  TreeLogLk after every round within
    the north star's 1e-4 relative; the final topology is the reference's."""
    from veryfasttree_amd import HipProfileOps
    from veryfasttree_amd.backend import nj_newick
    d = G.load(name)
    codes_all = d["codes"]
    names = ["s%d" % k for k in range(len(codes_all))]
    make = lambda n, L: HipProfileOps(n, L, 4, dt, max_nodes=3 * n)
    tree, loglk = nj_newick(make, codes_all, names, dtype=dt, me_lengths=True, me_nni=me, spr=spr, ml_nni=ncat, return_loglk=True)
    want = d["loglk"]
    print(name, "rounds", len(loglk) - 1, "vs", len(want) - 1, "final", loglk[-1], want[-1])
    assert len(loglk) == len(want)
    assert np.allclose(loglk, want, rtol=1e-4, atol=0)
    ref = bytes(d["newick"]).decode().strip()
    strip = lambda t: re.sub(r":[0-9.eE+-]+", ":", t)
    assert strip(tree) == strip(ref), "topology differs"
    got_len = np.array([float(x) for x in re.findall(r":([0-9.eE+-]+)", tree)])
    ref_len = np.array([float(x) for x in re.findall(r":([0-9.eE+-]+)", ref)])
    assert np.allclose(got_len, ref_len, rtol=5e-3, atol=2e-5 if dt == np.float32 else 1e-8)
    print(name, "printed lengths differing:", int((got_len != ref_len).sum()), "of", len(ref_len))
    assert tree == ref     # observed on MI355X: byte-identical
    boot = nj_newick(make, codes_all, names, dtype=dt, me_lengths=True, me_nni=me, spr=spr, ml_nni=ncat, n_bootstrap=1000)
    refb = bytes(d["newick_support"]).decode().strip()
    sup = lambda t: np.array([float(x) for x in re.findall(r"\)([0-9.]+):", t)])
    assert re.sub(r"\)[0-9.]+:", "):", re.sub(r":[0-9.eE+-]+", ":", boot)) == re.sub(r"\)[0-9.]+:", "):", re.sub(r":[0-9.eE+-]+", ":", refb))
    ds = np.abs(sup(boot) - sup(refb))
    print(name, "supports differing:", int((ds > 0).sum()), "of", len(ds), "max", ds.max() if len(ds) else 0)
    assert ds.max() <= 0.002 + 1e-9
    assert boot == refb    # observed on MI355X: byte-identical, supports included


@pytest.mark.parametrize("name,dt,full", [("ml_nt_200_gtr", np.float32, False), ("ml_nt_150_double_gtr", np.float64, False),
                                          ("full_nt_200_gtr", np.float32, True), ("full_nt_250_double_gtr", np.float64, True),
                                          ("full_nt_30_x1300_gtr_double", np.float64, True)])   # > 1024 columns
def test_gtr_model_fitted_like_the_reference(name, dt, full):
    """`-gtr`: Jukes-Cantor for the first ML round, then setMLGtr (NJ.tcc:6436-6500) fits base frequencies and the six
    rates by line searches over the whole tree's likelihood and the run continues under GTR + CAT (BASELINE config C2's
    model).  ml_*: `-nome -mllen`; full_*: the complete default pipeline.
    float32: the ML kernels compute the reference's ordered likelihood total with glibc's log (vft_kernels_ml.h,
    vft_lk_total_ordered), every per-site likelihood is bit-identical, and the output is the reference's byte for byte
    (round 1 allowed 4 % of the splits to differ here).  double: the P(t) tables come from glibc's exp restated on the
    device (vft_glibc_exp), so the same holds."""
    from veryfasttree_amd import HipProfileOps
    from veryfasttree_amd.backend import nj_newick
    d = G.load(name)
    codes_all = d["codes"]
    names = ["s%d" % k for k in range(len(codes_all))]
    make = lambda n, L: HipProfileOps(n, L, 4, dt, max_nodes=3 * n)
    kw = dict(me_nni=True, spr=2, ml_nni=20) if full else dict(mllen=20)
    tree, loglk, rates, freq = nj_newick(make, codes_all, names, dtype=dt, me_lengths=True, gtr=True, return_gtr=True, **kw)
    print(name, "rates", np.round(rates, 4), "vs", d["gtr_rates"], "loglk", loglk, "vs", d["loglk"])
    assert np.allclose(freq, d["gtr_freq"], rtol=0, atol=5.1e-5)      # printed with %.4f; frequencies are exact counts
    assert np.allclose(rates, d["gtr_rates"], rtol=1e-3, atol=5.1e-5)   # printed with %.4f
    want = d["loglk"]
    assert len(loglk) == len(want)
    assert np.allclose(loglk, want, rtol=1e-4, atol=0)                # north star
    assert np.allclose(loglk, want, rtol=0, atol=6e-5)                # observed: every printed digit
    ref = bytes(d["newick"]).decode().strip()
    strip = lambda t: re.sub(r":[0-9.eE+-]+", ":", t)
    assert strip(tree) == strip(ref), "topology differs"
    got_len = np.array([float(x) for x in re.findall(r":([0-9.eE+-]+)", tree)])
    ref_len = np.array([float(x) for x in re.findall(r":([0-9.eE+-]+)", ref)])
    print(name, "printed lengths differing:", int((got_len != ref_len).sum()), "of", len(ref_len), "max abs", np.abs(got_len - ref_len).max())
    assert tree == ref            # byte-identical, float32 and double
    boot = nj_newick(make, codes_all, names, dtype=dt, me_lengths=True, gtr=True, n_bootstrap=1000, **kw)
    assert boot == bytes(d["newick_support"]).decode().strip()   # SH-like supports included


def _splits(newick):
    """Non-trivial bipartitions of an unrooted Newick tree as frozensets of leaf names (the side without the first leaf)."""
    names = re.findall(r"[(,]([^(),:;]+):", newick)
    first, allset = names[0], frozenset(names)
    stack, out = [], set()
    for tok in re.findall(r"\(|\)|[^(),:;]+(?=:)|,", newick):
        if tok == "(":
            stack.append(set())
        elif tok == ")":
            top = stack.pop()
            if stack:
                stack[-1] |= top
            side = frozenset(top) if first not in top else allset - frozenset(top)
            if 1 < len(side) < len(allset) - 1:
                out.add(side)
        elif tok != ",":
            if tok in allset and stack:
                stack[-1].add(tok)
    return out
