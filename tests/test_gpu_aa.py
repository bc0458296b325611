"""Protein alignments end to end on the GPU backend (BASELINE config C5's path): BLOSUM45-derived distances in the NJ /
minimum-evolution phase, the built-in JTT / WAG / LG models (host/AAModels.h) + CAT in the ML phase, against whole
reference runs (oracle/gen_fixtures.py aa).  The driver installs every matrix itself (vft_nj_options.aa_model)."""
import re

import numpy as np
import pytest

import golden_util as G

pytestmark = pytest.mark.gpu

strip = lambda t: re.sub(r":[0-9.eE+-]+", ":", t)
lengths = lambda t: np.array([float(x) for x in re.findall(r":([0-9.eE+-]+)", t)])
supports = lambda t: np.array([float(x) for x in re.findall(r"\)([0-9.]+):", t)])
no_support = lambda t: re.sub(r"\)[0-9.]+:", "):", strip(t))


def _make(dt):
    from veryfasttree_amd import HipProfileOps
    return lambda n, L: HipProfileOps(n, L, 20, dt, max_nodes=3 * n)


def _model(d):
    flags = bytes(d["flags"]).decode().split()
    return "lg" if "-lg" in flags else "wag" if "-wag" in flags else "jtt", \
        np.float64 if "-double-precision" in flags else np.float32, 1 if "-nocat" in flags else 20


def test_min_evolution_nnis_and_sprs_on_proteins():
    """`VeryFastTree -noml` on a protein alignment: fastNJ, ME NNIs and SPRs under the BLOSUM45-derived distance matrix
    with the scoredist log-correction, ME lengths, local-bootstrap supports - byte for byte."""
    from veryfasttree_amd.backend import nj_newick
    d = G.load("nni_aa_150")
    names = ["s%d" % k for k in range(len(d["codes"]))]
    tree = nj_newick(_make(np.float32), d["codes"], names, me_lengths=True, me_nni=True, spr=2, aa_model="jtt")
    ref = bytes(d["newick"]).decode().strip()
    assert strip(tree) == strip(ref), "topology differs"
    assert tree == ref
    boot = nj_newick(_make(np.float32), d["codes"], names, me_lengths=True, me_nni=True, spr=2, aa_model="jtt", n_bootstrap=1000)
    assert boot == bytes(d["newick_support"]).decode().strip()


@pytest.mark.parametrize("name", ["ml_aa_100_lg_double", "ml_aa_120_jtt", "ml_aa_80_wag_nocat"])
def test_ml_lengths_on_proteins(name):
    """`VeryFastTree [-lg | -wag] [-double-precision] -nome -mllen`: the profiles re-averaged in the model's eigen-basis
    (recomputeProfiles(tmatAsDist)), ML lengths on the NJ topology, CAT rates, SH-like supports."""
    from veryfasttree_amd.backend import nj_newick
    d = G.load(name)
    model, dt, ncat = _model(d)
    names = ["s%d" % k for k in range(len(d["codes"]))]
    tree, loglk, rates, ratecat = nj_newick(_make(dt), d["codes"], names, dtype=dt, me_lengths=True, mllen=ncat, aa_model=model,
                                            return_rates=True)
    print(name, "loglk", loglk, "vs", d["loglk"])
    assert len(loglk) == len(d["loglk"])
    assert np.allclose(loglk, d["loglk"], rtol=1e-4, atol=0)          # north star: 1e-4 relative
    assert np.allclose(loglk, d["loglk"], rtol=0, atol=6e-5 if dt == np.float64 else 2e-2)   # observed: the printed digits
    assert np.array_equal(ratecat, d["ratecat"])
    assert np.allclose(rates, d["rates"], rtol=0, atol=1e-6 * max(1.0, d["rates"].max()))
    ref = bytes(d["newick"]).decode().strip()
    assert strip(tree) == strip(ref)
    dl = np.abs(lengths(tree) - lengths(ref))
    print(name, "printed lengths differing:", int((dl > 0).sum()), "of", len(dl), "max", dl.max())
    assert tree == ref            # ordered totals + glibc's exp / log: bit-identical likelihoods -> byte-identical tree
    boot = nj_newick(_make(dt), d["codes"], names, dtype=dt, me_lengths=True, mllen=ncat, aa_model=model, n_bootstrap=1000)
    refb = bytes(d["newick_support"]).decode().strip()
    assert no_support(boot) == no_support(refb)
    ds = np.abs(supports(boot) - supports(refb))
    print(name, "supports differing:", int((ds > 0).sum()), "of", len(ds), "max", ds.max() if len(ds) else 0)
    assert boot == refb


@pytest.mark.parametrize("name", ["full_aa_120_lg_double", "full_aa_150_lg", "full_aa_100_jtt", "full_aa_90_wag_double"])
def test_full_protein_pipeline_matches_the_reference_run(name):
    """`VeryFastTree [-lg | -wag] [-double-precision]` on proteins - the complete default pipeline (NJ, ME NNIs + SPRs, ML
    NNIs, CAT, SH-like supports); full_aa_120_lg_double carries BASELINE config C5's exact flags.  TreeLogLk of every
    round within 1e-4 relative (north star; observed: every printed digit) and the output byte for byte, supports included."""
    from veryfasttree_amd.backend import nj_newick
    d = G.load(name)
    model, dt, ncat = _model(d)
    names = ["s%d" % k for k in range(len(d["codes"]))]
    kw = dict(dtype=dt, me_lengths=True, me_nni=True, spr=2, ml_nni=ncat, aa_model=model)
    tree, loglk = nj_newick(_make(dt), d["codes"], names, return_loglk=True, **kw)
    want = d["loglk"]
    print(name, "rounds", len(loglk) - 1, "vs", len(want) - 1, "loglk", loglk, "vs", want)
    assert len(loglk) == len(want)
    assert np.allclose(loglk, want, rtol=1e-4, atol=0)
    ref = bytes(d["newick"]).decode().strip()
    assert strip(tree) == strip(ref), "topology differs"
    dl = np.abs(lengths(tree) - lengths(ref))
    print(name, "printed lengths differing:", int((dl > 0).sum()), "of", len(dl), "max", dl.max())
    assert tree == ref
    boot = nj_newick(_make(dt), d["codes"], names, n_bootstrap=1000, **kw)
    refb = bytes(d["newick_support"]).decode().strip()
    assert no_support(boot) == no_support(refb)
    ds = np.abs(supports(boot) - supports(refb))
    print(name, "supports differing:", int((ds > 0).sum()), "of", len(ds), "max", ds.max() if len(ds) else 0)
    assert boot == refb
