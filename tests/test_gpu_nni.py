"""Nearest-neighbour interchanges on the GPU backend: DoNNI (NJ.tcc:5797-6200) driven by the C++ host
(veryfasttree_amd/host/MLLengths.h::doNNI) against whole reference runs."""
import re

import numpy as np
import pytest

import golden_util as G

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,dt", [("nni_nt_12", np.float32), ("nni_nt_200", np.float32), ("nni_nt_500", np.float32),
                                     ("nni_nt_300_double", np.float64)])
def test_min_evolution_nnis_match_the_reference_run(name, dt):
    """`VeryFastTree -nt -noml -spr 0 [-nosupport]`: fastNJ, rounds of minimum-evolution NNIs, ME branch lengths,
    local-bootstrap supports - the final trees byte for byte (topology after 5 ... 345 NNIs, lengths, supports)."""
    from veryfasttree_amd import HipProfileOps
    from veryfasttree_amd.backend import nj_newick
    d = G.load(name)
    codes_all = d["codes"]
    names = ["s%d" % k for k in range(len(codes_all))]
    make = lambda n, L: HipProfileOps(n, L, 4, dt, max_nodes=3 * n)
    tree = nj_newick(make, codes_all, names, dtype=dt, me_lengths=True, me_nni=True)
    ref = bytes(d["newick"]).decode().strip()
    strip = lambda t: re.sub(r":[0-9.eE+-]+", ":", t)
    assert strip(tree) == strip(ref), "topology differs"
    assert tree == ref
    boot = nj_newick(make, codes_all, names, dtype=dt, me_lengths=True, me_nni=True, n_bootstrap=1000)
    assert boot == bytes(d["newick_support"]).decode().strip()
