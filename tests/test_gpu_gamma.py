"""`-gamma` (vft_nj_options.gamma; MLLengths::branchlengthScale after NJ.tcc:297-308, :5261-5357): the complete default pipeline at one
thread, then the fit of a discretised Gamma over the 20 rate categories and the rescaled branch lengths - against whole runs of the
reference (oracle/gen_fixtures.py gamma): the tree with its supports byte for byte, the numbers of the "Gamma(20) LogLk" line."""
import numpy as np
import pytest

import golden_util as G

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["gamma_nt_200", "gamma_nt_300_gtr", "gamma_aa_150_lg_double"])
def test_gamma_rescaled_tree_matches_the_reference(name):
    from veryfasttree_amd import HipProfileOps
    from veryfasttree_amd.backend import nj_newick, last_gamma
    d = G.load(name)
    flags = bytes(d["flags"]).decode().split()
    codes_all = d["codes"]
    nt = "-nt" in flags
    dt = np.float64 if "-double-precision" in flags else np.float32
    names = ["s%d" % k for k in range(len(codes_all))]
    kw = dict(dtype=dt, me_lengths=True, me_nni=True, spr=2, ml_nni=20, n_bootstrap=1000, gtr="-gtr" in flags, gamma=True, return_loglk=True)
    if not nt:
        kw["aa_model"] = "lg"
    tree, loglk = nj_newick(lambda n, L: HipProfileOps(n, L, 4 if nt else 20, dt, max_nodes=3 * n), codes_all, names, **kw)
    got, want = last_gamma(), d["gamma"]
    print(name, "Gamma(20) LogLk %.3f alpha %.3f rescale %.3f; reference" % got, list(want))
    assert np.allclose(loglk, d["loglk"], rtol=1e-4, atol=0)
    # the reference prints three decimals; the log-likelihood within the north star's 1e-4 relative as well
    assert abs(got[0] - want[0]) <= max(1e-4 * abs(want[0]), 6e-4)
    assert abs(got[1] - want[1]) < 6e-4 and abs(got[2] - want[2]) < 6e-4
    assert tree == bytes(d["newick"]).decode().strip()
