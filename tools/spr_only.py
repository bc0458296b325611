import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from veryfasttree_amd import HipProfileOps, synth
from veryfasttree_amd.backend import nj_newick
n, L = 1000, 500
codes = synth.random_descent_codes(n, L, 4, 0.03, 0.01, seed=2)
names = ["s%d" % k for k in range(n)]
make = lambda m, Lp: HipProfileOps(m, Lp, 4, np.float32, max_nodes=3 * m)
t0 = time.perf_counter()
nj_newick(make, codes, names, me_lengths=True, me_nni=True, spr=2)
print("ME NNIs + SPR: %.2f s" % (time.perf_counter() - t0))
