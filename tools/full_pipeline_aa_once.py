#!/usr/bin/env python3
"""One run of the complete default protein pipeline (-lg -double-precision) on a synthetic alignment, for rocprofv3:
full_pipeline_aa_once.py N L"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veryfasttree_amd import HipProfileOps, synth
from veryfasttree_amd.backend import nj_newick
n, L = int(sys.argv[1]), int(sys.argv[2])
codes = synth.random_descent_codes(n, L, 20, 0.03, 0.01, seed=2)
names = ["s%d" % k for k in range(n)]
t0 = time.perf_counter()
tree, ll = nj_newick(lambda m, Lp: HipProfileOps(m, Lp, 20, np.float64, max_nodes=3 * m), codes, names, dtype=np.float64,
                     me_lengths=True, me_nni=True, spr=2, ml_nni=20, n_bootstrap=1000, aa_model="lg", return_loglk=True)
print("whole pipeline: %.2f s, final TreeLogLk %.4f" % (time.perf_counter() - t0, ll[-1]))
