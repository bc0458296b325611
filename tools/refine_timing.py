#!/usr/bin/env python3
"""Wall-clock of the refinement stages on a synthetic alignment: refine_timing.py N L [threads] -> NJ+ME lengths, + ME NNIs,
+ ME NNIs + 2 SPR rounds, + ML NNIs (20 rate categories), each as a whole nj_newick call; threads > 1: the subtree schedule."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veryfasttree_amd import HipProfileOps, synth
from veryfasttree_amd.backend import nj_newick
n, L = int(sys.argv[1]), int(sys.argv[2])
T = int(sys.argv[3]) if len(sys.argv) > 3 else 1
codes = synth.random_descent_codes(n, L, 4, 0.03, 0.01, seed=2)
names = ["s%d" % k for k in range(n)]
make = lambda m, Lp: HipProfileOps(m, Lp, 4, np.float32, max_nodes=3 * m)
nj_newick(make, codes[:64], names[:64], me_lengths=True)
for label, kw in (("NJ + ME lengths", dict()), ("+ ME NNIs", dict(me_nni=True)), ("+ ME NNIs + 2 SPR rounds", dict(me_nni=True, spr=2)),
                  ("NJ + ML NNIs (CAT 20)", dict(ml_nni=20)), ("everything + SH supports", dict(me_nni=True, spr=2, ml_nni=20, n_bootstrap=1000))):
    t0 = time.perf_counter()
    nj_newick(make, codes, names, me_lengths=True, threads=T, **kw)
    print("%-28s %8.2f s   (threads = %d)" % (label, time.perf_counter() - t0, T), flush=True)
