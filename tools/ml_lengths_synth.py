#!/usr/bin/env python3
"""`-nome -mllen -nocat -nosupport` on a synthetic alignment: ml_lengths_synth.py N L [mu gap seed] -> wall-clock of the
NJ + ME-lengths pipeline without and with the ML branch-length stage, and the TreeLogLk of every round.
(BASELINE config C2 = 10000 1000 0.03 0.01 2.)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veryfasttree_amd import HipProfileOps, synth
from veryfasttree_amd.backend import nj_newick
n, L = int(sys.argv[1]), int(sys.argv[2])
mu = float(sys.argv[3]) if len(sys.argv) > 3 else 0.03
gap = float(sys.argv[4]) if len(sys.argv) > 4 else 0.01
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 2
full = os.environ.get("VFT_FULL")   # "1": the complete default pipeline (ME NNIs + 2 SPR rounds, ML NNIs, CAT, SH supports)
nboot = int(os.environ.get("VFT_ML_BOOT", "0"))   # 1000 = the reference's default SH-like supports
ncat = int(os.environ.get("VFT_ML_CAT", "1"))   # 1 = -nocat, 20 = the reference's default CAT approximation
codes = synth.random_descent_codes(n, L, 4, mu, gap, seed=seed)
names = ["s%d" % k for k in range(n)]
DT = np.float64 if os.environ.get('VFT_DOUBLE') else np.float32
make = lambda m, Lp: HipProfileOps(m, Lp, 4, DT, max_nodes=3 * m)
t0 = time.perf_counter()
nj_newick(make, codes, names, me_lengths=True, dtype=DT)
t_nj = time.perf_counter() - t0
t0 = time.perf_counter()
tree, loglk = nj_newick(make, codes, names, me_lengths=True, return_loglk=True, dtype=DT, **(dict(me_nni=True, spr=2, ml_nni=20, n_bootstrap=1000, gtr=bool(os.environ.get('VFT_GTR'))) if full else dict(mllen=ncat, n_bootstrap=nboot)))
t_ml = time.perf_counter() - t0
print("NJ + ME lengths                 %8.2f s" % t_nj)
print("NJ + ME lengths + ML lengths (%d rate categories, %d resamples)   %8.2f s   -> ML stage %.2f s, %d rounds (%.2f s per round incl. treeLogLk)"
      % (ncat, nboot, t_ml, t_ml - t_nj, len(loglk), (t_ml - t_nj) / max(len(loglk), 1)))
print("TreeLogLk per round: " + " ".join("%.4f" % x for x in loglk))
if len(sys.argv) > 6:
    open(sys.argv[6], "w").write(tree + "\n")
