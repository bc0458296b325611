#!/usr/bin/env python3
"""Where k_ml_quartet spends its time (a library whose vft_ml_kernels_quartet64.hip was built with -DVFT_ML_TIMING, installed
as veryfasttree_amd/lib/libvft_hip.so): clock ticks (100 MHz) of thread 0 of every workgroup, summed over the complete
protein pipeline (-lg -double-precision).  usage: ml_ticks.py N L"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veryfasttree_amd import HipProfileOps, synth, backend
from veryfasttree_amd.backend import nj_newick
n, L = int(sys.argv[1]), int(sys.argv[2])
codes = synth.random_descent_codes(n, L, 20, 0.03, 0.01, seed=2)
names = ["s%d" % k for k in range(n)]
t0 = time.perf_counter()
tree, ll = nj_newick(lambda m, Lp: HipProfileOps(m, Lp, 20, np.float64, max_nodes=3 * m), codes, names, dtype=np.float64,
                     me_lengths=True, me_nni=True, spr=2, ml_nni=20, n_bootstrap=1000, aa_model="lg", return_loglk=True)
print("whole pipeline: %.2f s, final TreeLogLk %.4f" % (time.perf_counter() - t0, ll[-1]))
t = (ctypes.c_ulonglong * 16)()
rc = backend.load_library().vft_ml_ticks(t)
assert rc == 0, rc
wgs, evals = max(t[9], 1), max(t[8], 1)
names = ["other (Brent, totals, stores)", "P(t) tables", "column likelihoods", "ordered total", "step set-up (tables, loads, posteriors)"]
tot = sum(t[:5])
print("workgroups %d, evaluations %d (%.1f per workgroup), %.1f us per workgroup" % (t[9], t[8], t[8] / wgs, tot / 100.0 / wgs))
for k, nm in enumerate(names):
    print("%-44s %8.2f us per workgroup  %5.1f %%   %6.2f us per evaluation" % (nm, t[k] / 100.0 / wgs, 100.0 * t[k] / max(tot, 1), t[k] / 100.0 / evals))
print("ordered totals that fell back to the plain chain: %d of %d; rescaling events per evaluation: %.1f" % (t[7], t[8], t[6] / evals))
print("inside the ordered total (us per evaluation): " + "  ".join("%s %.2f" % (nm, t[10 + k] / 100.0 / evals) for k, nm in enumerate(
    ["entry barrier", "log sums + scan", "intervals + clamp scan", "events + scan", "lists", "chain"])))
