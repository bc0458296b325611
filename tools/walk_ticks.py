#!/usr/bin/env python3
"""Where k_walk_step_args spends its time (a variant library built with VFT_EXTRA_HIPCC_FLAGS=-DVFT_WALK_TIMING, loaded through
VFT_LIB_DIR): clock ticks (100 MHz) of thread 0 of every workgroup, summed over a complete pipeline.  usage: walk_ticks.py N L [aa]"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veryfasttree_amd import HipProfileOps, synth, backend
from veryfasttree_amd.backend import nj_newick, last_stage_seconds
n, L = int(sys.argv[1]), int(sys.argv[2])
aa = "aa" in sys.argv[3:]
dt = np.float64 if aa else np.float32
codes = synth.random_descent_codes(n, L, 20 if aa else 4, 0.02, 0.01, seed=4)
names = ["s%d" % k for k in range(n)]
kw = dict(aa_model="lg") if aa else {}
t0 = time.perf_counter()
nj_newick(lambda m, Lp: HipProfileOps(m, Lp, 20 if aa else 4, dt, max_nodes=3 * m), codes, names, dtype=dt, me_lengths=True, me_nni=True, spr=2, **kw)
st = last_stage_seconds()
t = (ctypes.c_ulonglong * 8)()
lib = backend.load_library()
have = hasattr(lib, "vft_walk_ticks") and lib.vft_walk_ticks(t) == 0
wg = max(t[3], 1)
print("%d x %d: %.1f s, SPR %.2f s for %d steps (%.1f us per step)" % (n, L, time.perf_counter() - t0, st["of_which_spr"], st["spr_steps"], 1e6 * st["of_which_spr"] / max(st["spr_steps"], 1)))
if not have:
    sys.exit(0)   # (not a timing build: the wall-clock line only)
print("k_walk_step_args: %d launches, %.2f averages per launch; thread 0 of a workgroup, us per launch:" % (wg // 6, t[4] / wg))
for k, nm in enumerate(["averages (chain)", "pair columns + ordered sum", "publication"]):
    print("  %-28s %6.2f" % (nm, t[k] / 100.0 / wg))
