#!/usr/bin/env python3
"""Where k_nj_glue_scan spends its time (a library built with VFT_EXTRA_HIPCC_FLAGS=-DVFT_NJ_TIMING, installed as
veryfasttree_amd/lib/libvft_hip.so - the host driver links against that path): clock ticks (100 MHz) between the kernel's phases, summed over a run.  usage: engine_ticks.py N L"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veryfasttree_amd import HipProfileOps, synth, backend
from veryfasttree_amd.backend import nj_run
n, L = int(sys.argv[1]), int(sys.argv[2])
codes = synth.random_descent_codes(n, L, 4, 0.02, 0.01, seed=4)
codes = codes[np.sort(np.unique(codes, axis=0, return_index=True)[1])]
ops = HipProfileOps(codes.shape[0], L, 4, np.float32)
joins, _ = nj_run(ops, codes)
t = (ctypes.c_ulonglong * 16)()
rc = backend.load_library().vft_nj_engine_ticks(ops.ctx, t)
assert rc == 0, rc
names = ["tv + slot cache", "staging -> keys", "sort", "decision + save", "updateTopVisible(new)", "updateVisible tests", "updateVisible updates",
         "write back", "scan", "setOutDistance(cur)"]
tot = sum(t[:10])
for k, nm in enumerate(names):
    print("%-24s %8.2f us per join  %5.1f %%" % (nm, t[k] / 100.0 / len(joins), 100.0 * t[k] / max(tot, 1)))
print("sum %.2f us per join" % (tot / 100.0 / len(joins)))
