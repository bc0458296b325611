#!/usr/bin/env python3
"""Debugging aid: the join engine against the host-driven loop on one alignment; prints the first differing join.
usage: engine_diff.py N L [fastest]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veryfasttree_amd import HipProfileOps, synth
from veryfasttree_amd.backend import nj_run
n, L = int(sys.argv[1]), int(sys.argv[2])
fastest = "fastest" in sys.argv[3:]
kv = dict(a.split("=") for a in sys.argv[3:] if "=" in a)
codes = synth.random_descent_codes(n, L, 4, 0.04, 0.02, seed=int(kv.get("seed", 77)))
codes = codes[np.sort(np.unique(codes, axis=0, return_index=True)[1])]
runs = []
for host in (False, True):
    ops = HipProfileOps(codes.shape[0], codes.shape[1], 4, np.float32)
    runs.append(nj_run(ops, codes, fastest=fastest, second_level=False, debug_flags=1 if host else 0))
    ops.close()
(j0, c0), (j1, c1) = runs
bad = np.nonzero((j0 != j1).any(axis=1) | (c0 != c1))[0]
m = int(0.5 + np.sqrt(codes.shape[0]))
print("n %d m %d joins %d; differing joins: %d" % (codes.shape[0], m, len(j0), len(bad)))
if len(bad):
    k = int(bad[0])
    print("first difference at join %d (nActive %d)" % (k, codes.shape[0] - k))
    for t in range(max(0, k - 2), min(len(j0), k + 3)):
        print("  join %d: engine %s %.9g   host %s %.9g" % (t, j0[t], c0[t], j1[t], c1[t]))
