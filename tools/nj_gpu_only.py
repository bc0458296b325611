#!/usr/bin/env python3
"""NJ phase on the GPU only (no reference run): nj_gpu_only.py N L [fastest] [second] [mu=0.03] [gap=0.01] [seed=3]
BASELINE config C4: nj_gpu_only.py 1000000 200 mu=0.02 gap=0.01 seed=4   (default -nt: top hits without -fastest)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veryfasttree_amd import HipProfileOps, synth
from veryfasttree_amd.backend import nj_run
n, L = int(sys.argv[1]), int(sys.argv[2])
fastest = "fastest" in sys.argv[3:]
second = "second" in sys.argv[3:]
kv = dict(a.split("=") for a in sys.argv[3:] if "=" in a)
codes = synth.random_descent_codes(n, L, 4, float(kv.get("mu", 0.03)), float(kv.get("gap", 0.01)), seed=int(kv.get("seed", 3)))
_, first = np.unique(codes, axis=0, return_index=True)
codes = codes[np.sort(first)]
ops = HipProfileOps(codes.shape[0], L, 4, np.float32)
t0 = time.perf_counter()
joins, crit = nj_run(ops, codes, fastest=fastest, second_level=second)
import zlib
if "save" in kv:   # the join order as int32 [n, 3] (i, j, new node), e.g. save=gpurun_out/joins.npy
    np.save(kv["save"], np.ascontiguousarray(joins, np.int64).astype(np.int32))
print("GPU NJ phase: %.2f s for %d joins (%d unique seqs, L=%d, fastest=%s, 2nd=%s)  join-order crc %d" % (
    time.perf_counter() - t0, len(joins), codes.shape[0], L, fastest, second, zlib.crc32(np.ascontiguousarray(joins, np.int64).astype("<i4").tobytes())))
