#!/usr/bin/env python3
"""One sweep configuration repeated, for PMC runs: microbench_one.py <leaf|internal|all> <leaf|internal> [reps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veryfasttree_amd import HipProfileOps, synth
from veryfasttree_amd.workload import TopHitsState
targets, seed = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
n, L = 1000000, 200
codes = synth.random_descent_codes(n, L, 4, 0.02, 0.01, seed=4)
ops = HipProfileOps(n, L, 4, np.float32)
st = TopHitsState(ops, codes, n // 4)
n64 = (n // 64) * 64
lo, hi = {"leaf": (0, n64), "internal": (n64, st.maxnode), "all": (0, st.maxnode)}[targets]
ops.set_shard(lo, hi)
q = int(st.active[st.active < n][17]) if seed == "leaf" else int(st.active[st.active >= n][23])
for _ in range(reps):
    ops.setBestHit(q, st.n_active, st.n_diff_allow, st.totdiam, 0, want_best=False, want_hits=False)
ops.synchronize()
