#!/usr/bin/env python3
"""How many candidates the top-k selection ranks per seed on bench.py's workload (the histogram rounds: k_select_hist / collect / rank) and
what one batch of 8 seeds costs.  (Round 6 also tried a histogram-free front - a bound from the sweep's per-wavefront minimum criteria -
behind a debug option that is gone with it: 3 500-5 700 candidates per seed instead of ~2 030)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veryfasttree_amd import HipProfileOps, synth
from veryfasttree_amd.workload import TopHitsState
n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000, 200
codes = synth.random_descent_codes(n, L, 4, 0.02, 0.01, seed=4)
ops = HipProfileOps(n, L, 4, np.float32)
st = TopHitsState(ops, codes, n // 4)
k = 2 * int(0.5 + np.sqrt(n))
leaf_act, int_act = st.active[st.active < n], st.active[st.active >= n]
seeds = []
for s in range(4):
    seeds.append(int(leaf_act[(s * 7919 + 13) % len(leaf_act)]))
    seeds.append(int(int_act[(s * 104729 + 7) % len(int_act)]))
seeds = np.asarray(seeds, np.int64)
for _ in range(3):
    ops.setBestHitBatch(seeds, st.n_active, st.n_diff_allow, st.totdiam, k, view=True)
ops.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    ops.setBestHitBatch(seeds, st.n_active, st.n_diff_allow, st.totdiam, k, view=True)
ops.synchronize()
dt = (time.perf_counter() - t0) / 20
print("%.1f us per batch of 8 seeds; candidates per seed (k = %d): %s" % (dt * 1e6, k, [ops.sweep_batch_info(s)[0] for s in range(len(seeds))]))
