#!/usr/bin/env python3
"""Diagnostics for bench.py's first step under `rocprofv3 --pmc`: the batch of 8 seeds with a lazy refresh due, then the criteria."""
import os, sys
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
for rep in range(3):
    try:
        h, b = ops.setBestHitBatch(seeds, st.n_active, st.n_diff_allow, st.totdiam, k)
        print("step", rep, "ok; candidates", [ops.sweep_batch_info(s) for s in range(8)])
    except Exception as e:
        print("step", rep, "FAILED", e, [ops.sweep_batch_info(s) for s in range(8)])
    d, w, c = ops.sweep_results(0, st.maxnode)
    act = st.parent < 0
    ca = c[act]
    od, na = ops.get_out_distances(0, st.maxnode)
    print("   slot 0 crit active: min %g max %g nan %d inf %d distinct %d; inactive all 1e20 %s; outdist nan %d min %g max %g; stamps min %d max %d" % (
        np.nanmin(ca), np.nanmax(ca), np.isnan(ca).sum(), np.isinf(ca).sum(), len(np.unique(ca)), bool(np.all(c[~act] == np.float32(1e20))),
        int(np.isnan(od[act]).sum()), np.nanmin(od[act]), np.nanmax(od[act]), na[act].min(), na[act].max()))
