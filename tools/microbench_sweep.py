#!/usr/bin/env python3
"""Times the sweep kernel on sub-ranges (leaf targets only / internal targets only) for leaf and internal seeds."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veryfasttree_amd import HipProfileOps, synth
from veryfasttree_amd.workload import TopHitsState

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 200
codes = synth.random_descent_codes(n, L, 4, 0.02, 0.01, seed=4)
ops = HipProfileOps(n, L, 4, np.float32)
st = TopHitsState(ops, codes, n // 4)
leaf_q = int(st.active[st.active < n][17])
int_q = int(st.active[st.active >= n][23])
n64 = (n // 64) * 64
for name, lo, hi in (("leaf targets", 0, n64), ("internal targets", n64, st.maxnode), ("all", 0, st.maxnode)):
    ops.set_shard(lo, hi)
    for qn, q in (("leaf seed", leaf_q), ("internal seed", int_q)):
        for _ in range(2):
            ops.setBestHit(q, st.n_active, st.n_diff_allow, st.totdiam, 0, want_best=False, want_hits=False)
        ops.synchronize()
        ops.timer_start()
        for _ in range(10):
            ops.setBestHit(q, st.n_active, st.n_diff_allow, st.totdiam, 0, want_best=False, want_hits=False)
        tot = ops.timer_stop_ms()
        ms, nl = ops.sweep_kernel_ms()
        ms2, _ = ops.sweep_table_kernel_ms()
        print("%-18s %-14s k_sweep_nt %.1f us + k_sweep_nt_table %.1f us (x%d)   whole call %.1f us" % (
            name, qn, ms * 1e3, ms2 * 1e3, nl, tot * 100))
ops.set_shard(0, st.maxnode)
for k in (0, 200, 2000):
    t0 = time.perf_counter()
    for _ in range(20):
        ops.setBestHit(int_q, st.n_active, st.n_diff_allow, st.totdiam, k, want_best=False, want_hits=k > 0)
    ops.synchronize()
    print("k=%d: %.1f us per sweep+select (host wall)  select info (nCand, extra rounds) = %s" % (
        k, (time.perf_counter() - t0) / 20 * 1e6, ops.sweep_info() if k else None))
