#!/usr/bin/env python3
"""Sweep timing on the amino-acid configuration C5 (50k x 300, BLOSUM45-style distance matrix, double or float):
k_sweep_generic, one lane per (seed, target) pair."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veryfasttree_amd import HipProfileOps, synth
from veryfasttree_amd.workload import TopHitsState

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dt = np.float64 if (len(sys.argv) <= 3 or sys.argv[3] == "f64") else np.float32
d = np.load(os.path.join(ROOT, "tests", "golden", "wb_aa_f64.npz"))
codes = synth.random_descent_codes(n, L, 20, 0.08, 0.02, seed=5)
ops = HipProfileOps(n, L, 20, dt)
ops.set_distance_matrix(d["dmat.distances"], d["dmat.codefreq"], d["dmat.eigenval"], d["dmat.eigentot"])
t0 = time.perf_counter()
st = TopHitsState(ops, codes, n // 4)
print("state: %d active (%d internal) in %.1f s" % (st.n_active, n // 4, time.perf_counter() - t0))
S = np.dtype(dt).itemsize
nvec = ops.profile_nvectors(n, n // 4)
alg = (n // 2) * (L + 3 * S + 8) + (n // 4) * (L * (S + 1) + 3 * S + 8) + int(nvec.sum()) * 20 * S
for qn, q in (("leaf seed", int(st.active[st.active < n][17])), ("internal seed", int(st.active[st.active >= n][23]))):
    for _ in range(2):
        ops.setBestHit(q, st.n_active, st.n_diff_allow, st.totdiam, 0, want_best=False, want_hits=False)
    ops.synchronize()
    ops.timer_start()
    for _ in range(5):
        ops.setBestHit(q, st.n_active, st.n_diff_allow, st.totdiam, 0, want_best=False, want_hits=False)
    tot = ops.timer_stop_ms()
    ms, nl = ops.sweep_kernel_ms()
    print("%-14s sweep kernel %.1f us (x%d)  whole call %.1f us   algorithmic %.1f MB -> %.0f GB/s" % (
        qn, ms * 1e3, nl, tot * 200, alg / 1e6, alg / (ms * 1e-3) / 1e9))
