#!/usr/bin/env python3
"""pairLogLk / posteriorProfile throughput by batch size on the 1M x 200 nt state (JC), whole C-ABI call."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veryfasttree_amd import HipProfileOps, synth
from veryfasttree_amd.workload import TopHitsState
n, L = 1000000, 200
codes = synth.random_descent_codes(n, L, 4, 0.02, 0.01, seed=4)
ops = HipProfileOps(n, L, 4, np.float32)
st = TopHitsState(ops, codes, n // 4)
rates = np.exp(-np.log(20.0) + np.arange(20) * 2 * np.log(20.0) / 19)
ops.set_rates(rates, np.random.default_rng(1).integers(0, 20, L))
nj = n // 4
lv1 = n + np.arange(nj // 2, dtype=np.int64)
ops.posteriorProfile(lv1, 2 * np.arange(nj // 2, dtype=np.int64), 2 * np.arange(nj // 2, dtype=np.int64) + 1,
                     np.full(nj // 2, 0.05), np.full(nj // 2, 0.07))
S, V = 4, 16
phi = float(ops.profile_nvectors(n, nj // 2).mean()) / L
side = L * (S + 1) + phi * L * V
print("phi = %.3f" % phi)
for b in (1, 64, 1024, 8192, 62500):
    a, bb = lv1[:b], lv1[b:2 * b] if 2 * b <= len(lv1) else lv1[:b][::-1]
    ln = np.full(b, 0.1)
    out = n + nj // 2 + np.arange(b, dtype=np.int64)
    for name, fn, byt in (("pairLogLk", lambda: ops.pairLogLk(a, bb, ln), b * (2 * side + L + 8)),
                          ("posteriorProfile", lambda: ops.posteriorProfile(out, a, bb, ln, ln), b * (3 * side + L))):
        fn(); ops.synchronize()
        t0 = time.perf_counter()
        reps = 5 if b < 10000 else 2
        for _ in range(reps):
            fn()
        ops.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("%-18s batch %6d  %9.1f us/call  %10.3e ops/s  %7.0f GB/s (alg)" % (name, b, dt * 1e6, b / dt, byt / dt / 1e9))
