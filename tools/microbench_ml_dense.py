#!/usr/bin/env python3
"""pairLogLk / posteriorProfile on dense ML rows (the ML phase's own layout) by batch size: N x L nt, JC, 20 rate
categories.  Level-1 posteriors are built from leaves, level-2 from level-1; the timed calls read level-2 rows (dense,
phi ~ 0.5) and - for the posterior - write level-3 rows.  posterior = stream-ordered call + synchronize."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veryfasttree_amd import HipProfileOps, synth
n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 262144, int(sys.argv[2]) if len(sys.argv) > 2 else 200
codes = synth.random_descent_codes(n, L, 4, 0.05, 0.01, seed=4)
ops = HipProfileOps(n, L, 4, np.float32, max_nodes=2 * n)
ops.upload_leaves(codes)
ops.set_max_node(2 * n)
rates = np.exp(-np.log(20.0) + np.arange(20) * 2 * np.log(20.0) / 19)
ops.set_rates(rates, np.random.default_rng(1).integers(0, 20, L))
ops.set_ml_limits(5e-4, 2.5e-4, 1e-10)
ops.branch_lengths_set(0, np.full(2 * n, 0.05, np.float32))
h = n // 2
lv1 = n + np.arange(h, dtype=np.int64)
ops.posteriorProfileBlen(lv1, 2 * np.arange(h), 2 * np.arange(h) + 1, 2 * np.arange(h), 2 * np.arange(h) + 1)
q = h // 2
lv2 = n + h + np.arange(q, dtype=np.int64)
ops.posteriorProfileBlen(lv2, lv1[0::2], lv1[1::2], lv1[0::2], lv1[1::2])
ops.synchronize()
S, V = 4, 16
W, Cc, F = ops.profile_download(int(lv2[5]))
phi = float(np.mean([((ops.profile_download(int(v))[1] == 127) & (ops.profile_download(int(v))[0] > 0)).mean() for v in lv2[:32]]))
side = L * (S + 1) + phi * L * V
print("# %d x %d nt float32, level-2 posterior rows: phi = %.3f, %d B per profile (SURVEY 8d)" % (n, L, phi, side))
for b in (1, 1024, 8192, 65536):
    if 2 * b > len(lv2):
        break
    a, bb = lv2[:b], lv2[b:2 * b]
    ln = np.full(b, 0.1)
    out = n + h + q + np.arange(b, dtype=np.int64)
    def post():
        ops.posteriorProfileBlen(out, a, bb, a, bb)
        ops.synchronize()
    for name, fn, byt in (("pairLogLk", lambda: ops.pairLogLk(a, bb, ln), b * (2 * side + L + 8)),
                          ("posteriorProfile", post, b * (3 * side + L))):
        fn()
        t0 = time.perf_counter()
        reps = 5 if b < 10000 else 3
        for _ in range(reps):
            fn()
        ops.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("%-18s batch %6d  %9.1f us/call  %10.3e ops/s  %7.0f GB/s (alg)" % (name, b, dt * 1e6, b / dt, byt / dt / 1e9))
