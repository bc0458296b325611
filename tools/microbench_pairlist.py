#!/usr/bin/env python3
"""Latency of one pair-list call (vft_pair_distances: a hub node against n partners, as after a join) on the 1M x 200
benchmark state, by list length.  usage: microbench_pairlist.py [threads per pair: 64|128|256]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veryfasttree_amd import HipProfileOps, synth
from veryfasttree_amd.workload import TopHitsState
n, L = 1000000, 200
codes = synth.random_descent_codes(n, L, 4, 0.02, 0.01, seed=4)
ops = HipProfileOps(n, L, 4, np.float32)
pair_threads = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ops.debug_option(2, pair_threads)   # VFT_DEBUG_PAIR_THREADS (0 = the library's own choice)
st = TopHitsState(ops, codes, n // 4)
rng = np.random.default_rng(1)
act = np.array(st.active)
internal = act[act >= n]
hub = int(internal[-1])
for cnt in (1, 64, 512, 1024, 2000):
    for mix in ("internal", "leaves"):
        pool = internal[:-1] if mix == "internal" else act[act < n]
        pj = rng.choice(pool, cnt, replace=False).astype(np.int64)
        pi = np.full(cnt, hub, np.int64)
        for _ in range(5):
            ops.setDistCriterion(pi, pj, st.n_active, st.n_diff_allow, st.totdiam)
        t0 = time.perf_counter()
        reps = 200
        for _ in range(reps):
            ops.setDistCriterion(pi, pj, st.n_active, st.n_diff_allow, st.totdiam)
        print("%5d pairs, hub vs %-8s  %7.1f us / call (threads per pair: %s)" % (cnt, mix, (time.perf_counter() - t0) / reps * 1e6, pair_threads or "default"))
        if cnt >= 512 and mix == "internal":   # the hub needs a refresh in every call (as right after a join): the single-launch variant
            od, na = ops.get_out_distances(hub, 1)
            t0 = time.perf_counter()
            tset = 0.0
            for _ in range(reps):
                t1 = time.perf_counter()
                ops.set_out_distances(hub, od, np.array([10 * n], np.int64))
                tset += time.perf_counter() - t1
                ops.setDistCriterion(pi, pj, st.n_active, st.n_diff_allow, st.totdiam)
            print("%5d pairs, stale hub          %7.1f us / call" % (cnt, (time.perf_counter() - t0 - tset) / reps * 1e6))
