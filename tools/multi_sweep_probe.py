#!/usr/bin/env python3
"""Launch time of the multi-seed sweeps (k_sweep_nt_leafq_multi, k_sweep_nt_profq_multi) on bench.py's workload, against one launch per seed (VFT_DEBUG_NO_MULTI_SWEEP).  usage: multi_sweep_probe.py [N L joins]"""
import ctypes, os, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veryfasttree_amd import HipProfileOps, synth
from veryfasttree_amd.workload import TopHitsState
n, L, nj = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1000000, 200, 250000)
codes = synth.random_descent_codes(n, L, 4, 0.02, 0.01, seed=4)
ops = HipProfileOps(n, L, 4, np.float32)
st = TopHitsState(ops, codes, nj)
leaf_act = st.active[st.active < n]
int_act = st.active[st.active >= n]
for kind, nseeds in (("leaf", 4), ("leaf", 2), ("profile", 4), ("profile", 2)):
    act = leaf_act if kind == "leaf" else int_act
    seeds = np.asarray([int(act[(s * 7919 + 13) % len(act)]) for s in range(nseeds)], np.int64)
    for name, opt in [("one launch per seed", (12, 1)), ("seeds sharing passes", (12, 0))]:
        assert ops.lib.vft_debug_option(ops.ctx, ctypes.c_int32(opt[0]), ctypes.c_int64(opt[1])) == 0
        for _ in range(3):
            hits, _ = ops.setBestHitBatch(seeds, st.n_active, st.n_diff_allow, st.totdiam, 2000)
        ops.timer_start()
        for _ in range(10):
            hits, _ = ops.setBestHitBatch(seeds, st.n_active, st.n_diff_allow, st.totdiam, 2000)
        ops.timer_stop_ms()
        ms, nl = ops.sweep_kernel_ms()
        print("%d %s seeds, %-20s %7.1f us per launch, %d launches, %5.1f us per sweep, crc %08x" % (nseeds, kind, name, 1e3 * ms, nl, 1e3 * ms * nl / ops.sweep_kernel_sweeps(), zlib.crc32(hits.tobytes())))
