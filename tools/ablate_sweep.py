#!/usr/bin/env python3
"""Sweep-kernel ablations on the 1M benchmark state (VFT_SWEEP_ABLATE bits: 1 no vector loads, 2 no weight loads,
4 no arithmetic, 8 plain workgroup order).  Results of ablated runs are wrong by design; timing only.
The switches only exist in a library built with VFT_EXTRA_HIPCC_FLAGS=-DVFT_ABLATE (veryfasttree_amd/build.py);
point VFT_HIP_LIB at that build - the product library ignores VFT_SWEEP_ABLATE."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veryfasttree_amd import HipProfileOps, synth
from veryfasttree_amd.workload import TopHitsState
n, L = 1000000, 200
codes = synth.random_descent_codes(n, L, 4, 0.02, 0.01, seed=4)
ops = HipProfileOps(n, L, 4, np.float32)
st = TopHitsState(ops, codes, n // 4)
n64 = (n // 64) * 64
q_leaf = int(st.active[st.active < n][17])
q_int = int(st.active[st.active >= n][23])
for targets, (lo, hi) in (("internal", (n64, st.maxnode)), ("leaf", (0, (n // 1024) * 1024)), ("all", (0, st.maxnode))):
    ops.set_shard(lo, hi)
    for ab in [int(a) for a in (sys.argv[1:] or ["0", "1", "3", "4", "7"])]:
        os.environ["VFT_SWEEP_ABLATE"] = str(ab)
        for _ in range(3):
            ops.setBestHit(q_int, st.n_active, st.n_diff_allow, st.totdiam, 0, want_best=False, want_hits=False)
        ops.synchronize()
        ops.timer_start()
        for _ in range(10):
            ops.setBestHit(q_int, st.n_active, st.n_diff_allow, st.totdiam, 0, want_best=False, want_hits=False)
        ops.timer_stop_ms()
        ms, nl = ops.sweep_kernel_ms()
        ms2, _ = ops.sweep_table_kernel_ms()
        print("%-9s targets, internal seed, ablate=%d: k_sweep_nt %.1f us + k_sweep_nt_table %.1f us = %.1f us" % (
            targets, ab, ms * 1e3, ms2 * 1e3, (ms + ms2) * 1e3))
