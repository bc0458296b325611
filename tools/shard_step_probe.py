#!/usr/bin/env python3
"""What ONE rank of a W-rank job computes per bench step (1 x MI355X, no communication): bench.py's workload and seeds, the target range
set to rank r's shard of W (veryfasttree_amd.workload.shard_range, as bench.py --gpus W does), S seeds swept and selected per step.
The N = 8 projection of DESIGN.md section 5 uses these numbers instead of dividing the single-GPU kernels by eight.
usage: shard_step_probe.py [n_seqs] [n_pos] [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veryfasttree_amd import HipProfileOps, synth
from veryfasttree_amd.workload import TopHitsState, shard_range, sweep_cost_weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 200
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
m = int(0.5 + np.sqrt(n))
k = 2 * m
codes = synth.random_descent_codes(n, L, 4, 0.02, 0.01, seed=4)
ops = HipProfileOps(n, L, 4, np.float32)
state = TopHitsState(ops, codes, n // 4)
leaf_act = state.active[state.active < n]
int_act = state.active[state.active >= n]
seeds = []
for s in range(4):
    seeds.append(int(leaf_act[(s * 7919 + 13) % len(leaf_act)]))
    seeds.append(int(int_act[(s * 104729 + 7) % len(int_act)]))
seeds = np.asarray(seeds, np.int64)
weights = sweep_cost_weights(state.parent, n)
if os.environ.get("VFT_PROBE_MULTI"):   # VFT_DEBUG_NO_MULTI_SWEEP: 1 = a launch per seed, 2 / 4 = that many seeds per shared pass
    import ctypes
    assert ops.lib.vft_debug_option(ops.ctx, ctypes.c_int32(12), ctypes.c_int64(int(os.environ["VFT_PROBE_MULTI"]))) == 0
    print("# seeds per shared pass forced: VFT_DEBUG_NO_MULTI_SWEEP =", os.environ["VFT_PROBE_MULTI"])
print("# %d x %d, %d active, %d seeds per step, k = %d; us per step of one rank's local work (sweeps + selection, records left in the result blocks)" % (n, L, state.n_active, len(seeds), k))
print("%5s %5s %12s %12s %10s" % ("world", "rank", "lo", "hi", "us/step"))
for world in (1, 2, 4, 8):
    for rank in sorted({0, world // 2, world - 1}):
        lo, hi = shard_range(state.maxnode, rank, world, weights)
        ops.set_shard(lo, hi)
        for _ in range(5):
            ops.setBestHitBatch(seeds, state.n_active, state.n_diff_allow, state.totdiam, k, view=True)
        ops.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ops.setBestHitBatch(seeds, state.n_active, state.n_diff_allow, state.totdiam, k, view=True)
        ops.synchronize()
        print("%5d %5d %12d %12d %10.1f" % (world, rank, lo, hi, 1e6 * (time.perf_counter() - t0) / steps), flush=True)
