#!/usr/bin/env python3
"""Where a step of the walk server (csrc/vft_kernels_walk.h) spends its time: a variant library built with
VFT_EXTRA_HIPCC_FLAGS=-DVFT_WALK_TIMING, loaded through VFT_LIB_DIR - clock ticks (100 MHz) of thread 0 of workgroup 0, summed over a
complete pipeline; with the production library the wall-clock line only.  usage: walk_ticks.py N L [aa] [--no-server] [--device-mail] [--stride1]"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veryfasttree_amd import HipProfileOps, synth, backend
from veryfasttree_amd.backend import nj_newick, last_stage_seconds
n, L = int(sys.argv[1]), int(sys.argv[2])
aa = "aa" in sys.argv[3:]
# which walk: --no-server (a launch per step), --device-mail (the mailbox in device memory), --stride1 (six XCDs)
OPTS = [(9, 1)] * ("--no-server" in sys.argv) + [(10, 1)] * ("--device-mail" in sys.argv) + [(11, 1)] * ("--stride1" in sys.argv)


def make(m, Lp):
    ops = HipProfileOps(m, Lp, 20 if aa else 4, dt, max_nodes=3 * m)
    for o, val in OPTS:
        assert ops.lib.vft_debug_option(ops.ctx, ctypes.c_int32(o), ctypes.c_int64(val)) == 0
    return ops


dt = np.float64 if aa else np.float32
codes = synth.random_descent_codes(n, L, 20 if aa else 4, 0.02, 0.01, seed=4)
names = ["s%d" % k for k in range(n)]
kw = dict(aa_model="lg") if aa else {}
t0 = time.perf_counter()
nj_newick(make, codes, names, dtype=dt, me_lengths=True, me_nni=True, spr=2, **kw)
st = last_stage_seconds()
print("%d x %d: %.1f s, SPR %.2f s for %d steps (%.2f us per step)" % (n, L, time.perf_counter() - t0, st["of_which_spr"], st["spr_steps"], 1e6 * st["of_which_spr"] / max(st["spr_steps"], 1)))
t = (ctypes.c_int64 * 8)()
lib = backend.load_library()
if lib.vft_walk_server_ticks(t, 8) != 0 or t[6] == 0:
    sys.exit(0)   # (not a timing build: the wall-clock line only)
steps = t[6]
print("walk server, workgroup 0: %d steps with distances, %.2f averages per step; us per step:" % (steps, t[7] / steps))
for k, nm in enumerate(["waiting for the command (host + mailbox)", "averages (own slices)", "waiting for the other workgroups", "pair columns", "ordered sums", "answer"]):
    print("  %-42s %6.2f" % (nm, t[k] / 100.0 / steps))
