#!/usr/bin/env python3
"""SPR rounds with and without the dual commands (both continuations of a chain step handed to the walk server): seconds, steps, us per
step, and the trees' CRCs.  usage: spr_dual_timing.py [n] [L] [nt|aa]"""
import os, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veryfasttree_amd import HipProfileOps, synth
from veryfasttree_amd.backend import nj_newick, last_stage_seconds, DEBUG_NO_WALK_DUAL
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 200
aa = len(sys.argv) > 3 and sys.argv[3] == "aa"
nc, dt = (20, np.float64) if aa else (4, np.float32)
codes = synth.random_descent_codes(n, L, nc, 0.03, 0.01, seed=2)
names = ["s%d" % k for k in range(n)]
kw = dict(dtype=dt, me_lengths=True, me_nni=True, spr=2)
if aa:
    kw["aa_model"] = "lg"
for label, flags in (("every step waits for the host", DEBUG_NO_WALK_DUAL), ("dual commands", 0), ("every step waits for the host", DEBUG_NO_WALK_DUAL), ("dual commands", 0)):
    tree = nj_newick(lambda m, Lp: HipProfileOps(m, Lp, nc, dt, max_nodes=3 * m), codes, names, debug_flags=flags, **kw)
    st = last_stage_seconds()
    print("%-32s SPR %.3f s, %d steps, %.2f us per step; dual commands %d, continuations run by the device %d; tree crc %d" % (
        label, st["of_which_spr"], st["spr_steps"], 1e6 * st["of_which_spr"] / max(st["spr_steps"], 1), st["spr_dual_commands"], st["spr_dual_continuations"], zlib.crc32(tree.encode())))
