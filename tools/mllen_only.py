#!/usr/bin/env python3
"""`-nt -nome -mllen -nocat -nosupport` once on a synthetic alignment: mllen_only.py N L [out.tree] [level]
(`level`: the level-parallel rounds - vft_nj_options.debug_flags bit 16, not the reference's order)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veryfasttree_amd import HipProfileOps, synth
from veryfasttree_amd.backend import nj_newick
n, L = int(sys.argv[1]), int(sys.argv[2])
codes = synth.random_descent_codes(n, L, 4, 0.03, 0.01, seed=2)
names = ["s%d" % k for k in range(n)]
t0 = time.perf_counter()
tree, ll = nj_newick(lambda m, Lp: HipProfileOps(m, Lp, 4, np.float32, max_nodes=3 * m), codes, names, me_lengths=True,
                     mllen=1, return_loglk=True, debug_flags=16 if "level" in sys.argv[3:] else 0)
print("whole command: %.2f s, %d rounds, TreeLogLk %s" % (time.perf_counter() - t0, len(ll), " ".join("%.4f" % x for x in ll)))
if len(sys.argv) > 3 and sys.argv[3] != "level":
    open(sys.argv[3], "w").write(tree + "\n")
