#!/usr/bin/env python3
"""Whole NJ-only pipeline on a synthetic alignment: nj_tree_synth.py N L [fastest] [full] -> phase timings (full: only the last variant) (VFT_NJ_PROFILE=1
adds the per-call table).  Tree = fastNJ + root + ME branch lengths + 1000-resample supports, as tools/nj_tree.py."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veryfasttree_amd import HipProfileOps, synth
from veryfasttree_amd.backend import nj_newick
n, L = int(sys.argv[1]), int(sys.argv[2])
fastest = len(sys.argv) > 3 and sys.argv[3] == "fastest"
codes = synth.random_descent_codes(n, L, 4, 0.03, 0.01, seed=3)
names = ["s%d" % k for k in range(n)]
variants = (("NJ tree (NJ lengths)", dict()), ("+ ME lengths", dict(me_lengths=True)),
            ("+ ME lengths + 1000-resample supports", dict(me_lengths=True, n_bootstrap=1000)))
if "full" in sys.argv[3:]:
    variants = variants[-1:]
for label, kw in variants:
    t0 = time.perf_counter()
    tree = nj_newick(lambda m, Lp: HipProfileOps(m, Lp, 4, np.float32, max_nodes=3 * m), codes, names, fastest=fastest,
                     second_level=False, **kw)
    print("%-40s %8.2f s   (%d characters of Newick)" % (label, time.perf_counter() - t0, len(tree)), flush=True)
