#!/usr/bin/env python3
"""config C4's generator at 200 000 sequences on the 64-thread schedule: the tree into gpurun_out/ (compared with the reference's in the build container)"""
import os, sys, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veryfasttree_amd import HipProfileOps, synth
from veryfasttree_amd.backend import nj_newick
codes = synth.random_descent_codes(200000, 200, 4, 0.02, 0.01, seed=4)
names = ["s%d" % k for k in range(len(codes))]
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
tree, loglk = nj_newick(lambda m, Lp: HipProfileOps(m, Lp, 4, np.float32, max_nodes=3 * m), codes, names, me_lengths=True, me_nni=True, spr=2,
                        ml_nni=20, n_bootstrap=nb, return_loglk=True, threads=64)
open(sys.argv[1], "w").write(tree + "\n")
print(len(tree), zlib.crc32(tree.encode()), list(loglk))
