#!/usr/bin/env python3
"""config C4's generator at n sequences (argv[3], default 200 000) on the 64-thread schedule: the tree into gpurun_out/ (compared with the reference's in the build container)"""
import os, sys, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veryfasttree_amd import HipProfileOps, synth
from veryfasttree_amd.backend import nj_newick
n = int(sys.argv[3]) if len(sys.argv) > 3 else 200000
codes = synth.random_descent_codes(n, 200, 4, 0.02, 0.01, seed=4)
names = ["s%d" % k for k in range(len(codes))]
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
import ctypes, time
from veryfasttree_amd.backend import last_stage_seconds


def make(m, Lp):
    ops = HipProfileOps(m, Lp, 4, np.float32, max_nodes=3 * m)
    if os.environ.get("VFT_JC_EXACT"):   # Jukes-Cantor likelihoods bit for bit the reference's (vft_set_jc_exact)
        assert ops.lib.vft_set_jc_exact(ops.ctx, ctypes.c_int32(int(os.environ["VFT_JC_EXACT"]))) == 0
    return ops


t0 = time.perf_counter()
tree, loglk = nj_newick(make, codes, names, me_lengths=True, me_nni=True, spr=2,
                        ml_nni=20, n_bootstrap=nb, return_loglk=True, threads=int(os.environ.get("VFT_THREADS", "64")))
print("wall %.1f s" % (time.perf_counter() - t0), {k: v for k, v in last_stage_seconds().items() if k in ("nj", "me_nni_spr", "ml_stage", "of_which_ml_nni", "of_which_sh_supports", "of_which_model_fits")})
open(sys.argv[1], "w").write(tree + "\n")
print(len(tree), zlib.crc32(tree.encode()), list(loglk))
