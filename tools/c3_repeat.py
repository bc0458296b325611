"""The C3 tree (100k x 500, -nt -fastest, NJ + ME lengths) twice in one process: length and CRC-32 of the Newick must be
2539208 / 1604271881 (the reference's, tests/golden/bb_c3_crc.npz) every time.  c3_repeat.py [plain|torch] [sdma0]"""
import sys, zlib, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
if len(sys.argv) > 2 and sys.argv[2] == "sdma0":
    os.environ["HSA_ENABLE_SDMA"] = "0"
import numpy as np
from veryfasttree_amd import HipProfileOps, synth
from veryfasttree_amd.backend import nj_newick
n, L = 100000, 500
codes = synth.random_descent_codes(n, L, 4, 0.03, 0.01, seed=3)
names = ["s%d" % k for k in range(n)]
for rep in range(2):
    t0 = time.perf_counter()
    tree = nj_newick(lambda m, Lp: HipProfileOps(m, Lp, 4, np.float32, max_nodes=3 * m), codes, names, fastest=True, me_lengths=True)
    print(sys.argv[1:], rep, len(tree), zlib.crc32(tree.encode()), "%.1f s" % (time.perf_counter() - t0), flush=True)
