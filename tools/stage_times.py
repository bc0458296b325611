#!/usr/bin/env python3
"""Where the wall-clock of one complete pipeline goes (vft_nj_last_stage_seconds): stage_times.py N L [nt|aa] [threads] [gtr|lg] [f64] [noserver] [mu=M] [seed=S]
(noserver: the SPR / NNI walks without the resident walk server, vft_nj_options.debug_flags bit 128)"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veryfasttree_amd import HipProfileOps, synth
from veryfasttree_amd.backend import nj_newick, last_stage_seconds
n, L = int(sys.argv[1]), int(sys.argv[2])
rest = sys.argv[3:]
aa = "aa" in rest
T = max([int(a) for a in rest if a.isdigit()] or [1])
dt = np.float64 if "f64" in rest else np.float32
kv = dict(a.split("=") for a in rest if "=" in a)   # mu=0.02 seed=4: config C4's alignment
codes = synth.random_descent_codes(n, L, 20 if aa else 4, float(kv.get("mu", 0.03)), 0.01, seed=int(kv.get("seed", 2)))
names = ["s%d" % k for k in range(n)]
kw = dict(dtype=dt, me_lengths=True, me_nni=True, spr=2, ml_nni=20, n_bootstrap=1000, threads=T, return_loglk=True, debug_flags=128 if "noserver" in rest else 0)
if aa:
    kw["aa_model"] = "lg" if "lg" in rest else "jtt"
elif "gtr" in rest:
    kw["gtr"] = True
t0 = time.perf_counter()
def make(m, Lp):
    return HipProfileOps(m, Lp, 20 if aa else 4, dt, max_nodes=3 * m)
tree, ll = nj_newick(make, codes, names, **kw)
print("%d x %d %s %s threads=%d: %.1f s, TreeLogLk %.4f" % (n, L, "aa" if aa else "nt", " ".join(r for r in rest if not r.isdigit()), T, time.perf_counter() - t0, ll[-1]))
import zlib
print(json.dumps(dict(last_stage_seconds(), newick_bytes=len(tree), newick_crc=zlib.crc32(tree.encode()), tree_loglk=[round(float(x), 4) for x in ll])))
