"""Diagnostic: one quartet of a white-box fixture through vft_ml_split_tests and vft_ml_quartet_nni (argv: fixture, model)."""
import sys, ctypes as C
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, golden_util as G
from test_gpu_ml_lengths import _setup
from veryfasttree_amd import backend
from veryfasttree_amd.backend import _ptr, I64, I32
name = sys.argv[1] if len(sys.argv) > 1 else "wb_nt_f32"
model = sys.argv[2] if len(sys.argv) > 2 else "gtr"
d = G.load(name); n_seqs, root = int(d["nSeqs"]), int(d["nj.root"])
ops = _setup(d, model, 8)
bl, ll, ev = backend.ml_lengths(ops, n_seqs, d["nj.parent"][:root+1], d["nj.child"][:root+1], root, d["nj.branchlength"][:root+1], rounds=1, n_leaf_gaps=-1)
print("lengths ok", ll)
child, parent = d["nj.child"], d["nj.parent"]
# one quartet: a node whose parent is the root (D = other root child, no up-profile needed)
rc = [int(c) for c in child[root][:3]]
v = [c for c in rc if c >= n_seqs][0]
sibs = [c for c in rc if c != v]
ids = np.array([child[v][0], child[v][1], sibs[0], sibs[1]], np.int64)
li = np.array([child[v][0], child[v][1], sibs[0], sibs[1], v], np.int64)
loglk = np.zeros(3)
print("ids", ids)
r = ops.lib.vft_ml_split_tests(ops.ctx, I64(1), _ptr(ids), _ptr(li), C.c_double(0.001), C.c_double(1e-4), C.c_double(5.0), I32(0), _ptr(loglk), I32(0), None, None, None)
print("split test rc", r, loglk)
class R(C.Structure):
    _fields_ = [("criteria", C.c_double * 3), ("choice", I32), ("star", I32)]
res = R()
r = ops.lib.vft_ml_quartet_nni(ops.ctx, I64(1), _ptr(ids), _ptr(li), C.c_double(0.001), C.c_double(1e-4), C.c_double(5.0), I32(1), C.byref(res))
print("nni rc", r, list(res.criteria), res.choice, res.star)
