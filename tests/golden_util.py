"""Helpers shared by the CPU (oracle) and GPU parity tests: load a white-box fixture and rebuild the
state it describes.  Fixtures are produced by oracle/gen_fixtures.py from the compiled reference."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NOCODE = 127

WHITEBOX = ["wb_nt_f32", "wb_nt_f32_gappy", "wb_nt_f64", "wb_aa_f32", "wb_aa_f64"]
WHITEBOX_NT = [n for n in WHITEBOX if "_nt_" in n]
WHITEBOX_AA = [n for n in WHITEBOX if "_aa_" in n]


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def dtype_of(d):
    return np.float32 if int(d["precisionBytes"]) == 4 else np.float64


def fixture_profile(d, prefix):
    """(w, c, f) of a profile dumped by the harness under `prefix`."""
    return (np.ascontiguousarray(d[prefix + ".w"]), np.ascontiguousarray(d[prefix + ".c"]),
            np.ascontiguousarray(d[prefix + ".f"]))


def profiles_equal(a, b):
    """Exact equality on everything the reference stores: weights, codes and the vectors of vector columns."""
    if not (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])):
        return False
    has = (a[0] > 0) & (a[1] == NOCODE)
    return np.array_equal(a[2][has], b[2][has])


def internal_nodes(d):
    """Ids of internal nodes that carry a profile (everything but the 3-way root), children first."""
    n_seqs = int(d["nSeqs"])
    root = int(d["nj.root"])
    return [v for v in range(n_seqs, root)]


def build_nj_profiles(d, orc, dm=None):
    """Rebuild every node profile of the reference's NJ tree with the oracle's averageProfile."""
    n_seqs, n_codes = int(d["nSeqs"]), int(d["nCodes"])
    tol = 1e-10 if orc.dt == np.float32 else 1e-20
    profs = [orc.leaf_profile(d["leaf.codes"][i], n_codes) for i in range(n_seqs)]
    child = d["nj.child"]
    for v in internal_nodes(d):
        a, b = int(child[v, 0]), int(child[v, 1])
        profs.append(orc.average_profile(profs[a], profs[b], -1.0, dm, tol))
    return profs


def pack(profs, dt):
    W = np.ascontiguousarray(np.stack([p[0] for p in profs]), dtype=dt)
    Cc = np.ascontiguousarray(np.stack([p[1] for p in profs]), dtype=np.uint8)
    F = np.ascontiguousarray(np.stack([p[2] for p in profs]), dtype=dt)
    return W, Cc, F


def mid_parent(d, J):
    """parent[] as the reference saw it after J joins: active <=> parent created later (or never)."""
    n_seqs = int(d["nSeqs"])
    lim = n_seqs + J
    parent = d["nj.parent"][:lim].copy()
    parent[parent >= lim] = -1
    return parent


def dmat_of(d, orc):
    if "dmat.distances" not in d:
        return None
    return orc.dmat(d["dmat.distances"], d["dmat.codefreq"], d["dmat.eigenval"], d["dmat.eigentot"])


def tmat_of(d, orc, prefix):
    k = prefix + ".tm."
    return orc.tmat(d[k + "stat"], d[k + "statinv"], d[k + "eigenval"], d[k + "codefreq"], d[k + "eigeninv"],
                    d[k + "eigeninvT"])
