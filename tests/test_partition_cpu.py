"""vft_tree_partitioning (host/MLLengths.h partitionTree - what the subtree schedule makes its lanes from) against the reference's own
treePartitioning (NJ.tcc:5540-5750): oracle/whitebox.cpp calls the private member on the NJ trees of three alignments for penalties
1 and 2 and 2 ... 64 threads (oracle/gen_fixtures.py partition -> tests/golden/wb_partition_*.npz).  Pure host code, no GPU."""
import ctypes as C

import numpy as np
import pytest

import golden_util as G


@pytest.mark.parametrize("name", ["wb_partition_40", "wb_partition_300", "wb_partition_2000"])
def test_partitions_equal_the_references(name):
    from veryfasttree_amd.backend import load_host_library
    lib = load_host_library()
    d = G.load(name)
    child = np.ascontiguousarray(d["nj.child"], np.int64)
    n, root = child.shape[0], int(d["nj.root"])
    cases = 0
    for key in sorted(d):
        if not key.startswith("part."):
            continue
        threads, penalty = int(key.split(".")[1][1:]), int(key.split(".")[2][1:])
        want = d[key][d[key] >= 0]   # (the reference pads the threads' lists with -1)
        out = np.zeros(n, np.int64)
        n_out, speedup = C.c_int64(0), C.c_double(0)
        rc = lib.vft_tree_partitioning(C.c_int64(n), child.ctypes.data_as(C.c_void_p), C.c_int64(root), C.c_int32(penalty), C.c_int32(threads),
                                       C.c_int32(0), out.ctypes.data_as(C.c_void_p), C.c_int64(n), C.byref(n_out), C.byref(speedup))
        assert rc == 0
        assert np.array_equal(out[:n_out.value], want), (key, out[:8], want[:8])
        # an antichain of internal nodes: no root of a subtree lies below another
        roots = set(int(x) for x in want)
        parent = d["nj.parent"]
        for r in roots:
            p = int(parent[r])
            while p >= 0:
                assert p not in roots
                p = int(parent[p])
        assert speedup.value >= 1.0 or len(want) == 0
        cases += 1
    assert cases == 12


def test_malformed_trees_are_refused():
    """the child array comes from the caller of a public C entry point: ids out of range, a node with two parents (a cycle or a DAG) and
    the root as somebody's child are VFT_ERR_INVALID, not a walk outside the arrays or an endless one; a huge thread count is clamped"""
    from veryfasttree_amd.backend import load_host_library
    lib = load_host_library()
    d = G.load("wb_partition_40")
    good = np.ascontiguousarray(d["nj.child"], np.int64)
    n, root = good.shape[0], int(d["nj.root"])

    def call(child, threads=4):
        out = np.zeros(n, np.int64)
        n_out, speedup = C.c_int64(0), C.c_double(0)
        return lib.vft_tree_partitioning(C.c_int64(n), child.ctypes.data_as(C.c_void_p), C.c_int64(root), C.c_int32(2), C.c_int32(threads),
                                         C.c_int32(0), out.ctypes.data_as(C.c_void_p), C.c_int64(n), C.byref(n_out), C.byref(speedup))

    assert call(good) == 0
    assert call(good, threads=2 ** 31 - 1) == 0
    internal = [v for v in range(n) if good[v, 0] >= 0 and v != root]
    bad = good.copy()
    bad[internal[0], 0] = n + 5            # out of range
    assert call(bad) == 1
    bad = good.copy()
    bad[internal[0], 0] = -7
    assert call(bad) == 1
    bad = good.copy()
    bad[internal[1], 1] = bad[internal[0], 0]   # two parents for one node
    assert call(bad) == 1
    bad = good.copy()
    bad[internal[0], 1] = root             # the root below one of its descendants
    assert call(bad) == 1
