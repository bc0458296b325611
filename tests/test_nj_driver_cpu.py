"""Join-order parity of the host NJ driver against the reference's own `Join` lines (tests/golden/bb_*.npz, produced
by oracle/_ref/VeryFastTree -threads 1 -verbose 3).  The backend here is the oracle (CPU); tests/test_gpu_parity.py
runs the same driver on the HIP backend."""
import numpy as np
import pytest

import golden_util as G
from oracle_ops import OracleOps
from nj_driver_py import NJDriver
from veryfasttree_amd.synth import NOCODE


def unique_codes(codes):
    """first-occurrence uniquify (Alignment.cpp:494-526)"""
    seen, keep = set(), []
    for i, row in enumerate(codes):
        key = row.tobytes()
        if key not in seen:
            seen.add(key)
            keep.append(i)
    return codes[keep]


@pytest.mark.parametrize("name,fastest,second,limit", [("bb_nt_c1", True, True, None), ("bb_nt_200", False, False, None),
                                                       ("bb_nt_600_fastest_no2nd", True, False, None),
                                                       ("bb_nt_600_fastest", True, True, None),
                                                       ("bb_nt_1500", False, False, 450),
                                                       ("bb_nt_300_double", False, False, None)])
def test_join_order_matches_reference(name, fastest, second, limit):
    d = G.load(name)
    codes = unique_codes(d["codes"])
    ops = OracleOps(codes.shape[0], codes.shape[1], 4, np.float64 if "double" in name else np.float32)
    drv = NJDriver(ops, codes, fastest=fastest, use_tophits_2nd=second)
    if fastest:
        drv.tophits_refresh = 0.5   # main.cpp:339-343: -fastest
    joins = drv.run(max_joins=limit)
    want = d["joins"][:len(joins)]
    got = np.array([(a, b, c) for a, b, c, _ in joins], dtype=np.int64)
    assert len(got) > 0
    first_bad = np.nonzero((got != want).any(axis=1))[0]
    assert len(first_bad) == 0, "first differing join %d: got %s want %s" % (first_bad[0], got[first_bad[0]],
                                                                            want[first_bad[0]])
