"""vft_glibc_log (veryfasttree_amd/csrc/vft_glibc_log.h), the logarithm the ML kernels end a float-precision matrix-model
pairLogLk with, must be this image's libm log() bit for bit: on the host against libm itself (2 x 10^7 arguments over the
ranges the likelihood code produces and the whole exponent range), on the device against math.log (= libm)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_restated_log_equals_libm_on_the_host(tmp_path):
    exe = str(tmp_path / "glogcheck")
    subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-I" + os.path.join(ROOT, "veryfasttree_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "glibc_log_check.c"), "-lm", "-o", exe], check=True)
    res = subprocess.run([exe, "20000000"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert res.returncode == 0, res.stdout.decode()[-2000:]
    assert res.stdout.decode().split("\n")[:2] == ["0 mismatches in 20000000", "0 exp mismatches in 20000000"]


@pytest.mark.gpu
def test_device_log_equals_libm():
    from veryfasttree_amd import HipProfileOps
    ops = HipProfileOps(8, 16, 4, np.float32)
    rng = np.random.default_rng(11)
    import math
    x = np.concatenate([np.exp(rng.uniform(-9.3, 9.3, 200000)), rng.uniform(0.93, 1.07, 200000), rng.uniform(0, 1, 100000) + 1e-300,
                        np.array([1.0, 0.9375, 1.0 + float.fromhex("0x1.09p-4"), 1e-4, 1e4, 0.5, 2.0])])
    got = ops.debug_log(x)
    want = np.array([math.log(v) for v in x])   # libm itself (numpy's vectorised log is a different implementation)
    bad = got.view(np.int64) != want.view(np.int64)
    assert not bad.any(), (int(bad.sum()), x[bad][:5], got[bad][:5], want[bad][:5])
    ops.close()
