"""The drop-in boundary in the reference's own terms: HipOperations<Precision> must be acceptable
wherever the reference takes an Operations template (NeighbourJoining.h:19-22) and its ten per-vector methods must be
BasicOperations' bit for bit.  Both checks compile against the reference headers where they lie, so they only run in
the authoring container (skipped when /root/reference is absent, e.g. on the GPU box)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference headers not present")

INC = ["-I" + os.path.join(REF, p) for p in ("src", "libs/CLI11/include", "libs/bxzstr/include", "libs/boost-align/include",
                                             "libs/boost-core/include", "libs/boost-sort/include", "libs/bzip2",
                                             "libs/robin-map/include", "libs/xxhash/include")] + \
      ["-I" + os.path.join(ROOT, "veryfasttree_amd", "host")]
FLAGS = ["-std=c++11", "-fopenmp", "-DNDEBUG", "-DCLI11_BOOST_OPTIONAL=0", "-DUSE_CUDA=0", "-w"]   # oracle/Makefile's REFFLAGS


def test_pipeline_template_instantiates_with_hip_operations(tmp_path):
    """template class VeyFastTreeImpl<float|double, HipOperations>: the registration unit of INTEGRATION.md section 1
    compiles against the reference's headers - every member the pipeline calls on its Operations slot exists with a
    compatible signature (ALIGNMENT, Allocator, numeric_t, the ten methods, default construction)."""
    obj = str(tmp_path / "instantiate.o")
    res = subprocess.run(["g++", "-O0"] + FLAGS + INC + ["-c", os.path.join(ROOT, "tests", "trait", "instantiate.cpp"), "-o", obj],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert res.returncode == 0, res.stdout.decode()[-4000:]
    syms = subprocess.run(["nm", "-C", "--defined-only", obj], stdout=subprocess.PIPE, check=True).stdout.decode()
    for p in ("float", "double"):
        assert "veryfasttree::NeighbourJoining<%s, veryfasttree::HipOperations>::fastNJ()" % p in syms
        assert "veryfasttree::VeyFastTreeImpl<%s, veryfasttree::HipOperations>::run()" % p in syms


def test_ten_primitives_equal_basic_operations(tmp_path):
    """vector_multiply ... fastexp (levels 0-3) against BasicOperations on random 4 / 20 / 24 / 400 / 4000-element
    vectors, float and double, in-place aliasing included: bit-exact."""
    from veryfasttree_amd import build
    build.build()
    exe = str(tmp_path / "primitives")
    lib = os.path.join(ROOT, "veryfasttree_amd", "lib")
    res = subprocess.run(["g++", "-O2", "-mavx2"] + FLAGS + INC + [os.path.join(ROOT, "tests", "trait", "primitives.cpp"), "-o", exe,
                          "-L" + lib, "-lvft_hip", "-Wl,-rpath," + lib], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert res.returncode == 0, res.stdout.decode()[-4000:]
    run = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    out = run.stdout.decode()
    assert run.returncode == 0 and out.startswith("ok "), out[-4000:]
    assert int(out.split()[1]) > 20000
