"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol include/vft_hip.h
declares, and refuses to run without a GPU (no silent CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from veryfasttree_amd import build
    build.build()
    from veryfasttree_amd import backend
    return backend.load_library()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vft_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vft_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    names = declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_host_driver_library_exports_its_header(lib):
    from veryfasttree_amd import backend
    host = backend.load_host_library()
    text = open(os.path.join(ROOT, "include", "vft_host.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(vft_(?:nj|knuth|ml|gtr|aa|blosum45|tree)_[a-z0-9_]+)\s*\(", text)))
    assert names == sorted(backend.HOST_EXPORTS)
    assert all(hasattr(host, n) for n in names)


def test_python_binding_lists_every_symbol():
    from veryfasttree_amd import backend
    assert sorted(backend.EXPORTS) == declared_symbols()


def test_no_cpu_fallback_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from veryfasttree_amd import HipProfileOps, VftError
    with pytest.raises(VftError):
        HipProfileOps(8, 16, 4, np.float32)


def test_bad_config_is_rejected(lib):
    from veryfasttree_amd import HipProfileOps, VftError
    with pytest.raises(VftError):
        HipProfileOps(8, 16, 5, np.float32)


def test_host_knuth_generator_matches_the_reference_stream():
    """veryfasttree_amd/host/KnuthRng.h (the product's column resampler) against 5000 values of the reference's
    knuth_rand() stream (tests/golden/wb_knuth.npz)."""
    import ctypes as C
    import numpy as np
    import golden_util as G
    from veryfasttree_amd.backend import load_host_library
    lib = load_host_library()
    ref = G.load("wb_knuth")["knuth.rand"]
    out = np.zeros(len(ref), np.float64)
    lib.vft_knuth_stream(out.ctypes.data_as(C.c_void_p), C.c_int64(len(ref)))
    assert np.array_equal(out, ref)


def test_gtr_tables_equal_the_references():
    """createGTR + eigen-decomposition of the host driver against the tables the reference built for the same rates and
    frequencies (white-box fixture gtr.tm.*, stored as numeric_t)."""
    import golden_util as G
    from veryfasttree_amd import backend
    rates, freq = [1.2, 3.1, 0.7, 0.9, 3.6, 1.0], [0.31, 0.19, 0.23, 0.27]   # oracle/whitebox.cpp
    for name in ("wb_nt_f32", "wb_nt_f64"):
        d = G.load(name)
        dt = d["gtr.tm.stat"].dtype
        t = backend.gtr_tables(rates, freq, dt)
        for key in ("stat", "statinv", "eigenval", "codefreq", "eigeninv", "eigeninvT"):
            want = d["gtr.tm." + key]
            got = t[key].astype(dt).reshape(want.shape)
            assert np.array_equal(got, want), (name, key, got, want)


def test_amino_acid_model_tables_equal_the_references():
    """JTT92 / WAG01 / LG08 (host/AAModels.h: constants + eigen-decomposition) and the BLOSUM45-derived distance matrix
    against the tables the reference builds (createTransitionMatrix{JTT92,WAG01,LG08}, matrixBLOSUM45 +
    setupDistanceMatrix; tests/golden/wb_aa_tables.npz), bit for bit in both precisions; and the transition matrix in the
    distance-matrix slots (transMatToDistanceMat, VeryFastTreeImpl.tcc:517-542)."""
    import golden_util as G
    from veryfasttree_amd import backend
    d = G.load("wb_aa_tables")
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        for model in ("jtt", "wag", "lg"):
            t = backend.aa_model_tables(model, dt)
            for key in ("stat", "statinv", "eigenval", "codefreq", "eigeninv", "eigeninvT"):
                want = d["%s.%s.%s" % (model, tag, key)]
                assert want.dtype == dt
                got = t[key].astype(dt).reshape(want.shape)
                assert np.array_equal(got, want), (model, tag, key)
            # transMatToDistanceMat: codeFreq rows of the 20 codes, eigentot = row sums of eigeninv in numeric_t
            td = backend.distance_tables(model, dt)
            assert np.array_equal(td["codefreq"].astype(dt), d["%s.%s.codefreq" % (model, tag)][:20])
            tot = np.zeros(20, dt)
            for j in range(20):
                tot = (tot + d["%s.%s.eigeninv" % (model, tag)][:, j]).astype(dt)
            assert np.array_equal(td["eigentot"].astype(dt), tot)
            assert not td["distances"].any() and not td["eigenval"].any()
        b = backend.distance_tables(None, dt)
        for key in ("distances", "codefreq", "eigenval", "eigentot"):
            want = d["blosum45.%s.%s" % (tag, key)]
            assert np.array_equal(b[key].astype(dt).reshape(want.shape), want), (tag, key)
    # the white-box fixtures of round 1 carry the same BLOSUM45 / LG tables: cross-check
    w = G.load("wb_aa_f64")
    assert np.array_equal(backend.distance_tables(None, np.float64)["codefreq"], w["dmat.codefreq"])
    assert np.array_equal(backend.aa_model_tables("lg", np.float64)["codefreq"], w["lg.tm.codefreq"])
