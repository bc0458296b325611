"""Builds the HIP extension in-tree (veryfasttree_amd/lib/libvft_hip.so) with hipcc for gfx950.

-ffp-contract=off is part of the numerics contract: distance sums must not be fused.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libvft_hip.so")
SOURCES = [os.path.join(CSRC, f) for f in ("vft_api.hip", "vft_ml_kernels_lengths.hip", "vft_ml_kernels_quartet32.hip",
                                             "vft_ml_kernels_quartet64.hip", "vft_ml_kernels_long.hip", "vft_walk_kernels.hip")]
HEADERS = ["vft_layout.h", "vft_device.h", "vft_kernels_nj.h", "vft_kernels_aa.h", "vft_kernels_profile.h", "vft_kernels_tophits.h", "vft_kernels_njengine.h", "vft_kernels_walk.h", "vft_kernels_ml.h", "vft_kernels_ml_long.h", "vft_iterate_add.h", "vft_glibc_log.h",
           "vft_glibc_log_data.h"]   # deps of every unit
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-value"]
# Tools only (e.g. -DVFT_ABLATE, -DVFT_ML_TIMING, -DVFT_NJ_TIMING): a VARIANT build.  Its objects and its libraries live in
# build/variants/<sha1 of the flags>/ - never in build/obj or veryfasttree_amd/lib - so that an instrumented or ablated
# kernel cannot end up in the production library.  Tools load a variant with VFT_LIB_DIR=<that directory>.
EXTRA_FLAGS = os.environ.get("VFT_EXTRA_HIPCC_FLAGS", "").split()
FLAGS = BASE_FLAGS + EXTRA_FLAGS


HOST_LIB = os.path.join(HERE, "lib", "libvft_host.so")
HOST_SOURCES = [os.path.join(HERE, "host", "nj_driver.cpp")]
HOST_DEPS = HOST_SOURCES + [os.path.join(HERE, "host", h) for h in ("NJDriver.h", "MLLengths.h", "KnuthRng.h", "GtrModel.h", "AAModels.h", "AAModelData.h")] + \
    [os.path.join(HERE, "..", "include", "vft_host.h"), os.path.join(HERE, "..", "include", "vft_hip.h")]


def variant_dir():
    import hashlib
    return os.path.join(HERE, "..", "build", "variants", hashlib.sha1(" ".join(EXTRA_FLAGS).encode()).hexdigest()[:12])


def build_host(force=False, lib=None, host_lib=None):
    """The C++ host driver: plain g++, links against the HIP library next to it."""
    lib, host_lib = lib or LIB, host_lib or HOST_LIB
    if not force and os.path.exists(host_lib) and all(os.path.getmtime(d) <= os.path.getmtime(host_lib) for d in HOST_DEPS + [lib]):
        return host_lib
    subprocess.run(["g++", "-O3", "-mavx2", "-ffp-contract=off", "-std=c++11", "-fopenmp", "-fPIC", "-shared", "-Wall", "-o", host_lib] + HOST_SOURCES +
                   ["-L" + os.path.dirname(lib), "-lvft_hip", "-Wl,-rpath,$ORIGIN"], check=True)
    return host_lib


ML_HEADERS = ["vft_layout.h", "vft_device.h", "vft_kernels_profile.h", "vft_iterate_add.h", "vft_kernels_ml.h", "vft_glibc_log.h", "vft_glibc_log_data.h"]   # deps of the vft_ml_kernels_*.hip units
OBJ_DIR = os.path.join(HERE, "..", "build", "obj")   # git-ignored; objects are kept so that a header change recompiles only its units


WALK_HEADERS = ["vft_layout.h", "vft_device.h", "vft_kernels_profile.h", "vft_iterate_add.h", "vft_kernels_walk.h", "vft_glibc_log.h", "vft_glibc_log_data.h"]   # deps of vft_walk_kernels.hip


def unit_deps(src):
    if os.path.basename(src) == "vft_walk_kernels.hip":
        return [src] + [os.path.join(CSRC, h) for h in WALK_HEADERS]
    if os.path.basename(src) == "vft_ml_kernels_long.hip":
        return [src] + [os.path.join(CSRC, h) for h in ML_HEADERS + ["vft_kernels_ml_long.h"]]
    if os.path.basename(src) != "vft_api.hip":   # the explicit-instantiation units see the ML kernels only
        return [src] + [os.path.join(CSRC, h) for h in ML_HEADERS]
    return [src] + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(HERE, "..", "include", "vft_hip.h")]


def stale_units(force=False, obj_dir=None):
    out = []
    for src in SOURCES:
        obj = os.path.join(obj_dir or OBJ_DIR, os.path.basename(src) + ".o")
        if force or not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in unit_deps(src)):
            out.append((src, obj))
    return out


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = SOURCES + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(HERE, "..", "include", "vft_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def compile_and_link(lib, obj_dir, force):
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # one hipcc per translation unit whose sources changed, in parallel (the line-search kernels of the
    # vft_ml_kernels_*.hip units take as long as everything else together), then the link
    todo = stale_units(force, obj_dir)
    jobs = [subprocess.Popen([hipcc] + FLAGS + ["-c", "-o", obj, src]) for src, obj in todo]
    for (src, obj), job in zip(todo, jobs):
        if job.wait() != 0:
            if os.path.exists(obj):
                os.remove(obj)
            raise subprocess.CalledProcessError(job.returncode, "hipcc -c " + src)
    objs = [os.path.join(obj_dir, os.path.basename(src) + ".o") for src in SOURCES]
    subprocess.run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", lib] + objs, check=True)


def build(force=False):
    if EXTRA_FLAGS:   # a variant: its own objects, its own libraries (see EXTRA_FLAGS above)
        vdir = variant_dir()
        lib = os.path.join(vdir, "libvft_hip.so")
        compile_and_link(lib, os.path.join(vdir, "obj"), force)
        with open(os.path.join(vdir, "FLAGS"), "w") as f:
            f.write(" ".join(EXTRA_FLAGS) + "\n")
        build_host(True, lib, os.path.join(vdir, "libvft_host.so"))
        return lib
    if force or needs_build():
        compile_and_link(LIB, OBJ_DIR, force)
    build_host(force)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
