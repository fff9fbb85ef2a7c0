"""Builds ``libgendr_hip.so`` (the C-ABI HIP library) in-tree with hipcc for gfx950.

In-tree so that the built library travels with the source snapshot; nothing is
JIT-compiled at import time.  ``-ffp-contract=off`` is part of the parity
policy: the kernels reproduce the reference's operation order
without FMA contraction (a later ``-ffp-contract=`` of a variant overrides it).
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libgendr_hip.so")
# Build variants.  "default" is the shipped library.  "exact" compiles the backward kernels with
# -DGENDR_EXACT_GRADIENT=1: every gradient-side quotient, the gaussian / gamma densities and the t-conorm partials keep
# the reference's own rounding and promotions (kernel.cu:1026-1052) -- the parity build SURVEY H1 asks for; the tests
# and profiles/parity_r03.json compare the two against the oracle (tests/test_gpu_exact_gradient.py).
# "fast" (round 4; gendr_math.h GENDR_FAST_MATH): the reference's formulas, operation order, skip tests and culling, with the
# per-pair arithmetic at hardware accuracy (float reciprocals, v_sqrt_f32, 2^x-based exp, float instead of double
# sub-expressions) and FMA contraction ON -- what nvcc does to the reference by default (/root/reference/setup.py:10).
# Gated on the GPU by the spread of the reference's own two builds (tests/test_gpu_fast_variant.py); never the silent
# default: GENDR_VARIANT=fast or _native.use_variant('fast') select it, bench.py reports it under extra.fast_variant.
# ("on", not "fast": contraction is then a property of the source expression, so the compacted, dense and all-pairs paths of a kernel
# round alike and culled == all-pairs stays bit-exact; with "fast" the paths of one kernel were fused differently -- ADVICE r4)
VARIANTS = {"default": [], "exact": ["-DGENDR_EXACT_GRADIENT=1"], "fast": ["-DGENDR_FAST_MATH=1", "-ffp-contract=on"]}


def lib_path(variant="default"):
    if variant not in VARIANTS:
        raise ValueError("unknown build variant %r (have %s)" % (variant, sorted(VARIANTS)))
    return LIB_PATH if variant == "default" else os.path.join(_HERE, "libgendr_hip_%s.so" % variant)
SOURCES = ["gendr_capi.hip"]
HEADERS = ["gendr_kernels.h", "gendr_team.h", "gendr_math.h", "gendr_project.h", "gendr_voxel.h", "gendr_texture.h", "gendr_light.h", os.path.join("compat", "gendr_f64.h"), os.path.join("..", "..", "include", "gendr_hip.h")]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-munsafe-fp-atomics", "-fno-slp-vectorize"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libgendr_hip.so cannot be built")
    return exe


def needs_build(variant="default"):
    path = lib_path(variant)
    if not os.path.exists(path):
        return True
    built = os.path.getmtime(path)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > built for d in deps)


def build(force=False, verbose=False, variant="default"):
    """Compile the library (one build variant) if it is missing or older than its sources; returns its path.  Safe when
    several processes (one per GPU) call it at once: one compiles under a file lock into a temporary name and renames,
    the others find the library up to date when they get the lock."""
    path = lib_path(variant)
    if not (force or needs_build(variant)):
        return path
    import fcntl
    with open(path + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or needs_build(variant):
                tmp = "%s.%d.tmp" % (path, os.getpid())
                cmd = [_hipcc()] + HIPCC_FLAGS + VARIANTS[variant] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", tmp]
                if verbose:
                    print(" ".join(cmd[:-1] + [path]))
                try:
                    subprocess.check_call(cmd, cwd=CSRC)
                    os.replace(tmp, path)
                finally:
                    if os.path.exists(tmp):
                        os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return path


# ---- the C++ autograd node (csrc/gendr_torch.cpp): host-side plumbing over the C ABI, compiled with g++ against the installed PyTorch
TORCH_EXT_SRC = os.path.join(CSRC, "gendr_torch.cpp")
TORCH_EXT_PATH = os.path.join(_HERE, "_gendr_torch.so")


def torch_ext_needs_build():
    if not os.path.exists(TORCH_EXT_PATH):
        return True
    built = os.path.getmtime(TORCH_EXT_PATH)
    return any(os.path.getmtime(d) > built for d in (TORCH_EXT_SRC, os.path.join(CSRC, "..", "..", "include", "gendr_hip.h")))


def build_torch_ext(force=False, verbose=False):
    """gendr_amd/_gendr_torch.so: GenDRFunction as a torch::autograd::Function (two native calls per step, no Python frames).  Plain
    g++ -- the file holds no device code and uses PyTorch-ROCm's own c10::hip names, nothing is hipified; it binds the C-ABI entry
    points of libgendr_hip*.so at run time (addresses handed over by gendr_amd/_native.py), so it links none of them."""
    if not (force or torch_ext_needs_build()):
        return TORCH_EXT_PATH
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    import fcntl
    tl = os.path.join(os.path.dirname(torch.__file__), "lib")
    cxx = shutil.which("g++") or "g++"
    with open(TORCH_EXT_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or torch_ext_needs_build():
                tmp = "%s.%d.tmp" % (TORCH_EXT_PATH, os.getpid())
                cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
                       "-DTORCH_EXTENSION_NAME=_gendr_torch", "-DTORCH_API_INCLUDE_EXTENSION_H",
                       "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
                cmd += ["-I" + d for d in ce.include_paths()] + ["-I" + sysconfig.get_paths()["include"], "-I/opt/rocm/include"]   # (torch ships the pybind11 headers it was built with)
                cmd += [TORCH_EXT_SRC, "-o", tmp, "-L" + tl, "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-ltorch_python",
                        "-Wl,-rpath," + tl]
                if verbose:
                    print(" ".join(cmd))
                try:
                    subprocess.check_call(cmd)
                    os.replace(tmp, TORCH_EXT_PATH)
                finally:
                    if os.path.exists(tmp):
                        os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return TORCH_EXT_PATH


def source_sha():
    """Short hash of the kernel sources: stamps the profiler summaries under profiles/ so that bench.py only quotes
    counters that were collected on THESE kernels."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:12]


# What build_all() / __graft_entry__.build() compile: the shipped library and the parity artefact.  The `fast` variant answered
# its question in round 4 (+7 ... 10 % at BASELINE config 2: the rounding policy is not what keeps the path from the HBM roofline)
# and is built on request only (`python -m gendr_amd.build fast`); its tests and bench.py's extra.fast_variant skip without it.
DEFAULT_VARIANTS = ("default", "exact")


def build_all(force=False, verbose=False, variants=DEFAULT_VARIANTS):
    """The variants the suite needs, compiled side by side (each hipcc run is single-threaded)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(len(variants)) as ex:
        return list(ex.map(lambda v: build(force=force, verbose=verbose, variant=v), sorted(variants)))


if __name__ == "__main__":
    import sys
    names = tuple(a for a in sys.argv[1:] if a in VARIANTS) or DEFAULT_VARIANTS
    print(build_all(force="--force" in sys.argv or len(sys.argv) == 1, verbose=True, variants=names))
    print(build_torch_ext(force="--force" in sys.argv, verbose=True))
