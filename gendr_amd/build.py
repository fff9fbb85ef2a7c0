"""Builds ``libgendr_hip.so`` (the C-ABI HIP library) in-tree with hipcc for gfx950.

In-tree so that the built library travels with the source snapshot; nothing is
JIT-compiled at import time.  ``-ffp-contract=off`` is part of the parity
policy: the kernels reproduce the reference's operation order
without FMA contraction.
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libgendr_hip.so")
SOURCES = ["gendr_capi.hip"]
HEADERS = ["gendr_kernels.h", "gendr_math.h", "gendr_project.h", "gendr_voxel.h", "gendr_texture.h", "gendr_light.h", "gendr_f64.h", os.path.join("..", "..", "include", "gendr_hip.h")]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-munsafe-fp-atomics", "-fno-slp-vectorize"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libgendr_hip.so cannot be built")
    return exe


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > built for d in deps)


def build(force=False, verbose=False):
    """Compile the library if it is missing or older than its sources; returns its path.  Safe when several
    processes (one per GPU) call it at once: one compiles under a file lock into a temporary name and renames, the
    others find the library up to date when they get the lock."""
    if not (force or needs_build()):
        return LIB_PATH
    import fcntl
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or needs_build():
                tmp = "%s.%d.tmp" % (LIB_PATH, os.getpid())
                cmd = [_hipcc()] + HIPCC_FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", tmp]
                if verbose:
                    print(" ".join(cmd[:-1] + [LIB_PATH]))
                try:
                    subprocess.check_call(cmd, cwd=CSRC)
                    os.replace(tmp, LIB_PATH)
                finally:
                    if os.path.exists(tmp):
                        os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
