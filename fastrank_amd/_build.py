"""Builds libfastrank_amd.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

The library is the product: every public call in this package goes through it.  There is no
Python/NumPy/CPU implementation to fall back to -- if the library is missing or no GPU is
visible, compute calls raise.
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libfastrank_amd.so")
SOURCES = ["device.hip", "capi.cpp"]
HEADERS = ["device.hpp", "host.hpp", "loader.hpp", "rf_train.hpp", "kernels_rf.inc", "json.hpp", os.path.join("..", "..", "include", "fastrank.h"),
           "device_plumbing.inc", "kernels_score.inc", "kernels_tree.inc", "kernels_treerank.inc", "kernels_metric.inc",
           "kernels_linesearch.inc", "kernels_verify.inc", "kernels_fullrank.inc", "kernels_rr.inc", "kernels_sortnet.inc", "kernels_fullverify.inc",
           "device_dataset.inc"]
# -ffp-contract=off is a correctness flag, not a tuning flag: the reference's dot product is an
# unfused f64 multiply-then-add (src/dense_dataset.rs:71-74) and rank order must be bit-exact.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build libfastrank_amd.so for gfx950)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.exists(p) and os.path.getmtime(p) > built for p in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB_PATH
    extra = os.environ.get("FR_BUILD_FLAGS", "").split()  # kernel-tuning experiments (-DFV_WAVES_64=2 ...)
    cmd = [hipcc_path()] + FLAGS + extra + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB_PATH + ".tmp", "-lz"]
    if verbose:
        print(" ".join(cmd))
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + proc.stdout)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
