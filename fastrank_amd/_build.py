"""Builds libfastrank_amd.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

The library is the product: every public call in this package goes through it.  There is no
Python/NumPy/CPU implementation to fall back to -- if the library is missing or no GPU is
visible, compute calls raise.

The objects are compiled in parallel (fullverify.hip once per slice of the size classes of the full-ranking kernel:
-DFV_PART=k) and cached under fastrank_amd/build/ by a digest of their sources, so editing one kernel family rebuilds one
or a few objects.
"""
import concurrent.futures
import fcntl
import hashlib
import os
import re
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OBJ_DIR = os.path.join(_HERE, "build")
LIB_PATH = os.path.join(_HERE, "libfastrank_amd.so")
INCLUDE = os.path.join("..", "..", "include", "fastrank.h")
DEVICE_INCS = ["device.hpp", "fullverify.hpp", "device_plumbing.inc", "kernels_score.inc", "kernels_tree.inc", "kernels_treerank.inc", "kernels_metric.inc",
               "kernels_linesearch.inc", "kernels_chain.inc", "kernels_fillnet.inc", "kernels_order.inc", "kernels_verify.inc", "kernels_fullrank.inc", "kernels_rr.inc", "kernels_rf.inc", "device_dataset.inc", "rccl_exchange.inc"]
FV_INCS = ["device.hpp", "fullverify.hpp", "kernels_sortnet.inc", "kernels_fullverify.inc"]
HOST_INCS = ["device.hpp", "host.hpp", "loader.hpp", "rf_train.hpp", "json.hpp", INCLUDE]


def fv_parts() -> int:
    txt = open(os.path.join(CSRC, "fullverify.hpp")).read()
    return int(re.search(r"FV_PARTS\s*=\s*(\d+)", txt).group(1))


def units():
    """(object name, source, extra flags, dependencies)"""
    out = [("device.o", "device.hip", [], DEVICE_INCS), ("capi.o", "capi.cpp", [], HOST_INCS)]
    for k in range(fv_parts()):
        out.append(("fullverify_%d.o" % k, "fullverify.hip", ["-DFV_PART=%d" % k], FV_INCS))
    return out


# -ffp-contract=off is a correctness flag, not a tuning flag: the reference's dot product is an
# unfused f64 multiply-then-add (src/dense_dataset.rs:71-74) and rank order must be bit-exact.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build libfastrank_amd.so for gfx950)")


def _extra_flags():
    return os.environ.get("FR_BUILD_FLAGS", "").split()  # kernel-tuning experiments (-DFV_... ...)


def _digest(src, extra, deps) -> str:
    """What an object (or, over all units, the library) was built from: the contents of its source and of everything it
    includes, the compiler flags.  Staleness is judged by this, not by time stamps -- a tree that was copied (to the GPU
    box) or checked out keeps whatever mtimes the copy gave it."""
    h = hashlib.sha1()
    h.update(" ".join(FLAGS + _extra_flags() + list(extra)).encode())
    for name in [src] + list(deps):
        path = os.path.join(CSRC, name)
        h.update(b"\0" + name.encode() + b"\0")
        if os.path.exists(path):
            with open(path, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()


def _lib_digest() -> str:
    h = hashlib.sha1()
    for obj, src, extra, deps in units():
        h.update((obj + ":" + _digest(src, extra, deps) + "\n").encode())
    return h.hexdigest()


def _stale(obj, src, extra, deps) -> bool:
    path = os.path.join(OBJ_DIR, obj)
    flag_file = path + ".flags"
    return not (os.path.exists(path) and os.path.exists(flag_file) and open(flag_file).read() == _digest(src, extra, deps))


def needs_build() -> bool:
    """The library is current if it was built from exactly these sources and flags (libfastrank_amd.so.flags holds their
    digest; the objects under fastrank_amd/build/ are only a cache: a tree that travelled without them, e.g. to the GPU
    box, is not rebuilt)."""
    flag_file = LIB_PATH + ".flags"
    return not (os.path.exists(LIB_PATH) and os.path.exists(flag_file) and open(flag_file).read() == _lib_digest())


def _compile(hipcc, obj, src, extra, deps, verbose):
    path = os.path.join(OBJ_DIR, obj)
    digest = _digest(src, extra, deps)  # (of what is compiled now: taken before the compiler reads the files)
    tmp = "%s.%d.tmp" % (path, os.getpid())
    cmd = [hipcc] + FLAGS + _extra_flags() + extra + ["-c", os.path.join(CSRC, src), "-o", tmp]
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed on %s %s:\n%s" % (src, " ".join(extra), proc.stdout))
    os.replace(tmp, path)
    with open(path + ".flags", "w") as fh:
        fh.write(digest)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    # one builder at a time per tree: processes that import the package together on a fresh tree (the ranks of a
    # torch.distributed.run launch) queue up here, and all but the first find the library current when their turn comes
    with open(os.path.join(OBJ_DIR, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return LIB_PATH
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force: bool, verbose: bool) -> str:
    hipcc = hipcc_path()
    lib_digest = _lib_digest()
    todo = [(o, s, e, d) for o, s, e, d in units() if force or _stale(o, s, e, d)]
    jobs = max(1, min(len(todo), int(os.environ.get("FR_BUILD_JOBS", str(os.cpu_count() or 4)))))
    if todo:
        with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as pool:
            futs = [pool.submit(_compile, hipcc, o, s, e, d, verbose) for o, s, e, d in todo]
            for f in futs:
                f.result()
    objs = [os.path.join(OBJ_DIR, o) for o, _, _, _ in units()]
    tmp = "%s.%d.tmp" % (LIB_PATH, os.getpid())
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp, "-lz", "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("link failed:\n" + proc.stdout)
    os.replace(tmp, LIB_PATH)
    with open(LIB_PATH + ".flags", "w") as fh:
        fh.write(lib_digest)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
