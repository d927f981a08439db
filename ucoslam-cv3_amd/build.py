"""In-tree build of the HIP hot path (gfx950) and of the test oracle.

`build_all()` is what `__graft_entry__.build()` runs:
  * csrc/*.hip            -> ucoslam-cv3_amd/libucoslam_hip.so   (hipcc --offload-arch=gfx950, the product)
  * oracle/*.cpp          -> oracle/liboracle.so                  (g++, test infrastructure)
  * /root/reference/...   -> oracle/_ref/*.so                     (only where the reference tree exists)
hipcc cross-compiles without a GPU. Objects are cached per source by mtime.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
OBJ_DIR = os.path.join(PKG_DIR, "build")
LIB_PATH = os.path.join(PKG_DIR, "libucoslam_hip.so")
ORACLE_DIR = os.path.join(REPO, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "liboracle.so")

ARCH = "gfx950"
# -ffp-contract=off: the ORB stage reproduces the reference's un-contracted IEEE float math bit for bit
HIPCC_FLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
    "-Wall", "-Wno-unused-function", "-Wno-unused-result",
]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP hot path cannot be built (there is no CPU fallback)")


def _newer(src_list, target) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def _run(cmd, **kw):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, **kw)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("command failed: " + " ".join(cmd))
    return r.stdout


def source_digest() -> str:
    """SHA-256 over every source the HIP library is built from (csrc/ + include/): stored beside the library by build_hip() so that a
    test session can tell a stale .so from a current one without trusting file times (snapshots do not preserve them)."""
    import hashlib

    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp", ".h"))]
    inc = os.path.join(REPO, "include")
    for root, _, names in sorted(os.walk(inc)):
        files += [os.path.join(root, f) for f in sorted(names) if f.endswith((".h", ".hpp", ".inc"))]
    for f in files:
        h.update(os.path.relpath(f, REPO).encode())
        h.update(open(f, "rb").read())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()


def library_is_current() -> bool:
    stamp = LIB_PATH + ".srchash"
    return os.path.exists(LIB_PATH) and os.path.exists(stamp) and open(stamp).read().strip() == source_digest()


def build_hip(verbose: bool = False) -> str:
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    hdrs += [os.path.join(REPO, "include", f) for f in os.listdir(os.path.join(REPO, "include")) if f.endswith(".h")]
    objs, jobs = [], []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ_DIR, s[:-4] + ".o")
        objs.append(obj)
        if _newer([src] + hdrs, obj):
            jobs.append([hipcc, *HIPCC_FLAGS, "-c", src, "-o", obj])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for out in ex.map(_run, jobs):
                if verbose and out:
                    print(out)
    if jobs or _newer(objs, LIB_PATH):
        _run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH, *objs])
    with open(LIB_PATH + ".srchash", "w") as f:
        f.write(source_digest() + "\n")
    return LIB_PATH


def build_oracle() -> str:
    """Compile the CPU restatement and, where /root/reference exists, the real-reference builds."""
    _run(["make", "-C", ORACLE_DIR, "-s", "liboracle.so"])
    _run(["make", "-C", ORACLE_DIR, "-s", "ref"])
    return ORACLE_LIB


def build_all(verbose: bool = False):
    lib = build_hip(verbose)
    build_oracle()
    return lib


if __name__ == "__main__":
    print(build_all(verbose=True))
