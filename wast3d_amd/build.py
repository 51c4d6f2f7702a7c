"""Builds libwast3d_hip.so (all HIP kernels + the C ABI) in-tree with hipcc for gfx950.

No torch / pybind involvement: the library's interface is include/wast3d_hip.h (plain C).
hipcc cross-compiles without a GPU, so this runs in the build container and on the GPU box alike.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libwast3d_hip.so")
OBJ = os.path.join(HERE, "_obj")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result", "-ffp-contract=on",
         "-I/opt/rocm/include", "-DROCPRIM_NO_DEPRECATED_WARNINGS"]


# per-file additions.  knn_mfma.hip: scores are compared with fmaxf trees; without "no NaNs" every fmaxf input gets a canonicalising
# v_max_f32 x, x (15 instead of 9 VALU per 32x32 block in the selection pretest).  Infinities stay honoured (they are the sentinels).
EXTRA_FLAGS = {"knn_mfma.hip": ["-fno-honor-nans"]}


# knn_mfma.hip is compiled into a main object (entry points, dispatch) and one leaf object per (operand format, feature width): the leaves hold the kernel
# instantiations of one shape each and compile in parallel (a single translation unit took six minutes)
KNN_LEAVES = [(f, d) for f in (0, 1) for d in (64, 128, 256)] + [(2, 128), (2, 256)]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def units():
    """(source, object file, extra flags) of every object of the library."""
    out = []
    for src in sources():
        base = os.path.basename(src)
        extra = EXTRA_FLAGS.get(base, [])
        out.append((src, os.path.join(OBJ, base[:-4] + ".o"), extra))
        if base == "knn_mfma.hip":
            for f, d in KNN_LEAVES:
                out.append((src, os.path.join(OBJ, f"knn_mfma_f{f}_d{d}.o"), extra + [f"-DW3D_KNN_FMT={f}", f"-DW3D_KNN_D={d}"]))
    return out


def _toolchain_id() -> str:
    try:
        return subprocess.run(["hipcc", "--version"], capture_output=True, text=True).stdout
    except OSError as e:
        raise RuntimeError("hipcc not found: wast3d_amd has no fallback path without its HIP library") from e


def _key(src, hdrs, flags, tool) -> str:
    """Content key of one object file: its source, every header it can include, the flags and the compiler build.  (mtimes do not survive
    the copy to a GPU box and say nothing about a header edited back; the key does.)"""
    h = hashlib.sha256()
    for f in [src] + sorted(hdrs):
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read() + b"\0")
    h.update(" ".join(flags).encode() + b"\0" + tool.encode())
    return h.hexdigest()


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "wast3d_hip.h")]
    tool = _toolchain_id()
    jobs, keys = [], {}
    for src, obj, extra in units():
        keys[obj] = _key(src, hdrs, FLAGS + extra, tool)
        if force or not os.path.exists(obj) or _read(obj + ".key") != keys[obj]:
            jobs.append((src, obj, extra))
    jobs.sort(key=lambda j: 0 if "knn_mfma_f" in j[1] else 1)           # the long compiles first

    def cc(job):
        src, obj, extra = job
        cmd = ["hipcc", *FLAGS, *extra, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
        with open(obj + ".key", "w") as f:
            f.write(keys[obj])
        return src

    if jobs:
        with ThreadPoolExecutor(max_workers=min(max(2, (os.cpu_count() or 4) - 1), len(jobs))) as ex:
            for j, s in zip(jobs, ex.map(cc, jobs)):
                if verbose:
                    print(f"[wast3d_amd.build] compiled {os.path.basename(j[1])}", file=sys.stderr)
    objs = [u[1] for u in units()]
    libkey = hashlib.sha256("".join(keys[o] for o in objs).encode()).hexdigest()
    if force or jobs or not os.path.exists(LIB) or _read(LIB + ".key") != libkey:
        cmd = ["hipcc", "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        with open(LIB + ".key", "w") as f:
            f.write(libkey)
        if verbose:
            print(f"[wast3d_amd.build] linked {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
