"""Builds libwast3d_hip.so (all HIP kernels + the C ABI) in-tree with hipcc for gfx950.

No torch / pybind involvement: the library's interface is include/wast3d_hip.h (plain C).
hipcc cross-compiles without a GPU, so this runs in the build container and on the GPU box alike.
"""
from __future__ import annotations

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


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "wast3d_hip.h")]
    jobs = []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = ["hipcc", *FLAGS, *EXTRA_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
        return src

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for s in ex.map(cc, jobs):
                if verbose:
                    print(f"[wast3d_amd.build] compiled {os.path.basename(s)}", file=sys.stderr)
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + ".o") for s in sources()]
    if force or jobs or _stale(LIB, objs):
        cmd = ["hipcc", "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"[wast3d_amd.build] linked {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
