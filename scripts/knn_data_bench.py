#!/usr/bin/env python3
"""The fused kNN kernel alone on the style loop's own feature rows after a few iterations (dev tool): W3D_STYLE_DENSITY=sparse|matched picks the
style patch, argv = list of W3D_KNN_MODE values (debug bits need a -DW3D_KNN_DEBUG library via W3D_LIB_PATH).  Warm start = previous neighbours."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
from wast3d_amd.style_loop import make_synthetic_c3
modes = [int(a) for a in sys.argv[1:]] or [0]
ops.knn_set_tuning(mode=0)
loop = make_synthetic_c3(N=1_000_000, M=65536, d=256, k=20)
for _ in range(4): loop.step()
torch.cuda.synchronize()
FX, FY, k = loop.FX, loop.FY, loop.k
prev = loop.nn_idx.clone()
idx = torch.empty_like(prev); d2 = torch.empty_like(loop.nn_d2)
N, M, d = FX.shape[0], FY.shape[0], FX.shape[1]
for warm in ("prev", "cold"):
    for m in modes:
        ops.knn_set_tuning(mode=int(m))
        ts = []
        for it in range(3):
            w = ops.raw_knn_prepare(FX, FY, k, prev_idx=prev if warm == "prev" else None)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); ops.raw_knn_match_prepared(FX, FY, k, idx, d2, w); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = min(ts)
        print(f"{os.environ.get('W3D_STYLE_DENSITY', 'sparse')} warm={warm} mode={m}: {ms:.2f} ms  {2.0 * N * M * d / ms / 1e9:.0f} TFLOP/s", flush=True)
