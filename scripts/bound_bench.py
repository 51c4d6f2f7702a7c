#!/usr/bin/env python3
"""The warm-start bound pass alone (previous neighbours' rows gathered), on the loop's own rows (dev tool; W3D_LIB_PATH selects the library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
from wast3d_amd.style_loop import make_synthetic_c3
loop = make_synthetic_c3(N=1_000_000, M=65536, d=256, k=20, style_density="matched")
for _ in range(3): loop.step()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
ops.raw_knn_prepare(loop.FX, loop.FY, loop.k, prev_idx=loop.nn_idx); torch.cuda.synchronize()
e0.record()
for _ in range(10): ops.raw_knn_prepare(loop.FX, loop.FY, loop.k, prev_idx=loop.nn_idx)
e1.record(); torch.cuda.synchronize()
print(f"prepare with previous neighbours: {e0.elapsed_time(e1) / 10:.3f} ms")
