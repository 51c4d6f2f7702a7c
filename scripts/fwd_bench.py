#!/usr/bin/env python3
"""Micro-benchmark at the bench shape: featurisation, gather kernel vs LDS-window kernel (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
from wast3d_amd.style_loop import morton_order, windowed_knn_lists
N, d = 1_000_000, 256
g = torch.Generator().manual_seed(0)
X = (torch.rand(N, 3, generator=g) * 2.6 - 1.3).cuda()
X = X[morton_order(X)].contiguous()
idx = windowed_knn_lists(X, d + 1)
X4 = ops.raw_pack_xyz4(X)
F = torch.zeros(N, d, dtype=torch.bfloat16, device="cuda"); m2 = torch.empty(N, device="cuda"); xn = torch.empty(N, device="cuda")
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("gather kernel ms:", t(lambda: ops.raw_descriptor_fwd_x4_motion(X4, idx, F, m2, xn)))
print("window kernel ms:", t(lambda: ops.raw_descriptor_fwd_window(X4, idx, F, m2, xn)))
