#!/usr/bin/env python3
"""Micro-benchmark of the fused distance-GEMM + top-k kernel alone (dev tool; prints TFLOP/s).
One process, several configurations: --cfg n,m,d,k,mode[,dist] ..."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--cfg", nargs="+", default=["262144,65536,256,20,0", "262144,65536,256,20,1"])
ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
for cfg in a.cfg:
    f = cfg.split(",")
    n, m, d, k, mode = map(int, f[:5]); dist = f[5] if len(f) > 5 else "randn"
    ops.knn_set_tuning(mode=int(mode))
    g = torch.Generator().manual_seed(0)
    if dist == "randn":
        X = torch.randn(n, d, generator=g).to(torch.bfloat16).cuda(); Y = torch.randn(m, d, generator=g).to(torch.bfloat16).cuda()
    else:
        X = (torch.rand(n, d, generator=g) * 3).to(torch.bfloat16).cuda(); Y = (torch.rand(m, d, generator=g) * 3).to(torch.bfloat16).cuda()
    idx = torch.empty(n, k, dtype=torch.int32, device="cuda"); d2 = torch.empty(n, k, device="cuda")
    ops.raw_knn_match(X, Y, k, idx, d2); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        ops.raw_knn_match(X, Y, k, idx, d2)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print(f"N={n} M={m} d={d} k={k} mode={mode} {dist}: {ms:.3f} ms  {2.0*n*m*d/ms/1e9:.1f} TFLOP/s", flush=True)
    if mode in (0, 32):
        prev = idx.clone(); idx2 = torch.empty_like(idx)
        for tag, Xw in (("warm(same X)", X), ("warm(X+2% noise)", (X.float() * (1 + 0.02 * torch.randn(n, d, device="cuda"))).to(torch.bfloat16))):
            ops.raw_knn_match(Xw, Y, k, idx2, d2, prev_idx=prev); torch.cuda.synchronize()
            e0.record()
            for _ in range(a.iters):
                ops.raw_knn_match(Xw, Y, k, idx2, d2, prev_idx=prev)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            print(f"    {tag}: {ms:.3f} ms  {2.0*n*m*d/ms/1e9:.1f} TFLOP/s", flush=True)
