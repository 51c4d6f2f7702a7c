#!/usr/bin/env python3
"""Experimental ring-of-four variant of the kNN kernel (W3D_KNN_MODE=8192; apply scripts/experiments/knn_ring4.patch first) against the default: same
results, time on the loop's own rows (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
g = torch.Generator().manual_seed(3)
for (N, M) in ((5000, 3000), (512 * 300 + 17, 9000)):
    X = torch.randn(N, 256, generator=g).to(torch.bfloat16).cuda(); Y = torch.randn(M, 256, generator=g).to(torch.bfloat16).cuda()
    ops.knn_set_tuning(mode=0); i0, d0 = ops.raw_knn_match(X, Y, 20)
    ops.knn_set_tuning(mode=8192); i1, d1 = ops.raw_knn_match(X, Y, 20)
    i2, d2 = ops.raw_knn_match(X, Y, 20, prev_idx=i0.clone())
    torch.cuda.synchronize()
    print(N, M, "cold equal", torch.equal(i0, i1) and torch.equal(d0, d1), "warm equal", torch.equal(i0, i2) and torch.equal(d0, d2), flush=True)
from wast3d_amd.style_loop import make_synthetic_c3
ops.knn_set_tuning(mode=0)
for style in ("matched", "sparse"):
    loop = make_synthetic_c3(N=1_000_000, M=65536, d=256, k=20, style_density=style)
    for _ in range(4): loop.step()
    torch.cuda.synchronize()
    FX, FY, k = loop.FX, loop.FY, loop.k
    prev = loop.nn_idx.clone(); idx = torch.empty_like(prev); dd = torch.empty_like(loop.nn_d2)
    ref = None
    for m in (0, 8192, 0, 8192):
        ops.knn_set_tuning(mode=int(m))
        w = ops.raw_knn_prepare(FX, FY, k, prev_idx=prev)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); ops.raw_knn_match_prepared(FX, FY, k, idx, dd, w); e1.record(); torch.cuda.synchronize()
        if ref is None: ref = idx.clone()
        print(f"{style} mode {m}: {e0.elapsed_time(e1):.2f} ms  equal to mode 0: {torch.equal(idx, ref)}", flush=True)
    ops.knn_set_tuning(mode=0)
