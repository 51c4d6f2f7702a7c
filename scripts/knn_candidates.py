#!/usr/bin/env python3
"""How many style rows pass the warm-start bound of the fused kNN, per content row (dev tool, brute force on a sample of rows).
W3D_STYLE_DENSITY=sparse|matched picks the synthetic style patch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd.style_loop import make_synthetic_c3
loop = make_synthetic_c3(N=1_000_000, M=65536, d=256, k=20)
for _ in range(4): loop.step()
torch.cuda.synchronize()
sel = torch.arange(0, loop.N, loop.N // 4096, device="cuda")[:4096]
prev_idx = loop.nn_idx[sel].long().clone(); prev_kth = loop.nn_d2[sel, -1].clone()
FX_old = loop.FX[sel].float().clone()
loop.step(); torch.cuda.synchronize()
FX = loop.FX[sel].float(); FY = loop.FY.float()
D2 = (FX * FX).sum(1, keepdim=True) + (FY * FY).sum(1)[None, :] - 2 * FX @ FY.t()
kth = D2.topk(20, dim=1, largest=False).values[:, -1]
motion = (FX - FX_old).norm(dim=1)
b_motion = (prev_kth.clamp_min(0).sqrt() + motion) ** 2
b_prev = D2.gather(1, prev_idx).max(1).values
print("k-th distance^2 (mean)", kth.mean().item(), " row motion |dF| (mean)", motion.mean().item(), " sqrt(kth) mean", kth.clamp_min(0).sqrt().mean().item())
for name, b in (("exact k-th", kth), ("prev-neighbour bound", b_prev), ("motion bound", b_motion)):
    cnt = (D2 <= b[:, None]).sum(1).float()
    print(f"{name:22s}: candidates per row mean {cnt.mean().item():9.1f}  median {cnt.median().item():7.0f}  max {cnt.max().item():7.0f}   bound/kth {((b / kth).mean().item()):.3f}")
same = (loop.nn_idx[sel].long().sort(1).values == prev_idx.sort(1).values).float().mean().item()
print("neighbour lists unchanged since last iteration (entry-wise):", same)
