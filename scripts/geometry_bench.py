#!/usr/bin/env python3
"""The reference's own workload shape (NB11 cell 32): one style patch of N = 4165 points fitted to S = 189 domain circles of
400-900 points, 2000/100-neighbour descriptor lists, k = 20 domain neighbours.  The notebook runs the 189 problems one after the
other (480 s for 189 x 1001 iterations on its GPU = 394 circle-iterations/s); here they are one batched loop."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd.geometry_loop import GeometryFit

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4165); ap.add_argument("--circles", type=int, default=189); ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--batch", type=int, default=64)
a = ap.parse_args()
g = torch.Generator().manual_seed(0)
uv = torch.rand(a.n, 2, generator=g) * 2 - 1
T = torch.stack([uv[:, 0], uv[:, 1], 0.3 * torch.sin(2.5 * uv[:, 0]) * torch.cos(1.5 * uv[:, 1])], 1).mul(0.2).cuda()
doms = [(torch.randn(int(m), 3, generator=g) * 0.3 + torch.randn(3, generator=g)).cuda() for m in torch.randint(400, 900, (a.circles,), generator=g)]
tot_t = 0.0
for i in range(0, a.circles, a.batch):
    fit = GeometryFit(T, doms[i:i + a.batch])
    fit.step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        fit.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tot_t += dt
    print(f"batch of {fit.S} circles: {dt / a.iters * 1e3:.3f} ms/iter = {fit.S * a.iters / dt:.0f} circle-iters/s", flush=True)
print(f"TOTAL {a.circles} circles x {a.iters} iters: {tot_t:.2f} s  -> {a.circles * a.iters / tot_t:.0f} circle-iterations/s (reference notebook output: 394)")
