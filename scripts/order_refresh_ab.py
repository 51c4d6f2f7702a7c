"""Cost and effect of refreshing the content rows' processing order during a run (dev tool): C3 loop, n0 iterations, then 20 timed steps; prints the refresh cost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd.style_loop import make_synthetic_c3, descriptor_tree_order
loop = make_synthetic_c3(style_density="matched", lists="exact", timers=True)
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 600
for i in range(n0):
    loop.step()
torch.cuda.synchronize()
for p in (16, "nn0", "nn0+d2", "nnmean", 16):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if p == "nn0":                                                   # by the position (in the style rows' tree order) of the row's nearest style row
        ro = torch.argsort(loop.nn_idx[:, 0].long(), stable=True).to(torch.int32).contiguous()
    elif p == "nn0+d2":                                              # the same, ties (rows sharing their nearest style row) by distance to it
        key = loop.nn_idx[:, 0].double() + (loop.nn_d2[:, 0] / loop.nn_d2[:, 0].max().clamp_min(1e-30)).double() * 0.999
        ro = torch.argsort(key).to(torch.int32).contiguous()
    elif p == "nnmean":                                              # by the mean position of the row's k nearest style rows
        ro = torch.argsort(loop.nn_idx.double().mean(1)).to(torch.int32).contiguous()
    else:
        ro = descriptor_tree_order(loop.FX, leaf=32, iters=4, proj_dim=p).to(torch.int32).contiguous()
    torch.cuda.synchronize(); t_build = (time.perf_counter() - t0) * 1e3
    loop.row_order = ro
    for _ in range(3):
        loop.step()
    loop.knn_events = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        loop.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10 * 1e3
    print(f"proj_dim={p}: build {t_build:.0f} ms; step {dt:.2f} ms  knn {loop.knn_ms():.2f}", flush=True)
