"""Straggler buckets of the C3 loop (dev tool): pairs per plan, the buckets' spread, candidates the straggler pass emits.  usage: straggler_probe.py [T=16] [iterations=20,60,250]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
from wast3d_amd.style_loop import make_synthetic_c3, LoopConfig
T = int(sys.argv[1]) if len(sys.argv) > 1 else 16
its = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "20,60,250").split(",")]
loop = make_synthetic_c3(style_density="matched", lists="exact", timers=True, config=LoopConfig(straggler_rows=T))
N, M, d, k = loop.N, loop.M, loop.d, loop.k
it = 0
for s in its:
    while it < s:
        loop.step(); it += 1
    torch.cuda.synchronize()
    sc = ops.knn_straggler_counts(N, M, d, k, loop.kws).long()
    cnt = ops.knn_prune_counts(N, M, d, k, loop.kws).long()
    print(f"iteration {it}: kept (block, tile) pairs {int(cnt.sum())} ({float(cnt.sum()) / (cnt.numel() * sc.numel()):.4f}), straggler (row, tile) pairs {int(sc.sum())}, per tile mean {float(sc.float().mean()):.0f} "
          f"p50 {int(sc.median())} p99 {int(sc.float().quantile(0.99))} max {int(sc.max())} (capacity {int(ops.lib().w3d_knn_straggler_capacity(N, M))})", flush=True)
