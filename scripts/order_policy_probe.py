"""The C3 loop under refresh policies of the content rows' processing order: swept fraction, sweep time and step time along a run.
usage: order_policy_probe.py [kind:refresh ...]   default tree:512 nn0:32 tile:32 tile:8"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd.style_loop import make_synthetic_c3, LoopConfig
pol = sys.argv[1:] or ["tree:512", "nn0:32", "tile:32", "tile:8"]
marks = (20, 60, 150, 300, 450, 600)
for p in pol:
    kind, ref = p.split(":")
    loop = make_synthetic_c3(style_density="matched", lists="exact", timers=True, config=LoopConfig(row_order_kind=kind, row_order_refresh=int(ref)))
    pairs = ((loop.N + 511) // 512) * ((loop.M + 63) // 64)
    it = 0
    torch.cuda.synchronize(); t_run = time.perf_counter()
    for m in marks:
        loop.knn_events = []; loop.swept_log = []
        torch.cuda.synchronize(); t0 = time.perf_counter(); i0 = it
        while it < m:
            loop.step(); it += 1
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3 / (it - i0)
        kms = loop.knn_ms_per_step(); sw = loop.swept_pairs_per_step()
        print(f"{p:9s} its {i0:3d}-{it:3d}: mean step {dt:6.2f} ms, mean knn {sum(kms) / len(kms):6.2f} ms, last knn {kms[-1]:6.2f} ms, last swept {sw[-1] / pairs:.4f}", flush=True)
    print(f"{p:9s} 600 iterations in {time.perf_counter() - t_run:.2f} s", flush=True)
    del loop; torch.cuda.empty_cache()
