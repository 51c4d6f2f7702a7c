"""How fast does the content rows' processing order age, and what does a cheap refresh buy?  After n0 iterations of the C3 loop (order built at iteration 1, the
default until round 3): swept fraction of the prune plan and time of the pruned sweep under (a) the aged order, (b) a fresh 16-PC tree, (c) rows sorted by the
position of their nearest style row ("nn0"), (d) by the mean position of their k neighbours, (e) by the tile-median.  Also the cost of building each.
usage: order_probe.py [n0=300]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
from wast3d_amd.style_loop import make_synthetic_c3, descriptor_tree_order, LoopConfig
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 300
loop = make_synthetic_c3(style_density="matched", lists="exact", timers=True, config=LoopConfig(row_order_refresh=0))
N, M, d, k = loop.N, loop.M, loop.d, loop.k
pairs = ((N + 511) // 512) * ((M + 63) // 64)
idx = torch.empty_like(loop.nn_idx); d2 = torch.empty_like(loop.nn_d2)

def measure(order, tag):
    ev = []
    for _ in range(3):
        ops.raw_knn_prepare(loop.FX, loop.FY, k, prev_idx=loop.nn_idx, ws=loop.kws, row_order=order)
        ops.raw_knn_prune_plan(loop.FX, k, loop.prune, loop.kws, row_order=order)
        ev.append(ops.raw_knn_match_prepared(loop.FX, loop.FY, k, idx, d2, 3, timed=True, ws=loop.kws, row_order=order))
    torch.cuda.synchronize()
    sw = float(ops.knn_prune_counts(N, M, d, k, loop.kws).sum()) / pairs
    ms = sorted(a.elapsed_time(b) for a, b in ev)[1]
    assert torch.equal(idx, loop.nn_idx)
    print(f"  {tag:34s} swept {sw:.4f}  pruned sweep {ms:6.2f} ms", flush=True)

def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize()
    return r, (time.perf_counter() - t0) * 1e3

marks = [20, n0] if n0 > 20 else [n0]
it = 0
for m in marks:
    while it < m:
        loop.step(); it += 1
    torch.cuda.synchronize()
    print(f"after {it} iterations (order built after iteration 1):")
    measure(loop.row_order, "aged order")
    o, t = timed(lambda: descriptor_tree_order(loop.FX, leaf=32, iters=4, proj_dim=16).to(torch.int32).contiguous()); measure(o, f"fresh tree (16 PCs), {t:.1f} ms")
    o, t = timed(lambda: torch.argsort(loop.nn_idx[:, 0], stable=True).to(torch.int32).contiguous()); measure(o, f"nn0 (nearest style row), {t:.2f} ms")
    o, t = timed(lambda: torch.argsort(loop.nn_idx.float().mean(1)).to(torch.int32).contiguous()); measure(o, f"mean neighbour position, {t:.2f} ms")
    o, t = timed(lambda: torch.argsort(loop.nn_idx.median(1).values, stable=True).to(torch.int32).contiguous()); measure(o, f"median neighbour position, {t:.2f} ms")
    # two-level: tile of the nearest neighbour, then mean position (rows of one tile stay together, ordered by where the rest of their neighbours are)
    o, t = timed(lambda: torch.argsort((loop.nn_idx[:, 0] // 64).double() * 1e6 + loop.nn_idx.double().mean(1)).to(torch.int32).contiguous()); measure(o, f"nn0 tile, then mean position, {t:.2f} ms")
