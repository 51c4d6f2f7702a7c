"""The C3 loop at several stations of a run: the plan's swept fraction of (row block, style tile) pairs against the pairs that hold a true neighbour of the block (what ANY
exact test must keep), and the spread of a block's neighbours over tiles.  usage: need_trace.py [stations=20,100,250,400,600]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
from wast3d_amd.style_loop import make_synthetic_c3
st = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "20,100,250,400,600").split(",")]
loop = make_synthetic_c3(style_density="matched", lists="exact", timers=True)
N, M, d, k = loop.N, loop.M, loop.d, loop.k
nblk, T = (N + 511) // 512, (M + 63) // 64
Nc = N // 512 * 512
it = 0
for s in st:
    while it < s:
        loop.step(); it += 1
    torch.cuda.synchronize()
    ro = loop.row_order.long()[:Nc]
    plan = float(ops.knn_prune_counts(N, M, d, k, loop.kws).sum()) / (nblk * T)
    tiles = (loop.nn_idx.long()[ro] >> 6)
    blk = torch.arange(Nc, device="cuda")[:, None].expand(-1, k) // 512
    need = torch.zeros(Nc // 512, T, dtype=torch.bool, device="cuda")
    need[blk.reshape(-1), tiles.reshape(-1)] = True
    # per 64-row wave group: tiles needed
    wv = torch.arange(Nc, device="cuda")[:, None].expand(-1, k) // 64
    needw = torch.zeros(Nc // 64, T, dtype=torch.bool, device="cuda")
    needw[wv.reshape(-1), tiles.reshape(-1)] = True
    mot = float(loop.motion2_q.float().sqrt().mean()) if hasattr(loop, "motion2_q") and loop.motion2_q is not None else float("nan")
    kth = float(loop.nn_d2[:, -1].clamp_min(0).sqrt().mean())
    print(f"iteration {it:4d}: plan keeps {plan:.4f}; pairs with a true neighbour {float(need.float().mean()):.4f} (per 64-row group {float(needw.float().mean()):.4f}); "
          f"mean k-th distance {kth:.4f}, mean row motion per iteration {mot:.5f}", flush=True)
