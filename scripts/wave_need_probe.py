"""How many of a row block's 8 waves (64 rows each) need a style tile the plan keeps?  The sweep runs every kept tile on all 8 waves; the plan's level 2 knows, per row, whether
the cached bound still rules the tile out.  On the C3 loop's own states: per kept (block, tile) pair the number of waves with a row whose bound does NOT rule it out, as
the plan saw it (the bounds before the sweep), and a cost model — waves w and w + 4 share a SIMD, a tile costs as much as the busiest SIMD.
usage: wave_need_probe.py [iterations=40,280] [blocks=64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
from wast3d_amd._lib import lib
from wast3d_amd.style_loop import make_synthetic_c3
st = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "40,280").split(",")]
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 64
loop = make_synthetic_c3(style_density="matched", lists="exact", timers=True)      # (LoopConfig defaults: straggler pairs on — the kept tiles are what the blocks still sweep)
N, M, d, k = loop.N, loop.M, loop.d, loop.k
T, nblk = (M + 63) // 64, (N + 511) // 512
Np = (N + 63) // 64 * 64
it = 0
for s in st:
    while it < s - 1:
        loop.step(); it += 1
    torch.cuda.synchronize()
    old_lb = loop.bounds.lb.clone()
    loop.step(); it += 1
    torch.cuda.synchronize()
    w = loop.kws.t if hasattr(loop.kws, "t") else loop.kws
    wf = w.view(torch.uint8)[: 2 * Np * 4].view(torch.float32)
    xnorm, thr = wf[:N], wf[Np:Np + N]
    off = int(lib().w3d_knn_prune_counts_offset(N, M, d, k))
    cnt = w.view(torch.uint8)[off: off + nblk * 4].view(torch.int32)
    tord = w.view(torch.uint8)[off - nblk * T * 4: off].view(torch.int32).view(nblk, T)
    cmax = float(loop.prune.cnorm_max)
    dx = (xnorm - 2 * thr).clamp_min(0).add(4e-6 * (xnorm + 2 * cmax)).sqrt()
    q = (loop.bounds.cum + dx) * (1 + 1e-6) + 1e-7
    ro = loop.row_order.long()
    hist = torch.zeros(9); simd_cost = 0.0; kept = 0; true_hist = torch.zeros(9); row_hist = torch.zeros(10); row_pairs = torch.zeros(10)
    for b in torch.linspace(0, nblk - 2, nb).long().tolist():
        c = int(cnt[b]); tl = tord[b, :c].long()
        rows = ro[b * 512:(b + 1) * 512]
        need = old_lb[b, tl].float() <= q[rows][None, :]                 # [c, 512] as the plan saw it
        wn = need.view(c, 8, 64).any(2)                                  # [c, 8]
        nw = wn.sum(1)
        hist += torch.bincount(nw.cpu(), minlength=9)[:9].float()
        simd = (wn[:, :4].int() + wn[:, 4:].int()).max(1).values          # busiest SIMD: waves w and w + 4
        simd_cost += float(simd.sum()); kept += c
        rn = need.sum(1)                                                 # rows of the block whose bound does not rule the tile out
        edges = torch.tensor([0, 1, 2, 4, 8, 16, 32, 64, 512], device=rn.device)
        rb = torch.bucketize(rn, edges, right=False)                     # bucket i: edges[i-1] < rn <= edges[i]
        row_hist += torch.bincount(rb.cpu(), minlength=10)[:10].float(); row_pairs += torch.bincount(rb.cpu(), weights=rn.float().cpu(), minlength=10)[:10]
        tneed = loop.bounds.lb[b, tl].float() <= q[rows][None, :]        # with the bounds the sweep has just recorded: rows that really have something within reach
        true_hist += torch.bincount(tneed.view(c, 8, 64).any(2).sum(1).cpu(), minlength=9)[:9].float()
    print(f"iteration {it}: swept {float(cnt.sum()) / (nblk * T):.4f}; waves needing a kept tile (plan's view) mean {float((hist * torch.arange(9)).sum() / hist.sum()):.2f} of 8, "
          f"histogram 0..8 {[round(float(x), 3) for x in hist / hist.sum()]}; cost model (busiest SIMD, 2 = today) {simd_cost / kept:.3f}; "
          f"ROWS needing a kept tile, buckets (0, 1, 2, <=4, <=8, <=16, <=32, <=64, <=512): share of kept pairs {[round(float(x), 3) for x in (row_hist / row_hist.sum())[:9]]}, "
          f"(row, tile) pairs in each bucket per kept pair {[round(float(x), 2) for x in (row_pairs / row_hist.sum())[:9]]}; "
          f"waves with a row truly within reach mean {float((true_hist * torch.arange(9)).sum() / true_hist.sum()):.2f}, none at all {float(true_hist[0] / true_hist.sum()):.3f}", flush=True)
