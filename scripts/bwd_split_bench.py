"""Fused NN-loss backward vs the split form (gF producer in the sweep's row order + window scatter with gF given), on the C3 loop's own state (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
from wast3d_amd.style_loop import make_synthetic_c3

def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

loop = make_synthetic_c3(style_density="matched", lists=os.environ.get("LISTS", "exact"))
for _ in range(int(os.environ.get("STEPS", "4"))):
    loop.step()
torch.cuda.synchronize()
N, M, k = loop.N, loop.FY.shape[0], loop.k
scale = 1.0 / (N * M)
gx = torch.zeros(N, 3, device="cuda"); gF = torch.empty(N, 256, device="cuda")
fya = loop._fy_absmax()
print("far fraction", loop.far.far_fraction if loop.far is not None else 0)
print("fused (far plan)            ms:", t(lambda: ops.raw_knn_descriptor_bwd_fused(loop.xyz4, loop.idx, loop.FX, loop.FY, fya, loop.nn_idx, scale, gx, loop.max_fanin, far=loop.far)))
print("gF producer, row order      ms:", t(lambda: ops.raw_knn_match_bwd(loop.FX, loop.FY, loop.nn_idx, scale, gF, False, row_order=loop.row_order)))
print("gF producer, natural order  ms:", t(lambda: ops.raw_knn_match_bwd(loop.FX, loop.FY, loop.nn_idx, scale, gF, False)))
print("window scatter (gF given, far -> global atomics) ms:", t(lambda: ops.raw_descriptor_bwd_window(loop.xyz4, loop.idx, gF, gx, loop.max_fanin)))

# the same fused launch right after a kNN sweep (what the loop does): is the stand-alone figure what the loop sees?
idx2 = torch.empty_like(loop.nn_idx); d22 = torch.empty_like(loop.nn_d2)
def after_knn(gap_fn):
    ts = []
    for _ in range(4):
        w = ops.raw_knn_prepare(loop.FX, loop.FY, k, prev_idx=loop.nn_idx, ws=loop.kws, row_order=loop.row_order)
        ops.raw_knn_match_prepared(loop.FX, loop.FY, k, idx2, d22, w, ws=loop.kws, row_order=loop.row_order)
        gap_fn()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.raw_knn_descriptor_bwd_fused(loop.xyz4, loop.idx, loop.FX, loop.FY, fya, loop.nn_idx, scale, gx, loop.max_fanin, far=loop.far)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts), sum(ts) / len(ts)
print("fused right after a kNN sweep (min, mean) ms:", after_knn(lambda: None))
import time
print("fused 20 ms after a kNN sweep (host sleep)  ms:", after_knn(lambda: (torch.cuda.synchronize(), time.sleep(0.02))))
print("fused with nn_dist2 + loss                  ms:", t(lambda: ops.raw_knn_descriptor_bwd_fused(loop.xyz4, loop.idx, loop.FX, loop.FY, fya, loop.nn_idx, scale, gx, loop.max_fanin, far=loop.far, nn_dist2=loop.nn_d2, loss_accum=loop.loss[0:1])))

for rows in (1024, 2048, 4096):
    sp = ops.SlotPlan(loop.idx, N, rows_per_block=rows)
    print(f"fused, slot plan rows/block={rows}: blocks {sp.blocks} max set {sp.max_set} outside {sp.outside_fraction:.4f}  ms:",
          t(lambda: ops.raw_knn_descriptor_bwd_fused(loop.xyz4, loop.idx, loop.FX, loop.FY, fya, loop.nn_idx, scale, gx, loop.max_fanin, slots=sp)))
