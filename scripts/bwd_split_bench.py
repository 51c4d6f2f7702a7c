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

loop = make_synthetic_c3(style_density="matched", lists="knn")
for _ in range(4):
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
