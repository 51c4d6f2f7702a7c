"""Runs the fused backward over the slot plan a few times on the C3 loop's own state (for scripts/bwd_pmc.sh: counters of exactly this kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
from wast3d_amd.style_loop import make_synthetic_c3
loop = make_synthetic_c3(style_density="matched", lists="exact")
for _ in range(3):
    loop.step()
torch.cuda.synchronize()
N, M = loop.N, loop.FY.shape[0]
gx = torch.zeros(N, 3, device="cuda")
for _ in range(3):
    ops.raw_knn_descriptor_bwd_fused(loop.xyz4, loop.idx, loop.FX, loop.FY, loop._fy_absmax(), loop.nn_idx, 1.0 / (N * M), gx, loop.max_fanin, slots=loop.slots)
torch.cuda.synchronize()
