#!/usr/bin/env python3
"""Micro-benchmark at the bench shape: NN-loss backward + descriptor backward, three-kernel CSR form vs the fused LDS-window pass (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
from wast3d_amd.style_loop import morton_order, windowed_knn_lists
N, M, d, k = 1_000_000, 65536, 256, 20
g = torch.Generator().manual_seed(0)
X = (torch.rand(N, 3, generator=g) * 2.6 - 1.3).cuda()
X = X[morton_order(X)].contiguous()
idx = windowed_knn_lists(X, d + 1)
X4 = ops.raw_pack_xyz4(X)
FX = torch.empty(N, d, dtype=torch.bfloat16, device="cuda"); ops.raw_descriptor_fwd_x4(X4, idx, FX)
FY = (torch.rand(M, d, generator=g) * 0.3).to(torch.bfloat16).cuda()
nn = torch.randint(0, M, (N, k), generator=g, dtype=torch.int32).cuda()
gF = torch.empty(N, d, device="cuda"); gX = torch.zeros(N, 3, device="cuda")
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
s = 1.0 / (N * M)
print("knn_bwd ms:", t(lambda: ops.raw_knn_match_bwd(FX, FY, nn, s, gF, accumulate=False)))
fan = ops.list_max_fanin(idx, N); ymax = ops.absmax_bf16(FY); print("max fan-in", fan, "ymax", ymax)
print("window (gF given) ms:", t(lambda: ops.raw_descriptor_bwd_window(X4, idx, gF, gX, fan)))
for w in (0, 12288, 10240):
    print(f"fused window={w} ms:", t(lambda: ops.raw_knn_descriptor_bwd_fused(X4, idx, FX, FY, ymax, nn, s, gX, fan, window=w)))
csr = ops.DescriptorCSR(idx, N)
print("csr x4 ms:", t(lambda: csr.backward_x4(X4, gF, gX, accumulate=True)))
