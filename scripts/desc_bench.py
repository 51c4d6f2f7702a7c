#!/usr/bin/env python3
"""Micro-benchmark: descriptor backward, atomic vs CSR (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
N, K = 1_000_000, 257
g = torch.Generator().manual_seed(0)
X = (torch.rand(N, 3, generator=g) * 2.6 - 1.3).cuda()
idx = torch.randint(0, N, (N, K), generator=g, dtype=torch.int32); idx[:, 0] = torch.arange(N, dtype=torch.int32); idx = idx.cuda()
gF = torch.randn(N, K - 1, generator=g).cuda()
gX = torch.zeros_like(X)
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("atomic bwd ms:", t(lambda: ops.raw_descriptor_bwd(X, idx, gF, gX)))
import time; t0 = time.time(); csr = ops.DescriptorCSR(idx, N); torch.cuda.synchronize(); print("csr build s:", time.time() - t0)
g1 = torch.zeros_like(X)
print("csr bwd ms:", t(lambda: csr.backward(X, gF, g1, accumulate=False)))
gX.zero_(); ops.raw_descriptor_bwd(X, idx, gF, gX)
print("max diff:", float((gX - g1).abs().max()), float(gX.abs().max()))
g2 = torch.zeros_like(X)
print("sorted two-pass bwd ms:", t(lambda: csr.backward_sorted(X, gF, g2, accumulate=False)))
print("max diff sorted vs csr:", float((g2 - g1).abs().max()))
