#!/usr/bin/env python3
"""Reference points for the distance-GEMM roofline on the box at hand (dev tool): what the vendor GEMM (hipBLASLt through torch.matmul) sustains
 (a) on a large square bf16 problem — the practical MFMA ceiling under the power limit, and
 (b) on the kNN's own shape (K = d = 256, output materialised) — the formulation the fused kernel replaces."""
import torch
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (m, n, k) in ((8192, 8192, 8192), (16384, 16384, 4096), (131072, 65536, 256)):
    A = torch.randn(m, k, device="cuda", dtype=torch.bfloat16); B = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
    C = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    ms = t(lambda: torch.matmul(A, B.t(), out=C))
    print(f"matmul bf16 {m}x{n}x{k}: {ms:.3f} ms  {2*m*n*k/ms/1e9:.1f} TFLOP/s  (output {m*n*2/1e9:.1f} GB -> {m*n*2/ms/1e6:.0f} GB/s)")
    del A, B, C
