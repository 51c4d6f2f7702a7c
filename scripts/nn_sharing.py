#!/usr/bin/env python3
"""How many distinct style rows the loop's matches use, per group of consecutive content rows (dev tool; W3D_STYLE_DENSITY=sparse|matched)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd.style_loop import make_synthetic_c3
loop = make_synthetic_c3(N=1_000_000, M=65536, d=256, k=20)
for _ in range(3): loop.step()
torch.cuda.synchronize()
nn = loop.nn_idx.long()
N = nn.shape[0]
for G in (4, 16, 64, 256, 3907):
    n = (N // G) * G
    g = nn[:n].view(-1, G * 20)
    s = g.sort(1).values
    distinct = 1 + (s[:, 1:] != s[:, :-1]).sum(1)
    print(f"group of {G} consecutive rows: {G*20} refs, distinct style rows mean {distinct.float().mean().item():.1f} ({distinct.float().mean().item()/(G*20):.3f})")
print("distinct style rows used overall:", nn.unique().numel())
