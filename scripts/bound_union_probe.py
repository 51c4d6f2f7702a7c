#!/usr/bin/env python3
"""Probe (round 6): how many DISTINCT style rows do the previous-neighbour lists of G consecutive content rows (in the sweep order) name?  The bound kernel gathers k rows per
content row (10 GB through L2 at C3); a workgroup that staged the union of its rows' lists in LDS once would gather union / (G k) of that."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd.style_loop import make_synthetic_c3
loop = make_synthetic_c3(N=1_000_000, M=65536, d=256, k=20, lists="exact", style_density="matched")
for it in range(26):
    loop.step()
    if it in (6, 15, 25):
        torch.cuda.synchronize()
        order = loop.row_order.long()
        nn = loop.nn_idx[order]                                   # lists in sweep order
        for G in (16, 32, 64, 128, 256):
            n = (nn.shape[0] // G) * G
            blk = nn[:n].view(-1, G * loop.k)
            s = blk.sort(1).values
            uniq = 1 + (s[:, 1:] != s[:, :-1]).sum(1)
            print(f"iteration {it + 1}: G = {G:3d} rows: {float(uniq.float().mean()):7.1f} distinct of {G * loop.k} listed ({float(uniq.float().mean()) / (G * loop.k):.3f}), max {int(uniq.max())}, "
                  f"LDS for the mean union {float(uniq.float().mean()) * 512 / 1024:.0f} KB")
