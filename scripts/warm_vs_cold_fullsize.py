#!/usr/bin/env python3
"""At the bench size (1M x 65 536 x 256), both style patches: the loop's warm-started matches against a cold sweep of the same rows (dev tool).
Expected: 0 index mismatches of 20M, 0 rows redone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
from wast3d_amd.style_loop import make_synthetic_c3
for style in ("matched", "sparse"):
    loop = make_synthetic_c3(N=1_000_000, M=65536, d=256, k=20, style_density=style)
    for it in range(5):
        loop.step(); torch.cuda.synchronize()
        if it == 0: continue
        idx_c, d2_c = ops.raw_knn_match(loop.FX, loop.FY, loop.k)
        neq = int((loop.nn_idx != idx_c).sum())
        print(style, it, loop._warm_mode, "index mismatches", neq, "max |d2 diff|", float((loop.nn_d2 - d2_c).abs().max()), "repaired", loop.repaired_rows_max)
