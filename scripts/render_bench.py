#!/usr/bin/env python3
"""Micro-benchmark: the render leg of the C3 loop alone (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd.style_loop import make_synthetic_c3
loop = make_synthetic_c3(N=1_000_000, M=4096, d=64, k=20)
xyz = loop.pc._xyz.data
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("render leg ms:", t(lambda: loop._render_leg(xyz)), "R", loop.last_R)
