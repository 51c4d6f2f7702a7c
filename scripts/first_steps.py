#!/usr/bin/env python3
"""Wall time and kNN-kernel time of the first iterations of the C3 loop (dev tool): the first step is cold, the second one sets up the warm start."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd.style_loop import make_synthetic_c3
loop = make_synthetic_c3(N=1_000_000, M=65536, d=256, k=20, timers=True)
for it in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loop.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    print(f"step {it+1}: {dt:.2f} ms  kNN launches {loop.knn_ms():.2f} ms  warm={loop._warm_mode}  motion mean {float(loop.motion2.sqrt().mean()):.5f}  kth mean {float(loop.nn_d2[:, -1].sqrt().mean()):.5f}")
