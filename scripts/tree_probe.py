"""Where the content rows' tree order spends its time at C3 (GPU time by events): usage tree_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd.style_loop import descriptor_tree_order
g = torch.Generator(device="cuda").manual_seed(0)
F = (torch.randn(1_000_000, 24, device="cuda", generator=g) @ torch.randn(24, 256, device="cuda", generator=g)).to(torch.bfloat16)
def ev(): e = torch.cuda.Event(enable_timing=True); e.record(); return e
for rep in range(2):
    t = time.time(); a = ev(); o = descriptor_tree_order(F, leaf=32, iters=4, proj_dim=16); b = ev(); torch.cuda.synchronize()
    print(f"tree order: GPU {a.elapsed_time(b):.1f} ms, host {(time.time() - t) * 1e3:.1f} ms")
X = F.float(); a = ev(); Xm = X - X.mean(0, keepdim=True); cov = Xm.T @ Xm; b = ev(); _, vec = torch.linalg.eigh(cov); c = ev(); P = (Xm @ vec[:, -16:]).contiguous(); d = ev(); torch.cuda.synchronize()
print(f"centre + covariance {a.elapsed_time(b):.2f} ms, eigh {b.elapsed_time(c):.2f} ms, projection {c.elapsed_time(d):.2f} ms")
a = ev(); o1 = torch.argsort(P[:, 0], stable=True); b = ev(); torch.cuda.synchronize(); print(f"one stable argsort of 1M floats {a.elapsed_time(b):.2f} ms")
