"""What the C3 loop's first iteration is made of: the content rows' tree order alone, and the iteration: usage first_iter_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd.style_loop import make_synthetic_c3, descriptor_tree_order
loop = make_synthetic_c3(style_density="matched", lists="exact", timers=True)
torch.cuda.synchronize()
t = time.time(); loop.step(); torch.cuda.synchronize(); t1 = time.time() - t
for rep in range(2):
    t = time.time(); o = descriptor_tree_order(loop.FX, leaf=32, iters=4, proj_dim=loop.cfg.row_order_proj); torch.cuda.synchronize(); t2 = time.time() - t
t = time.time(); loop.step(); torch.cuda.synchronize(); t3 = time.time() - t
print(f"first iteration {t1 * 1e3:.1f} ms (host clock), of which the tree order {t2 * 1e3:.1f} ms; second iteration {t3 * 1e3:.1f} ms; knn events {[round(a.elapsed_time(b), 2) for a, b in loop.knn_events[:2]]}")
