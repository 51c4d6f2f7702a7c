"""How stable are the content rows' neighbour SETS from one iteration to the next (C3 loop)?  Fraction of rows whose k neighbours are the same set as in the previous
iteration, and the mean number of neighbours that changed: usage nn_stability_probe.py [stations=10,40,280,600]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd.style_loop import make_synthetic_c3
st = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "10,40,280,600").split(",")]
loop = make_synthetic_c3(style_density="matched", lists="exact", timers=True)
it = 0
for s in st:
    while it < s:
        loop.step(); it += 1
    torch.cuda.synchronize()
    prev = loop.nn_idx.sort(1).values.clone()
    for n in range(1, 4):
        loop.step(); it += 1
        torch.cuda.synchronize()
        cur = loop.nn_idx.sort(1).values
        same = (cur == prev).all(1)
        # neighbours of the new list that were not in the old one
        new_in_old = (loop.nn_idx[:, :, None] == prev[:, None, :]).any(2)
        changed = (~new_in_old).sum(1).float()
        hist = torch.bincount(changed.long(), minlength=6)[:6].float() / changed.numel()
        print(f"iteration {it}: same set {float(same.float().mean()):.3f}; neighbours replaced per row mean {float(changed.mean()):.2f}, histogram 0..5 {[round(float(x), 3) for x in hist]}", flush=True)
        prev = cur.clone()
