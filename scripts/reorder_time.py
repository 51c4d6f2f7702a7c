"""Time of KnnBounds.reorder at a refresh of the C3 loop (iteration 288): usage reorder_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd.style_loop import make_synthetic_c3
loop = make_synthetic_c3(style_density="matched", lists="exact", timers=True)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 287):
    loop.step()
torch.cuda.synchronize()
cur = loop.row_order
new = cur.long()[torch.argsort(loop.nn_idx[cur.long(), 0] >> 6, stable=True)].to(torch.int32).contiguous()
b = loop.bounds
import copy
for rep in range(3):
    lb0, cum0, summ0 = b.lb.clone(), b.cum.clone(), b.summ.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); b.reorder(cur, new); e1.record(); torch.cuda.synchronize()
    print(f"reorder (host ops + kernel + fills; summaries folded: {rep == 0}): {e0.elapsed_time(e1):.3f} ms")
    b.lb.copy_(lb0); b.cum.copy_(cum0); b.summ.copy_(summ0); b.rebase = False; b.folded = rep >= 0
