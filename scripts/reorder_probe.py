"""How scattered is a row-order refresh?  At iteration `at` (order age = at mod 32) form the 'tile' re-bucketing the loop would and report: rows whose successor in the
new order was their successor in the old one, distinct 16-row sectors of the old order per 64 new positions, rows that changed 512-row block: usage reorder_probe.py [at=288]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd.style_loop import make_synthetic_c3
at = int(sys.argv[1]) if len(sys.argv) > 1 else 288
loop = make_synthetic_c3(style_density="matched", lists="exact", timers=True)
for _ in range(at):
    loop.step()
torch.cuda.synchronize()
N = loop.N
cur = loop.row_order.long()
key = loop.nn_idx[cur, 0] >> 6
new = cur[torch.argsort(key, stable=True)]
inv = torch.empty(N, dtype=torch.long, device="cuda"); inv[cur] = torch.arange(N, device="cuda")
src = inv[new]
cont = float((src[1:] == src[:-1] + 1).float().mean())
n64 = N // 64
sec = (src[:n64 * 64] // 16).view(n64, 64)
distinct = float(torch.tensor([len(torch.unique(r)) for r in sec[::97]]).float().mean())
moved = float(((src // 512) != (torch.arange(N, device="cuda") // 512)).float().mean())
mono = float((key[1:] >= key[:-1]).float().mean())
print(f"iteration {at}: order age {loop._sweeps % 32}: successor kept {cont:.3f}, distinct old sectors per 64 new positions {distinct:.1f} (4 = contiguous), rows in another block {moved:.3f}, "
      f"old order still sorted by bucket at {mono:.3f} of the positions")
