import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd.style_loop import make_synthetic_c3
loop = make_synthetic_c3(N=1_000_000, M=65536, d=256, k=20, style_density="matched")
for _ in range(4): loop.step()
torch.cuda.synchronize()
nn = loop.nn_idx.long()
order = torch.argsort(nn[:, 0], stable=True)
for name, o in (("morton order", torch.arange(nn.shape[0], device="cuda")), ("sorted by nearest style row", order)):
    s = nn[o]
    for G in (16, 64, 512, 4096, 32768):
        n = (s.shape[0] // G) * G
        g = s[:n].view(-1, G * 20).sort(1).values
        distinct = 1 + (g[:, 1:] != g[:, :-1]).sum(1)
        print(f"{name}: group {G}: distinct/refs {distinct.float().mean().item() / (G * 20):.3f}  distinct rows {distinct.float().mean().item():.0f} ({distinct.float().mean().item()*512/1e6:.2f} MB)")
