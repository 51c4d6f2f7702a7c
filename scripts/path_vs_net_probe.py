"""Do the content rows DRIFT or JITTER?  The cached lower bounds decay by the PATH LENGTH of a row's feature motion since a pair was swept (sum of |F_t - F_{t-1}|); a
snapshot-based bound would decay by net displacements |F_t - F_ref|.  Over windows of W iterations at several stations of the C3 loop: path length vs net displacement
(rows sampled), next to the margin that decides a tile (k-th distance).  usage: path_vs_net_probe.py [stations=20,250,500] [W=32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd.style_loop import make_synthetic_c3
st = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "20,250,500").split(",")]
W = int(sys.argv[2]) if len(sys.argv) > 2 else 32
loop = make_synthetic_c3(style_density="matched", lists="exact", timers=True)
g = torch.Generator(device="cuda").manual_seed(3)
rows = torch.randperm(loop.N, device="cuda", generator=g)[:50000]
it = 0
for s in st:
    while it < s:
        loop.step(); it += 1
    ref = loop.FX[rows].float().clone(); prev = ref.clone()
    path = torch.zeros(rows.numel(), device="cuda")
    for w in range(1, W + 1):
        loop.step(); it += 1
        cur = loop.FX[rows].float()
        path += (cur - prev).norm(dim=1); prev = cur.clone()
        if w in (1, 2, 4, 8, 16, 32, W):
            net = (cur - ref).norm(dim=1)
            kth = loop.nn_d2[rows, -1].clamp_min(0).sqrt()
            print(f"station {s:4d} + {w:2d}: path {float(path.mean()):.5f}  net {float(net.mean()):.5f}  net/path {float((net / path.clamp_min(1e-12)).mean()):.3f}   "
                  f"k-th distance {float(kth.mean()):.4f}  path/kth {float((path / kth).mean()):.3f}", flush=True)
