"""Is the content rows' motion a drift or a jitter?  Ratio of the displacement |x_{i+n} - x_i| to the path length sum |dx| over n iterations (1 = pure drift), per-iteration
motion against the k-th distance: usage motion_probe.py [stations=30,275,600]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd.style_loop import make_synthetic_c3
st = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "30,275,600").split(",")]
loop = make_synthetic_c3(style_density="matched", lists="exact", timers=True)
it = 0
for s in st:
    while it < s:
        loop.step(); it += 1
    torch.cuda.synchronize()
    F0 = loop.FX.float().clone(); prev = F0.clone(); path = torch.zeros(loop.N, device="cuda")
    for n in range(1, 17):
        loop.step(); it += 1
        torch.cuda.synchronize()
        F = loop.FX.float()
        step = (F - prev).norm(dim=1); path += step; prev = F.clone()
        if n in (1, 2, 4, 8, 16):
            disp = (F - F0).norm(dim=1)
            kth = loop.nn_d2[:, -1].clamp_min(0).sqrt(); k1 = loop.nn_d2[:, 0].clamp_min(0).sqrt()
            print(f"it {s}+{n:2d}: step motion mean {float(step.mean()):.4g} median {float(step.median()):.4g} | displacement/path mean {float((disp / path.clamp_min(1e-12)).mean()):.3f}"
                  f" | path/kth mean {float((path / kth).mean()):.4f}  disp/kth {float((disp / kth).mean()):.4f} | kth mean {float(kth.mean()):.4g} 1st {float(k1.mean()):.4g}", flush=True)
