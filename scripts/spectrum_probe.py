import os, sys
sys.path.insert(0, os.getcwd())
import torch
from wast3d_amd.style_loop import make_synthetic_c3
for kw in (dict(N=1_000_000, M=65536, d=256), dict(N=200_000, M=16384, d=128), dict(N=1_000_000, M=65536, d=256, style_density="sparse")):
    loop = make_synthetic_c3(k=20, lists="exact", **({"style_density": "matched"} | kw))
    loop.step(); torch.cuda.synchronize()
    for name, F in (("style", loop.FY), ("content", loop.FX)):
        X = F[:: max(1, F.shape[0] // 65536)].float(); X = X - X.mean(0, keepdim=True)
        ev = torch.linalg.eigvalsh(X.T @ X).flip(0).clamp_min(0).sqrt()
        print(kw, name, "singular values / largest:", [round(float(v), 3) for v in (ev[:8] / ev[0])])
    del loop; torch.cuda.empty_cache()
