"""Would a HYBRID warm-start bound do?  The bound kernel gathers all k = 20 previous neighbours of every row (10 GB of style rows per iteration at C3, L2-bandwidth-bound:
0.8 ms alone, 1.6-1.9 ms next to the render leg).  Hybrid: gather only the G FARTHEST previous neighbours; the others are covered by the triangle inequality,
d_new(x', y_j) <= d_old(x, y_j) + |x' - x| <= d_old(k-G-1) + m.   bound_hybrid = max(max_{j >= k-G} d_new(j), d_old(k-G-1) + m) >= bound_exact = max_j d_new(j).
This probe measures, on the C3 loop's own consecutive iterations, how often the hybrid bound equals the exact one and how much larger it is otherwise.
OUTCOME (round 3; the kernel variant was built, measured and removed): at G = 10 the hybrid bound equals the exact one for 98 % of the rows and the bound kernel
drops from 0.74 to 0.46 ms — but the sweep that follows goes from 8.91 to 9.13 ms (G = 8: 9.22, G = 6: 9.44): whatever the bound kernel saves the sweep loses.
usage: hybrid_bound_probe.py [n0=20]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd.style_loop import make_synthetic_c3

n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 20
loop = make_synthetic_c3(style_density=os.environ.get("STYLE", "matched"), lists="exact")
for _ in range(n0):
    loop.step()
torch.cuda.synchronize()
S = 100_000
g = torch.Generator(device="cuda").manual_seed(1)
rows = torch.randperm(loop.N, device="cuda", generator=g)[:S]
FX_old = loop.FX[rows].float()
idx_old = loop.nn_idx[rows].long()
d2_old = loop.nn_d2[rows].clone()                                   # ascending
loop.step(); torch.cuda.synchronize()
FX_new = loop.FX[rows].float()
FY = loop.FY.float()
k = loop.k
m = (FX_new - FX_old).norm(dim=1)
d_new = torch.stack([(FX_new - FY[idx_old[:, j]]).norm(dim=1) for j in range(k)], 1)      # current distance to every previous neighbour
d_old = d2_old.clamp_min(0).sqrt()
exact = d_new.max(1).values
true_kth = loop.nn_d2[rows][:, k - 1].clamp_min(0).sqrt()           # the k-th distance the sweep then found
print(f"after {n0} iterations, {S} rows: motion |x' - x| median {float(m.median()):.4g}; old k-th distance median {float(d_old[:, -1].median()):.4g}; "
      f"exact bound / new k-th distance: median {float((exact / true_kth).median()):.4f}, mean {float((exact / true_kth).mean()):.4f}")
print(f"  pure motion bound (G = 0): (d_old(k-1) + m) / exact: median {float(((d_old[:, -1] + m) / exact).median()):.4f}, mean {float(((d_old[:, -1] + m) / exact).mean()):.4f}")
for G in (2, 4, 5, 8, 10, 14):
    gathered = d_new[:, k - G:].max(1).values
    rest = d_old[:, k - G - 1] + m
    hyb = torch.maximum(gathered, rest)
    eq = (hyb <= exact * (1 + 1e-6)).float().mean()
    ratio = hyb / exact
    print(f"  G = {G:2d}: hybrid == exact for {float(eq) * 100:.1f} % of the rows; hybrid / exact: mean {float(ratio.mean()):.4f}, p90 {float(ratio.quantile(0.9)):.4f}, p99 {float(ratio.quantile(0.99)):.4f}, max {float(ratio.max()):.3f}; "
          f"mean of (hybrid / exact)^2 (~ volume reached in 2-D) {float((ratio ** 2).mean()):.4f}")
