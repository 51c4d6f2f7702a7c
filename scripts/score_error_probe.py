"""How large is the sweep's own rounding?  The prune plan's margin assumes |computed score - exact score| (+ half the norm error) <= E = 2e-6 (|x|^2 + 2 max|c|^2).
Measured here on the C3 loop's state: the kernel's squared distances of the selected pairs against a float64 evaluation of the same rounded rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
from wast3d_amd.style_loop import make_synthetic_c3
for style in ("matched", "sparse"):
    loop = make_synthetic_c3(style_density=style, lists="exact")
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
        loop.step()
    torch.cuda.synchronize()
    rows = torch.arange(0, loop.N, max(1, loop.N // 20000), device="cuda")
    X = loop.FX[rows].double(); idx = loop.nn_idx[rows].long()
    Yg = loop.FY.double()[idx]                                        # [n, k, d]
    d2_exact = ((X[:, None] - Yg) ** 2).sum(2)
    d2_k = loop.nn_d2[rows].double()
    xn = (X * X).sum(1)
    pr = ops.KnnPrune(loop.FY)
    E = 2e-6 * (xn + 2 * pr.cnorm_max)
    err = (d2_k - d2_exact).abs() / 2                                 # in score units (d2 = |x|^2 - 2 score)
    print(f"{style}: |x|^2 median {float(xn.median()):.4g}, max|c|^2 bound {pr.cnorm_max:.4g}; score error max {float(err.max()):.3g}, median {float(err.median()):.3g}; "
          f"assumed bound E min {float(E.min()):.3g}; worst error / E = {float((err / E[:, None]).max()):.4f}")
    del loop; torch.cuda.empty_cache()
