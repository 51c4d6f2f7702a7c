"""Would other 64-row tiles of the style set prune better than the leaves of the principal-axis tree?  Radii (enclosing-ball centre) and the ball test's skip fraction at
block level for: the tree order (as stored), k-means clusters in the top-p principal components stored cluster by cluster (tiles straddle cluster borders), and a
capacity-64 assignment to those centres (dev tool; uses the C3 loop's state at iteration n)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd.style_loop import make_synthetic_c3
os.environ["W3D_KNN_PRUNE"] = "0"
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 20
loop = make_synthetic_c3(style_density="matched", lists="exact")
for _ in range(n0):
    loop.step()
torch.cuda.synchronize()
FX = loop.FX.float(); FY = loop.FY.float()
M, d = FY.shape; T = M // 64
D = loop.nn_d2[:, -1].clamp_min(0).sqrt()
ro = loop.row_order.long()
N = FX.shape[0] // 512 * 512; ro = ro[:N]

def evaluate(name, perm):
    Yt = FY[perm][: T * 64].view(T, 64, d)
    c = Yt.mean(1); ar = torch.arange(T, device=c.device)
    for i in range(100):
        far = (Yt - c[:, None]).norm(dim=2).argmax(1)
        c = c + (Yt[ar, far] - c) / (i + 2)
    r = (Yt - c[:, None]).norm(dim=2).max(1).values
    tot = 0
    for b0 in range(0, N, 65536):
        idx = ro[b0:b0 + 65536]
        skip = (torch.cdist(FX[idx], c) - r[None, :]) > D[idx][:, None] * 1.0001
        tot += int(skip.view(-1, 512, T).all(1).sum())
    print(f"{name}: radius median {float(r.median()):.4g} mean {float(r.mean()):.4g} max {float(r.max()):.4g}; (512-row block, tile) pairs skippable {tot / (N // 512 * T):.3f}", flush=True)

evaluate("tree leaves (as stored)", torch.arange(M, device="cuda"))
for p in (8, 16):
    Ym = FY - FY.mean(0, keepdim=True)
    _, vec = torch.linalg.eigh(Ym.T @ Ym)
    Z = Ym @ vec[:, -p:]
    g = torch.Generator(device="cuda").manual_seed(0)
    cen = Z[torch.randperm(M, device="cuda", generator=g)[:T]].clone()
    for it in range(30):
        lab = torch.cdist(Z, cen).argmin(1)
        cnt = torch.bincount(lab, minlength=T).clamp_min(1)
        cen = torch.zeros_like(cen).index_add_(0, lab, Z) / cnt[:, None]
    evaluate(f"k-means in {p} components, rows stored by cluster", torch.argsort(lab, stable=True))
    # capacity 64: rows in order of how much they lose by not getting their nearest centre; each takes the nearest centre that still has room (a few rounds)
    dist = torch.cdist(Z, cen)
    order_c = dist.argsort(1)                                        # per row: centres by distance
    assign = torch.full((M,), -1, dtype=torch.long, device="cuda"); room = torch.full((T,), 64, dtype=torch.long, device="cuda")
    rank = torch.zeros(M, dtype=torch.long, device="cuda")
    for rnd in range(64):
        free = (assign < 0).nonzero().flatten()
        if free.numel() == 0: break
        want = order_c[free, rank[free].clamp(max=T - 1)]
        dwant = dist[free, want]
        o = torch.argsort(want * 1e3 + dwant / dwant.max())           # by centre, nearest first
        fs, ws = free[o], want[o]
        first = torch.searchsorted(ws, torch.arange(T, device="cuda"))
        pos = torch.arange(fs.numel(), device="cuda") - first[ws]
        ok = pos < room[ws]
        assign[fs[ok]] = ws[ok]
        room -= torch.bincount(ws[ok], minlength=T)
        rank[fs[~ok]] += 1
    if (assign < 0).any():
        left = (assign < 0).nonzero().flatten(); slots = torch.repeat_interleave(torch.arange(T, device="cuda"), room.clamp_min(0))
        assign[left] = slots[: left.numel()]
    evaluate(f"capacity-64 assignment to those centres ({p} components)", torch.argsort(assign, stable=True))
