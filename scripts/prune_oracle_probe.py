"""How much of the pruned sweep is unavoidable?  For the C3 loop after n0 iterations: the fraction of (row block, style tile) pairs that hold at least one of the block's
rows' k nearest neighbours (the pairs ANY exact test must keep) against what the ball test keeps (the plan's swept fraction), and the same with the ball test done in a
p-dimensional PCA subspace (|x - y| >= |Px - Py|: exact, tiles are tighter there) or as the intersection of both.
usage: prune_oracle_probe.py [n0=20]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
from wast3d_amd.style_loop import make_synthetic_c3
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 20
loop = make_synthetic_c3(style_density="matched", lists="exact", timers=True)
for _ in range(n0):
    loop.step()
torch.cuda.synchronize()
N, M, d, k = loop.N, loop.M, loop.d, loop.k
nblk, T = (N + 511) // 512, (M + 63) // 64
Nc = N // 512 * 512
ro = loop.row_order.long()[:Nc]
plan = float(ops.knn_prune_counts(N, M, d, k, loop.kws).sum()) / (nblk * T)
# oracle: tiles that hold a true neighbour of some row of the block
tiles = (loop.nn_idx.long()[ro] >> 6)                                # [Nc, k]
blk = torch.arange(Nc, device="cuda")[:, None].expand(-1, k) // 512
need = torch.zeros(Nc // 512, T, dtype=torch.bool, device="cuda")
need[blk.reshape(-1), tiles.reshape(-1)] = True
print(f"after {n0} iterations: plan keeps {plan:.4f} of the (row block, tile) pairs; pairs that hold a true neighbour of the block: {float(need.float().mean()):.4f}")
FX = loop.FX.float(); FY = loop.FY.float()
D = loop.nn_d2[:, -1].clamp_min(0).sqrt() * 1.0001

def geometry(Y):
    Yt = Y[: T * 64].view(T, 64, -1)
    c = Yt.mean(1); ar = torch.arange(T, device=c.device)
    for i in range(100):
        far = (Yt - c[:, None]).norm(dim=2).argmax(1)
        c = c + (Yt[ar, far] - c) / (i + 2)
    return c, (Yt - c[:, None]).norm(dim=2).max(1).values

c, r = geometry(FY)
Ym = FY - FY.mean(0, keepdim=True)
_, vec = torch.linalg.eigh(Ym.T @ Ym)
res = {}
for p in (0, 8, 16, 32):
    if p:
        V = vec[:, -p:]; cp, rp = geometry(FY @ V); Xp = FX @ V
    tot = 0; both = 0
    for b0 in range(0, Nc, 32768):
        ii = ro[b0:b0 + 32768]
        n_ball = (torch.cdist(FX[ii], c) - r[None, :]) <= D[ii][:, None]
        if p:
            n_proj = (torch.cdist(Xp[ii], cp) - rp[None, :]) <= D[ii][:, None]
            tot += int(n_proj.view(-1, 512, T).any(1).sum()); both += int((n_ball & n_proj).view(-1, 512, T).any(1).sum())
        else:
            tot += int(n_ball.view(-1, 512, T).any(1).sum())
    if p:
        print(f"  ball test in the top-{p} PCA subspace: keeps {tot / (Nc // 512 * T):.4f}; both tests: {both / (Nc // 512 * T):.4f}   (projected tile radius median {float(rp.median()):.4g} vs {float(r.median()):.4g})")
    else:
        print(f"  ball test (this script's float32 geometry): keeps {tot / (Nc // 512 * T):.4f}")
# per-row oracle vs per-row ball: how loose is the ball for a single row?
ii = ro[:32768]
n_ball = (torch.cdist(FX[ii], c) - r[None, :]) <= D[ii][:, None]
row_need = torch.zeros(32768, T, dtype=torch.bool, device="cuda"); row_need[torch.arange(32768, device="cuda")[:, None].expand(-1, k).reshape(-1), tiles[:32768].reshape(-1)] = True
print(f"  per ROW (first 32768 rows of the sweep order): ball keeps {float(n_ball.float().mean()):.4f} of the tiles, true-neighbour tiles {float(row_need.float().mean()):.4f}")

# ---- pivot (LAESA) test: for a pivot p, |d(x,p) - d(y,p)| <= d(x,y): tile t (rows y) can hold a candidate of x only if d(x,p) lies in [min_t d(y,p) - D_x, max_t d(y,p) + D_x]
def fps(Y, P, seed=0):                                               # farthest-point sampling of P pivots among the rows of Y
    g = torch.Generator(device="cuda").manual_seed(seed)
    first = int(torch.randint(0, Y.shape[0], (1,), device="cuda", generator=g))
    piv = [first]; dmin = (Y - Y[first]).norm(dim=1)
    for _ in range(P - 1):
        nxt = int(dmin.argmax()); piv.append(nxt); dmin = torch.minimum(dmin, (Y - Y[nxt]).norm(dim=1))
    return torch.tensor(piv, device="cuda")

Yt = FY[: T * 64].view(T, 64, -1)
nb = 64                                                               # row blocks evaluated
ii = ro[: nb * 512]
Dx = D[ii]
ball = (torch.cdist(FX[ii], c) - r[None, :]) <= Dx[:, None]          # [rows, T]
for kind in ("fps", "centres"):
    for P in (4, 8, 16, 32, 64):
        if kind == "fps":
            pv = FY[fps(FY, P)]
        else:
            pv = c[torch.linspace(0, T - 1, P).long()]               # tile centres evenly spaced along the style tree order
        dyp = torch.cdist(FY[: T * 64], pv).view(T, 64, P)
        lo, hi = dyp.min(1).values, dyp.max(1).values                 # [T, P]
        dxp = torch.cdist(FX[ii], pv)                                 # [rows, P]
        keep = torch.ones_like(ball)
        for p in range(P):
            keep &= (dxp[:, p:p + 1] >= lo[None, :, p] - Dx[:, None]) & (dxp[:, p:p + 1] <= hi[None, :, p] + Dx[:, None])
        both = keep & ball
        print(f"  pivots {kind:7s} P={P:2d}: per row keeps {float(keep.float().mean()):.4f} (with ball {float(both.float().mean()):.4f}); per block of 512 rows: {float(both.view(nb, 512, T).any(1).float().mean()):.4f} "
              f"(ball alone {float(ball.view(nb, 512, T).any(1).float().mean()):.4f}); per 64 rows {float(both.view(-1, 64, T).any(1).float().mean()):.4f}", flush=True)
