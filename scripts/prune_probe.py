"""Could a sweep skip (row group, style tile) pairs EXACTLY?  Ball test on the C3 loop's own state: a row x can have no candidate in tile t if
|x - c_t| - r_t > D_x (c_t, r_t: centroid and radius of the tile's 64 style rows; D_x: the row's current k-th neighbour distance = the best bound a warm
start can have).  Reports the fraction of (group, tile) pairs every row of the group could skip, for groups of 16 / 64 / 512 rows in the sweep's row order."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd.style_loop import make_synthetic_c3
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 20
loop = make_synthetic_c3(style_density=os.environ.get("STYLE", "matched"), lists="exact")
for _ in range(n0):
    loop.step()
torch.cuda.synchronize()
FX = loop.FX.float(); FY = loop.FY.float()
TS = int(os.environ.get("TILE", "64"))                              # rows per tile (the kernel's tiles are 64; 32 = its MFMA sub-blocks)
M = FY.shape[0]; T = M // TS
Yt = FY[: T * TS].view(T, TS, -1)
c = Yt.mean(1)                                                       # [T, d]
r = (Yt - c[:, None]).norm(dim=2).max(1).values                      # [T]
if os.environ.get("MEB"):                                            # minimum enclosing ball (Badoiu-Clarkson) instead of the mean as the centre
    for i in range(int(os.environ["MEB"])):
        dd = (Yt - c[:, None]).norm(dim=2)
        far = dd.argmax(1)
        c = c + (Yt[torch.arange(T, device=c.device), far] - c) / (i + 2)
    r2 = (Yt - c[:, None]).norm(dim=2).max(1).values
    print("radius: mean centre median %.4g -> enclosing-ball centre median %.4g" % (float(r.median()), float(r2.median())))
    r = r2
D = loop.nn_d2[:, -1].clamp_min(0).sqrt()                            # [N] current k-th distance
ro = loop.row_order.long() if loop.row_order is not None else torch.arange(FX.shape[0], device="cuda")
N = FX.shape[0] // 512 * 512
ro = ro[:N]
print("k-th distance: median %.4g; tile radius: median %.4g; |x - c_t| median over a sample %.4g" % (float(D.median()), float(r.median()), float(torch.cdist(FX[ro[:2048]], c).median())))
tot = {16: 0, 64: 0, 512: 0}; rows_skip = 0
for b0 in range(0, N, 65536):
    idx = ro[b0:b0 + 65536]
    d_it = torch.cdist(FX[idx], c)                                  # [rows, T]
    skip = (d_it - r[None, :]) > D[idx][:, None] * 1.0001
    rows_skip += int(skip.sum())
    for g in tot:
        tot[g] += int(skip.view(-1, g, T).all(1).sum())
print("rows x tiles skippable: %.3f" % (rows_skip / (N * T)))
for g in tot:
    print("groups of %d rows x tiles skippable: %.3f" % (g, tot[g] / (N // g * T)))

# ---- a second exact test in a p-dimensional projection (|x - y| >= |Px - Py| for any orthogonal projection P): tiles are tighter there (less of the high-dimensional
# noise), the bound D_x stays the full-space one.  Skippable = ball test OR projected test.
if os.environ.get("PROJ"):
    p = int(os.environ["PROJ"])
    Ym = FY - FY.mean(0, keepdim=True)
    _, vec = torch.linalg.eigh(Ym.T @ Ym)
    V = vec[:, -p:]                                                  # [d, p] top principal directions of the style set
    Yp = (FY @ V)[: T * TS].view(T, TS, p)
    cp = Yp.mean(1)
    for i in range(100):
        far = (Yp - cp[:, None]).norm(dim=2).argmax(1)
        cp = cp + (Yp[torch.arange(T, device=cp.device), far] - cp) / (i + 2)
    rp = (Yp - cp[:, None]).norm(dim=2).max(1).values
    print("projected (p=%d) tile radius median %.4g" % (p, float(rp.median())))
    Xp = FX @ V
    tot2 = {16: 0, 64: 0, 512: 0}; both = {16: 0, 64: 0, 512: 0}
    for b0 in range(0, N, 65536):
        idx = ro[b0:b0 + 65536]
        s1 = (torch.cdist(FX[idx], c) - r[None, :]) > D[idx][:, None] * 1.0001
        s2 = (torch.cdist(Xp[idx], cp) - rp[None, :]) > D[idx][:, None] * 1.0001
        for g in tot2:
            tot2[g] += int(s2.view(-1, g, T).all(1).sum())
            both[g] += int((s1 | s2).view(-1, g, T).all(1).sum())
    for g in tot2:
        print("groups of %d rows: projected test alone %.3f, either test %.3f" % (g, tot2[g] / (N // g * T), both[g] / (N // g * T)))
