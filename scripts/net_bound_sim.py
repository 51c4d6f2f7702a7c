"""Would lower bounds that decay by NET displacement pay?  The cached bounds of the prune plan decay by the PATH LENGTH of a row's motion since the pair was swept; the rows
jitter (scripts/path_vs_net_probe.py: net / path = 0.2-0.4 after 32 iterations).  A second stored value per pair, relative to a snapshot of the rows taken at the epoch's
start (bound = m - disp(t_s) - disp(t), disp = |x - x_ref|), is valid too; the plan could take the larger of the two.
Simulation on the C3 loop's own rows: 40 row blocks (the loop's blocks, in its row order), one epoch of W iterations from a station; per iteration the exact per-(row, tile)
minimum distances, the block-level keep decision under (a) path bounds, (b) the larger of path and net bounds; swept pairs are re-recorded exactly (as the sweep does).
Every pair starts with its exact distance at the epoch's start (the real plan starts from ball bounds for never-swept pairs: both schemes are flattered alike).
usage: net_bound_sim.py [station=250] [W=32] [blocks=40]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd.style_loop import make_synthetic_c3
s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 250
W = int(sys.argv[2]) if len(sys.argv) > 2 else 32
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 40
loop = make_synthetic_c3(style_density="matched", lists="exact", timers=True)
for _ in range(s0):
    loop.step()
torch.cuda.synchronize()
N, M = loop.N, loop.M
T = (M + 63) // 64
nblk = N // 512
g = torch.Generator().manual_seed(5)
blocks = torch.randperm(nblk, generator=g)[:NB].cuda()
rows = loop.row_order.long().view(-1)[: nblk * 512].view(nblk, 512)[blocks].reshape(-1)          # [NB * 512] row ids, block-major (the order of station s0: fixed for the epoch here)
FY = loop.FY.float()

def mins(X):                                                          # exact min distance of every row to every 64-row tile
    out = torch.empty(X.shape[0], T, device="cuda")
    for i in range(0, X.shape[0], 2048):
        out[i:i + 2048] = torch.cdist(X[i:i + 2048], FY)[:, : T * 64].view(-1, T, 64).min(-1).values
    return out

ref = loop.FX[rows].float().clone(); prev = ref.clone()
m0 = mins(ref)
zero = torch.zeros(rows.numel(), device="cuda")
st = {"path": {"hp": m0.clone()}, "both": {"hp": m0.clone(), "hn": m0.clone()}}   # hp = m + cum(t_s), hn = m - disp(t_s)
cum = zero.clone()
for t in range(1, W + 1):
    loop.step()
    cur = loop.FX[rows].float()
    cum += (cur - prev).norm(dim=1); prev = cur.clone()
    disp = (cur - ref).norm(dim=1)
    D = loop.nn_d2[rows, -1].clamp_min(0).sqrt() * 1.00001
    m = mins(cur)
    need = (m <= D[:, None]).view(NB, 512, T).any(1)
    line = f"epoch iteration {t:2d}: pairs with a true neighbour {float(need.float().mean()):.4f}"
    for name, s in st.items():
        b = s["hp"] - cum[:, None]
        if name == "both":
            b = torch.maximum(b, s["hn"] - disp[:, None])
        keep = (b <= D[:, None]).view(NB, 512, T).any(1)               # block keeps the tile if any of its rows may have a neighbour there
        assert bool((keep | ~need).all())                              # exactness: a needed pair is never pruned
        km = keep[:, None, :].expand(-1, 512, -1).reshape(-1, T)
        s["hp"] = torch.where(km, m + cum[:, None], s["hp"])
        if name == "both":
            s["hn"] = torch.where(km, m - disp[:, None], s["hn"])
        line += f"   kept by {name} bounds {float(keep.float().mean()):.4f}"
    print(line, flush=True)
