"""Would CACHED lower bounds (Elkan-style, exact) prune more than the ball test?  When a (row block, style tile) pair is swept, every row x of the block learns the
exact minimum distance m_t(x) to the tile's 64 rows (it is the maximum score of the unit, which the selection's pretest already forms).  While x moves by a
cumulative |dx| since then, min_y |x' - y| >= m_t(x) - sum |dx| (triangle inequality): the tile can hold no candidate of x as long as m_t(x) - drift > D_x (D_x =
this iteration's bound, recomputed every iteration).  A block re-sweeps a tile when ANY of its rows needs it, which refreshes all its rows' m_t.
This probe simulates that on the C3 loop's own states with torch arithmetic (a sample of row blocks, fixed membership) and prints, per iteration, the fraction of
(block, tile) pairs kept by: the ball test, the cached bounds, both, and the pairs that hold a true neighbour (what any exact test must keep).
usage: cache_bound_probe.py [n0=20] [iters=60] [blocks=24]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
from wast3d_amd.style_loop import make_synthetic_c3

n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 20
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 24
loop = make_synthetic_c3(style_density="matched", lists="exact", timers=True)
for _ in range(n0):
    loop.step()
torch.cuda.synchronize()
N, M, d, k = loop.N, loop.M, loop.d, loop.k
T = (M + 63) // 64
nblk = (N + 511) // 512
ro = loop.row_order.long()
sel = torch.linspace(0, nblk - 2, nb).long()
rows = torch.stack([ro[int(b) * 512:(int(b) + 1) * 512] for b in sel]).cuda()        # [nb, 512] fixed membership
FY = loop.FY.float()
yn = (FY * FY).sum(1)
pr = loop.prune
c, r = pr.centres.float(), pr.cradius


def tile_min_dist(X):                                                 # X [512, d] f32 -> [512, T] exact min distance to each tile's rows
    d2 = (X * X).sum(1, keepdim=True) + yn[None, :] - 2.0 * (X @ FY.T)
    return d2.view(X.shape[0], T, 64).min(2).values.clamp_min(0).sqrt()


state = None                                                          # [nb, 512, T]: m + C at the last sweep of the pair
C = torch.zeros(N, device="cuda")
prevF = loop.FX.float().clone()
log = []
for it in range(iters):
    loop.step()
    torch.cuda.synchronize()
    F = loop.FX.float()
    mot = (F - prevF).norm(dim=1)
    C += mot * (1 + 1e-6)
    prevF = F.clone()
    D = loop.nn_d2[:, -1].clamp_min(0).sqrt() * 1.0001               # the new k-th distance (the plan's bound is the previous neighbours' farthest: >= this, within a few %)
    plan = float(ops.knn_prune_counts(N, M, d, k, loop.kws).sum()) / (nblk * T)
    keep_ball = keep_cache = keep_both = keep_true = 0
    per_row_cache = 0.0
    if state is None:
        state = torch.empty(nb, 512, T, device="cuda")
    for bi in range(nb):
        ii = rows[bi]
        X = F[ii]
        m = tile_min_dist(X)
        Dx = D[ii][:, None]
        ball = ((torch.cdist(X, c) - r[None, :]) <= Dx)
        true = (m <= Dx)
        if it == 0:
            need_c = torch.ones_like(ball)
        else:
            need_c = (state[bi] - C[ii][:, None]) <= Dx
        bt_ball, bt_cache, bt_true = ball.any(0), need_c.any(0), true.any(0)
        assert bool((bt_true & ~bt_cache).sum() == 0) and bool((bt_true & ~bt_ball).sum() == 0), "a bound skipped a tile that holds a candidate"
        both = bt_ball & bt_cache
        keep_ball += int(bt_ball.sum()); keep_cache += int(bt_cache.sum()); keep_both += int(both.sum()); keep_true += int(bt_true.sum())
        per_row_cache += float(need_c.float().mean())
        # the pairs swept (policy: both tests) refresh their rows' bounds
        state[bi][:, both] = (m + C[ii][:, None])[:, both]
    den = nb * T
    log.append((plan, keep_ball / den, keep_cache / den, keep_both / den, keep_true / den))
    print(f"it {n0 + it + 1:4d}: plan {plan:.4f} | sample: ball {keep_ball / den:.4f}  cache {keep_cache / den:.4f}  both {keep_both / den:.4f}  true {keep_true / den:.4f}"
          f"  | per-row cache need {per_row_cache / nb:.4f}  motion mean {float(mot.mean()):.3g}  kth {float(D.mean()):.3g}", flush=True)
tail = log[len(log) // 2:]
mean = [sum(x[j] for x in tail) / len(tail) for j in range(5)]
print(f"steady state (last {len(tail)} iterations): plan {mean[0]:.4f}  ball {mean[1]:.4f}  cache {mean[2]:.4f}  both {mean[3]:.4f}  true {mean[4]:.4f}")
