"""EXPERIMENT: warm unpruned sweeps with the ring of four 32-row tiles + arrival counters (tuning bit 0x8000000) against the shipped kernel (two 64-row tiles, one
barrier per tile): same lists bit for bit?  time?   usage: ring4_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
from wast3d_amd.style_loop import make_synthetic_c3

def ab(X, Y, k, prev, name, row_order=None):
    ws = ops.KnnWorkspace()
    out = {}
    for mode in (0, 0x8000000, 0, 0x8000000):
        ops.knn_set_tuning(mode=mode)
        i2 = torch.empty_like(prev); dd = torch.empty(prev.shape, dtype=torch.float32, device="cuda"); evs = []
        for _ in range(4):
            w = ops.raw_knn_prepare(X, Y, k, prev_idx=prev, ws=ws, row_order=row_order)
            evs.append(ops.raw_knn_match_prepared(X, Y, k, i2, dd, w, timed=True, ws=ws, row_order=row_order))
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in evs)[1]
        out.setdefault(mode, []).append((ms, i2.clone(), dd.clone()))
    ops.knn_set_tuning(mode=0)
    same = torch.equal(out[0][0][1], out[0x8000000][0][1]) and torch.equal(out[0][0][2], out[0x8000000][0][2])
    print(f"{name}: shipped {out[0][0][0]:.3f} / {out[0][1][0]:.3f} ms   ring4 {out[0x8000000][0][0]:.3f} / {out[0x8000000][1][0]:.3f} ms   results identical: {same}")

g = torch.Generator().manual_seed(10)
N = M = 100_000; k = 20
X = torch.randn(N, 256, generator=g).to(torch.bfloat16).cuda(); Y = torch.randn(M, 256, generator=g).to(torch.bfloat16).cuda()
idx, _ = ops.raw_knn_match(X, Y, k)
ab(X, Y, k, idx, "100k x 100k x 256, structureless")
loop = make_synthetic_c3(style_density="matched", lists="exact")
for _ in range(6):
    loop.step()
torch.cuda.synchronize()
ab(loop.FX, loop.FY, loop.k, loop.nn_idx.clone(), "C3 rows (1M x 65536 x 256), unpruned warm sweep", row_order=loop.row_order)
