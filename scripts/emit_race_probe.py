"""Round 5: does the emit sweep return the same lists every time?  On the C3 loop's state after n0 iterations, REPS replays of prepare -> plan (with / without the cached
bounds) -> emit sweep, each compared with the cold unpruned sweep of the same rows; prints how many rows differ and where they sit (row block, wave, lane of the sweep
order) for the first few.   usage: emit_race_probe.py [n0=20] [reps=40]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
from wast3d_amd.style_loop import make_synthetic_c3

n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
loop = make_synthetic_c3(style_density="matched", lists="exact", timers=False)
for _ in range(n0):
    loop.step()
torch.cuda.synchronize()
N, M, d, k = loop.N, loop.M, loop.d, loop.k
ref_i, ref_d = ops.raw_knn_match(loop.FX, loop.FY, k)
print("loop's own lists equal the cold sweep:", bool(torch.equal(ref_i, loop.nn_idx)))
pos_of = torch.empty(N, dtype=torch.long, device="cuda"); pos_of[loop.row_order.long()] = torch.arange(N, device="cuda")
idx = torch.empty_like(ref_i); d2 = torch.empty_like(ref_d)
bad_runs = 0
for name, b in (("bounds", loop.bounds), ("ball only", None)):
    for r in range(reps):
        ops.raw_knn_prepare(loop.FX, loop.FY, k, prev_idx=loop.nn_idx, ws=loop.kws, row_order=loop.row_order)
        ops.raw_knn_prune_plan(loop.FX, k, loop.prune, loop.kws, row_order=loop.row_order, bounds=b)
        ops.raw_knn_match_prepared(loop.FX, loop.FY, k, idx, d2, 4, ws=loop.kws, row_order=loop.row_order, bounds=b, prune=loop.prune)
        torch.cuda.synchronize()
        diff = (idx != ref_i).any(1)
        nbad = int(diff.sum())
        if nbad:
            bad_runs += 1
            rows = torch.nonzero(diff).flatten()[:6]
            det = []
            for rw in rows.tolist():
                p = int(pos_of[rw]); a = set(idx[rw].tolist()); c = set(ref_i[rw].tolist())
                det.append(f"row {rw} pos {p} (block {p // 512}, wave {(p % 512) // 64}, lane {p % 64}): missing {sorted(c - a)} extra {sorted(a - c)} short {int((idx[rw] >= 0x3ffffff).sum())}")
            print(f"[{name}] replay {r}: {nbad} rows differ;", " | ".join(det))
print("replays with differences:", bad_runs, "of", 2 * reps)
