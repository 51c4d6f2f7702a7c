"""Which iterations of the C3 loop send rows through the cold repair, and is the result still what a cold sweep returns?  At such an iteration the sweep is replayed from
the saved previous neighbours in several forms to see which part is responsible.  usage: repair_trace.py [iterations=700] [start_checking=400]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
from wast3d_amd.style_loop import make_synthetic_c3
n = int(sys.argv[1]) if len(sys.argv) > 1 else 700
start = int(sys.argv[2]) if len(sys.argv) > 2 else 400
loop = make_synthetic_c3(style_density="matched", lists="exact")
N, M, d, k = loop.N, loop.M, loop.d, loop.k
idx = torch.empty_like(loop.nn_idx); d2 = torch.empty_like(loop.nn_d2)
ws = ops.KnnWorkspace()
done = 0
for it in range(n):
    prev = loop.nn_idx.clone() if it >= start else None
    order = loop.row_order
    loop.step()
    c, cap = ops.knn_repair_count(loop.FX, loop.FY, k, ws=loop.kws)
    if not c or prev is None:
        continue
    idx_c, d2_c = ops.raw_knn_match(loop.FX, loop.FY, k)
    bad = (loop.nn_idx.sort(1).values != idx_c.sort(1).values).any(1)
    print(f"iteration {it}: {c} rows redone cold; loop result differs from a cold sweep in {int(bad.sum())} rows (as sets)", flush=True)
    def replay(tag, pruned, mode=0, ro=order):
        ops.knn_set_tuning(mode=mode)
        w = ops.raw_knn_prepare(loop.FX, loop.FY, k, prev_idx=prev, ws=ws, row_order=ro)
        if pruned:
            ops.raw_knn_prune_plan(loop.FX, k, loop.prune, ws, row_order=ro); w = 3
        ops.raw_knn_match_prepared(loop.FX, loop.FY, k, idx, d2, w, ws=ws, row_order=ro)
        torch.cuda.synchronize()
        cc = ops.knn_repair_count(loop.FX, loop.FY, k, ws=ws)[0]
        b = (idx.sort(1).values != idx_c.sort(1).values).any(1)
        print(f"    replay [{tag}]: {cc} rows redone, {int(b.sum())} rows differ from the cold sweep" + (f" (rows {torch.nonzero(b).flatten()[:6].tolist()})" if int(b.sum()) else ""), flush=True)
        ops.knn_set_tuning(mode=0)
    Np = (N + 63) // 64 * 64
    thr_loop = loop.kws.t[Np * 4: Np * 4 + N * 4].view(torch.float32).clone()
    xn_loop = loop.kws.t[: N * 4].view(torch.float32).clone()
    cnt_loop = ops.knn_prune_counts(N, M, d, k, loop.kws).clone()
    replay("pruned, default", True)
    thr_re = ws.t[Np * 4: Np * 4 + N * 4].view(torch.float32)
    xn_re = ws.t[: N * 4].view(torch.float32)
    cnt_re = ops.knn_prune_counts(N, M, d, k, ws)
    print(f"    workspace of the loop's sweep vs the replay's: |x|^2 equal {bool(torch.equal(xn_loop, xn_re))}, bounds equal {bool(torch.equal(thr_loop, thr_re))} "
          f"(rows with a different bound: {int((thr_loop != thr_re).sum())}, loop tighter in {int((thr_loop > thr_re).sum())}), tile counts equal {bool(torch.equal(cnt_loop, cnt_re))}", flush=True)
    if not torch.equal(thr_loop, thr_re):
        r = torch.nonzero(thr_loop != thr_re).flatten()[:5]
        print("      rows", r.tolist(), "loop", thr_loop[r].tolist(), "replay", thr_re[r].tolist(), "differing-result rows among them:", int(bad[torch.nonzero(thr_loop != thr_re).flatten()].sum()), flush=True)
    replay("pruned, no row order", True, ro=None)
    replay("pruned, one workgroup per row block", True, mode=0x8000)
    replay("pruned, lock step", True, mode=0x4000000)
    replay("unpruned warm", False)
    done += 1
    if done >= 3:
        break
