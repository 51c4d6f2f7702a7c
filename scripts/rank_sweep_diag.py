"""Anatomy of a rank's pruned sweep ("per-rank sweep at 8 ranks = its longest block"): the C3 loop's state after n0 iterations, the rows of one
rank's share (a contiguous block of the scene's Z-order) as a problem of its own — bound, plan, sliced sweep with the in-kernel clocks of every part-block.
usage: rank_sweep_diag.py [W=8] [rank=4] [n0=20]   (env KNN_MODE = tuning word)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from wast3d_amd import ops, _lib
from wast3d_amd.style_loop import make_synthetic_c3

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n0 = int(sys.argv[3]) if len(sys.argv) > 3 else 20
loop = make_synthetic_c3(style_density=os.environ.get("STYLE", "matched"), lists="exact", timers=True)
for _ in range(n0):
    loop.step()
torch.cuda.synchronize()
N, M, d, k = loop.N // W, loop.M, loop.d, loop.k
r0 = R * N
FX = loop.FX[r0:r0 + N].contiguous()
prev = loop.nn_idx[r0:r0 + N].contiguous()
order = torch.argsort(prev[:, 0].long() >> 6, stable=True).int()
kws = ops.KnnWorkspace()
idx = torch.empty(N, k, dtype=torch.int32, device="cuda"); d2 = torch.empty(N, k, device="cuda")
nblk, ntiles = (N + 511) // 512, (M + 63) // 64
L = _lib.lib()
mode = int(os.environ.get("KNN_MODE", "0"), 0)
ops.knn_set_tuning(mode=mode)

def sweep(timed=True):
    ops.raw_knn_prepare(FX, loop.FY, k, prev_idx=prev, ws=kws, row_order=order)
    ops.raw_knn_prune_plan(FX, k, loop.prune, kws, row_order=order)
    return ops.raw_knn_match_prepared(FX, loop.FY, k, idx, d2, 3, timed=timed, ws=kws, row_order=order)

for _ in range(3):
    sweep()
torch.cuda.synchronize()
cnt = ops.knn_prune_counts(N, M, d, k, kws).clone().long()
if os.environ.get("DIAG_DEBUG"):
    print("  lists equal the loop's:", bool(torch.equal(idx, prev)), "; d2 max rel diff", float(((d2 - loop.nn_d2[r0:r0 + N]).abs() / loop.nn_d2[r0:r0 + N].clamp_min(1e-12)).max()))
    print("  repairs:", ops.knn_repair_count(FX, loop.FY, k, kws), " counts head", cnt[:16].tolist())
print(f"rank {R} of {W}: {N} rows = {nblk} row blocks; swept fraction {float(cnt.sum()) / (nblk * ntiles):.4f}; tiles per block min {int(cnt.min())} mean {float(cnt.float().mean()):.1f} max {int(cnt.max())}")
print(f"  ideal: {float(cnt.sum()) / 256:.1f} tiles per CU")
evs = [sweep() for _ in range(5)]
torch.cuda.synchronize()
print("  sweep launch (sweep + merge + repair chain), HIP events:", [round(a.elapsed_time(b), 3) for a, b in evs])
nclk = nblk * 8
clk = torch.zeros(nclk * 8, dtype=torch.int64, device="cuda")
_lib.check(L.w3d_knn_mfma_debug_clocks(C.c_void_p(clk.data_ptr()), None))
ev = sweep()
torch.cuda.synchronize()
_lib.check(L.w3d_knn_mfma_debug_clocks(None, None))
c = clk.view(-1, 8).cpu()
c = c[c[:, 1] > 0]
t0, t1, hw, nt = c[:, 0], c[:, 1], c[:, 2], c[:, 3]
dur = (t1 - t0).double() / 100.0                                       # us
span = (t1.max() - t0.min()).item() / 100e3
print(f"  with clocks: {ev[0].elapsed_time(ev[1]):.3f} ms by events; {c.shape[0]} part-blocks; first start -> last end {span:.3f} ms; sum of durations / 256 = {float(dur.sum()) / 256e3:.3f} ms")
print(f"  part-blocks: tiles min {int(nt.min())} mean {float(nt.double().mean()):.1f} max {int(nt.max())}; duration us min {float(dur.min()):.1f} mean {float(dur.mean()):.1f} max {float(dur.max()):.1f}")
pro, loopd, wout = (c[:, 6] - t0).double() / 100.0, (c[:, 7] - c[:, 6]).double() / 100.0, (t1 - c[:, 7]).double() / 100.0
print(f"  per part-block: prologue {float(pro.mean()):.1f} us (max {float(pro.max()):.1f}), tile loop {float(loopd.mean()):.1f}, write-out {float(wout.mean()):.1f} (max {float(wout.max()):.1f})")
Al = torch.stack([nt.double(), torch.ones_like(dur)], 1)
sl = torch.linalg.lstsq(Al, loopd[:, None]).solution.flatten()
print(f"  tile loop = {float(sl[1]):.1f} us + {float(sl[0]):.3f} us per tile")
A = torch.stack([nt.double(), torch.ones_like(dur)], 1)
sol = torch.linalg.lstsq(A, dur[:, None]).solution.flatten()
print(f"  duration = {float(sol[1]):.1f} us + {float(sol[0]):.3f} us per tile (least squares)")
start_rel = (t0 - t0.min()).double() / 100e3; end_rel = (t1 - t0.min()).double() / 100e3
grid = torch.linspace(0, span, 21, dtype=torch.float64)
running = ((start_rel[None, :] <= grid[:, None]) & (end_rel[None, :] > grid[:, None])).sum(1)
print("  running part-blocks at 0,5,...,100 % of the span:", running.tolist())
cu = hw & 0xffffffff; xcc = (hw >> 32) & 0xf
key = xcc * 1024 + ((cu >> 13) & 7) * 64 + ((cu >> 12) & 1) * 16 + ((cu >> 8) & 0xf)
busy = torch.zeros(int(key.max()) + 1, dtype=torch.float64).index_add_(0, key, dur); busy = busy[busy > 0]
print(f"  CUs used {busy.numel()}; per-CU busy ms: mean {float(busy.mean()) / 1e3:.3f} min {float(busy.min()) / 1e3:.3f} max {float(busy.max()) / 1e3:.3f}")
# the full-chip figure for comparison
km = loop.knn_ms_per_step(); km = sum(km) / len(km) if isinstance(km, (list, tuple)) else km
print(f"  1M-row loop: kNN {km:.2f} ms per step -> 1/{W} = {km / W:.2f} ms")
