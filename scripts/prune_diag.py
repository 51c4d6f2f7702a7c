"""Where does the pruned sweep lose against the unpruned one per executed flop?    On the C3 loop's own state after `n0` iterations:
  1. the plan's tile counts per 512-row block: distribution, sum / 256 CUs vs a longest-first list-scheduling makespan (what the hardware dispatcher can reach at best);
  2. per-block clocks from inside the kernel (w3d_knn_mfma_debug_clocks): launch makespan, per-CU busy time, time per tile as a function of block length, the tail;
  3. the pruned and the unpruned sweep alone on the chip (HIP events), same rows;
  4. the exact ball test at other granularities: row groups of 512 / 256 / 128 / 64 rows x style tiles of 64 / 32 rows (what finer pruning could skip).
usage: prune_diag.py [n0=20]   (env W_NN, STYLE)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from wast3d_amd import ops, _lib
from wast3d_amd.style_loop import make_synthetic_c3

n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 20
kw = {}
if os.environ.get("W_NN"):
    kw["w_nn"] = float(os.environ["W_NN"])
loop = make_synthetic_c3(style_density=os.environ.get("STYLE", "matched"), lists="exact", timers=True, **kw)
for _ in range(n0):
    loop.step()
torch.cuda.synchronize()
N, M, d, k = loop.N, loop.M, loop.d, loop.k
nblk, ntiles = (N + 511) // 512, (M + 63) // 64
cnt = ops.knn_prune_counts(N, M, d, k, loop.kws).clone().long()
print(f"after {n0} iterations: swept fraction {float(cnt.sum()) / (nblk * ntiles):.4f}; tiles per block min {int(cnt.min())} median {int(cnt.median())} mean {float(cnt.float().mean()):.1f} "
      f"max {int(cnt.max())} of {ntiles}")
# longest-first list scheduling on 256 CUs (one block per CU)
import heapq
srt = sorted(cnt.tolist(), reverse=True)
h = [0] * 256
heapq.heapify(h)
for c in srt:
    heapq.heappush(h, heapq.heappop(h) + c)
print(f"ideal sum/256 = {sum(srt) / 256:.1f} tiles; longest-first makespan = {max(h)} tiles ({max(h) / (sum(srt) / 256):.4f} x ideal); longest block {srt[0]}")
# static round-robin of blockIdx over 8 XCDs (each XCD then schedules dynamically on its 32 CUs)
xs = [sorted(srt[i::8], reverse=True) for i in range(8)]
worst = 0
for x in xs:
    hh = [0] * 32; heapq.heapify(hh)
    for c in x:
        heapq.heappush(hh, heapq.heappop(hh) + c)
    worst = max(worst, max(hh))
print(f"with blockIdx round-robin over 8 XCDs x 32 CUs: makespan {worst} tiles ({worst / (sum(srt) / 256):.4f} x ideal)")

# ---- 2. clocks from inside the kernel
L = _lib.lib()
clk = torch.zeros(nblk * 8 * 8, dtype=torch.int64, device="cuda")
idx = torch.empty_like(loop.nn_idx); d2 = torch.empty_like(loop.nn_d2)

def pruned_sweep(timed=True):
    w = ops.raw_knn_prepare(loop.FX, loop.FY, k, prev_idx=loop.nn_idx, ws=loop.kws, row_order=loop.row_order)
    ops.raw_knn_prune_plan(loop.FX, k, loop.prune, loop.kws, row_order=loop.row_order)
    return ops.raw_knn_match_prepared(loop.FX, loop.FY, k, idx, d2, 3, timed=timed, ws=loop.kws, row_order=loop.row_order)

def unpruned_sweep():
    w = ops.raw_knn_prepare(loop.FX, loop.FY, k, prev_idx=loop.nn_idx, ws=loop.kws, row_order=loop.row_order)
    return ops.raw_knn_match_prepared(loop.FX, loop.FY, k, idx, d2, w, timed=True, ws=loop.kws, row_order=loop.row_order)

for _ in range(2):
    pruned_sweep()
torch.cuda.synchronize()
ops.knn_set_tuning(mode=int(os.environ.get("DIAG_MODE", "0")))
pclk = torch.zeros(nblk * 4, dtype=torch.int64, device="cuda")
_lib.check(L.w3d_knn_mfma_debug_clocks(C.c_void_p(clk.data_ptr()), C.c_void_p(pclk.data_ptr())))
ev = pruned_sweep()
torch.cuda.synchronize()
_lib.check(L.w3d_knn_mfma_debug_clocks(None, None))
ms = ev[0].elapsed_time(ev[1])
cnt2 = ops.knn_prune_counts(N, M, d, k, loop.kws).clone().long()
c = clk.view(-1, 8)[:nblk].cpu()
t0, t1, hw, nt = c[:, 0], c[:, 1], c[:, 2], c[:, 3]
ghz = ((c[:, 5] - c[:, 4]).double() / (c[:, 1] - c[:, 0]).double().clamp_min(1) * 0.1)
print(f"in-kernel clock of the pruned sweep (shader clock / wall clock per row block): median {float(ghz.median()):.3f} GHz, min {float(ghz.min()):.3f}, max {float(ghz.max()):.3f}")
span = (t1.max() - t0.min()).item() / 100e3                          # 100-MHz clock -> ms
dur = (t1 - t0).double() / 100e3
xcc = (hw >> 32) & 0xf
cu = (hw & 0xffffffff)
cu_id = ((cu >> 8) & 0xf) | (((cu >> 13) & 0x7) << 4)              # HW_ID: CU_ID bits 11:8, SE_ID bits 15:13 (SH_ID bit 12)
sh = (cu >> 12) & 1
key = xcc * 1024 + ((cu >> 13) & 7) * 64 + sh * 16 + ((cu >> 8) & 0xf)
print(f"pruned sweep with clocks: {ms:.3f} ms by events; first start -> last end {span:.3f} ms; blocks {nblk}; distinct (xcc, se, sh, cu) = {key.unique().numel()}; swept {float(cnt2.sum()) / (nblk * ntiles):.4f}")
busy = torch.zeros(int(key.max()) + 1, dtype=torch.float64).index_add_(0, key, dur)
busy = busy[busy > 0]
print(f"per-CU busy: mean {busy.mean():.3f} ms, min {busy.min():.3f}, max {busy.max():.3f} (of {span:.3f}); sum of block durations / (CUs x span) = {float(dur.sum()) / (busy.numel() * span):.4f}")
per_tile = dur * 1e3 / nt.double().clamp_min(1)                       # us per tile
for lo, hi in ((1, 100), (100, 200), (200, 400), (400, 600), (600, 800), (800, 1025)):
    m = (nt >= lo) & (nt < hi)
    if m.any():
        print(f"  blocks with {lo:4d}-{hi:4d} tiles: {int(m.sum()):5d} blocks, {float(per_tile[m].mean()):.3f} us per tile (min {float(per_tile[m].min()):.3f}, max {float(per_tile[m].max()):.3f})")
end_rel = (t1 - t0.min()).double() / 100e3
last = torch.sort(end_rel, descending=True).values[:8]
print("  last 8 block ends (ms):", [round(float(x), 3) for x in last])
start_rel = (t0 - t0.min()).double() / 100e3
# how much of the chip is idle in the tail: integrate (256 - running blocks) over the last 10 % of the span
grid = torch.linspace(0, span, 201, dtype=torch.float64)
running = ((start_rel[None, :] <= grid[:, None]) & (end_rel[None, :] > grid[:, None])).sum(1)
print("  running blocks at 0,10,...,100 % of the span:", running[::20].tolist())
idle = float((256 - running.double()).clamp_min(0).mean()) / 256
print(f"  mean idle CU fraction over the launch: {idle:.4f}")
xb = torch.zeros(8, dtype=torch.float64).index_add_(0, xcc, dur)
print("  busy ms per XCD:", [round(float(v), 2) for v in xb])

# ---- 2a. the same kernel with a plan that keeps EVERY tile (radii blown up): the in-kernel clock of a full sweep, and its time per tile
import copy
big = copy.copy(loop.prune); big.cradius = loop.prune.cradius * 1e6
_lib.check(L.w3d_knn_mfma_debug_clocks(C.c_void_p(clk.data_ptr()), None))
for _ in range(2):
    ops.raw_knn_prepare(loop.FX, loop.FY, k, prev_idx=loop.nn_idx, ws=loop.kws, row_order=loop.row_order)
    ops.raw_knn_prune_plan(loop.FX, k, big, loop.kws, row_order=loop.row_order)
    evf = ops.raw_knn_match_prepared(loop.FX, loop.FY, k, idx, d2, 3, timed=True, ws=loop.kws, row_order=loop.row_order)
torch.cuda.synchronize()
_lib.check(L.w3d_knn_mfma_debug_clocks(None, None))
cf = clk.view(-1, 8)[:nblk].cpu()
ghz_f = ((cf[:, 5] - cf[:, 4]).double() / (cf[:, 1] - cf[:, 0]).double().clamp_min(1) * 0.1)
ptf = (cf[:, 1] - cf[:, 0]).double() / 100.0 / cf[:, 3].double().clamp_min(1)
print(f"full sweep through the same kernel (every tile kept): {evf[0].elapsed_time(evf[1]):.3f} ms; in-kernel clock median {float(ghz_f.median()):.3f} GHz; {float(ptf.median()):.3f} us per tile "
      f"= {float(ptf.median()) * float(ghz_f.median()) * 1e3:.0f} shader cycles per tile (256 MFMAs of 16 cycles per SIMD = 4096)")

# ---- 2b. the plan kernel's blocks: prologue (X fragments, constants), centre loop, compaction
pc = pclk.view(-1, 4).cpu().double() / 100.0                          # us
print(f"plan kernel: span {float(pc[:, 3].max() - pc[:, 0].min()) / 1e3:.3f} ms; per block: prologue {float((pc[:, 1] - pc[:, 0]).mean()):.1f} us, centre loop {float((pc[:, 2] - pc[:, 1]).mean()):.1f} us "
      f"({float((pc[:, 2] - pc[:, 1]).mean()) / ((ntiles + 63) // 64):.2f} us per group of 64 centres), compaction {float((pc[:, 3] - pc[:, 2]).mean()):.1f} us, whole block {float((pc[:, 3] - pc[:, 0]).mean()):.1f} us")

# ---- 3. alone on the chip
evs = [pruned_sweep() for _ in range(5)]
torch.cuda.synchronize()
pm = sorted(a.elapsed_time(b) for a, b in evs)[2]
evs = [unpruned_sweep() for _ in range(3)]
torch.cuda.synchronize()
um = sorted(a.elapsed_time(b) for a, b in evs)[1]
ops.knn_set_tuning(mode=0x4000000)
evs = [unpruned_sweep() for _ in range(3)]
torch.cuda.synchronize()
print(f"unpruned sweep in lock step (tuning bit 0x4000000): {sorted(a.elapsed_time(b) for a, b in evs)[1]:.3f} ms; out of phase: {um:.3f} ms")
ops.knn_set_tuning(mode=0)
fl = 2.0 * N * M * d
sw = float(cnt2.sum()) / (nblk * ntiles)
print(f"alone: pruned sweep {pm:.3f} ms = {fl * sw / pm / 1e9:.0f} TFLOP/s executed ({fl * sw / pm / 1e9 / 2500:.3f} of peak); unpruned {um:.3f} ms = {fl / um / 1e9:.0f} TFLOP/s ({fl / um / 1e9 / 2500:.3f})")

# ---- 3b. launch forms of the pruned sweep (tuning word: 0x8000 = one workgroup per row block; bits 24-25 = queue order of the persistent launch)
for tag, mode in (("persistent, waves of a SIMD out of phase (default)", 0), ("persistent, lock step (round 2 timing)", 0x4000000), ("persistent, lock step, no priorities", 0x4004000), ("one workgroup per row block, out of phase", 0x8000)):
    ops.knn_set_tuning(mode=mode)
    evs = [pruned_sweep() for _ in range(5)]
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    assert torch.equal(idx, loop.nn_idx)
    print(f"launch form [{tag}]: median {t[2]:.3f} ms (min {t[0]:.3f}) = {fl * sw / t[2] / 1e9:.0f} TFLOP/s executed ({fl * sw / t[2] / 1e9 / 2500:.3f} of peak)", flush=True)
ops.knn_set_tuning(mode=0)

# ---- 4. ball test at other granularities (the plan's own geometry: centres / radii of ops.KnnPrune for 64-row tiles; 32-row half tiles built the same way)
FX = loop.FX.float(); FY = loop.FY.float()
thr = None
D = loop.nn_d2[:, -1].clamp_min(0).sqrt() * 1.0001
ro = loop.row_order.long() if loop.row_order is not None else torch.arange(N, device="cuda")
Nc = N // 512 * 512
ro = ro[:Nc]

def tiles(ts):
    T = M // ts
    Yt = FY[: T * ts].view(T, ts, -1)
    c = Yt.mean(1)
    ar = torch.arange(T, device=c.device)
    for i in range(100):
        far = (Yt - c[:, None]).norm(dim=2).argmax(1)
        c = c + (Yt[ar, far] - c) / (i + 2)
    return c, (Yt - c[:, None]).norm(dim=2).max(1).values

for ts in (64, 32):
    c, r = tiles(ts)
    T = c.shape[0]
    tot = {512: 0, 256: 0, 128: 0, 64: 0}
    # work actually executed if tiles were decided per group g: sum over groups of (tiles needed) * g rows
    for b0 in range(0, Nc, 32768):
        ii = ro[b0:b0 + 32768]
        need = (torch.cdist(FX[ii], c) - r[None, :]) <= D[ii][:, None]
        for g in tot:
            tot[g] += int(need.view(-1, g, T).any(1).sum()) * g
    print(f"tile = {ts} rows (median radius {float(r.median()):.4g}): swept fraction if the plan decided per group of " +
          ", ".join(f"{g} rows: {tot[g] / (Nc * T):.3f}" for g in tot))

