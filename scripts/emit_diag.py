"""Round 4: anatomy of the emit sweep under the cached lower bounds, from inside the kernel (w3d_knn_mfma_debug_clocks).  On the C3 loop's own state after n0 iterations
the plan (ball test + bounds, no advance) and the warm = 4 sweep are replayed with the clocks on: per row block the prologue (X fragments), the tile loop and the tail
(last spill), and a fit of the tile loop's time against the number of tiles; the select kernel and the plan kernel alone.
usage: emit_diag.py [n0=20]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops, _lib
from wast3d_amd.style_loop import make_synthetic_c3

n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 20
loop = make_synthetic_c3(style_density="matched", lists="exact", timers=True)
for _ in range(n0):
    loop.step()
torch.cuda.synchronize()
N, M, d, k = loop.N, loop.M, loop.d, loop.k
nblk, T = (N + 511) // 512, (M + 63) // 64
L = _lib.lib()
clk = torch.zeros(nblk * 8 * 8, dtype=torch.int64, device="cuda")
idx = torch.empty_like(loop.nn_idx); d2 = torch.empty_like(loop.nn_d2)


def replay(bounds, timed=True, record=True):
    ops.raw_knn_prepare(loop.FX, loop.FY, k, prev_idx=loop.nn_idx, ws=loop.kws, row_order=loop.row_order)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.raw_knn_prune_plan(loop.FX, k, loop.prune, loop.kws, row_order=loop.row_order, bounds=bounds)
    e1.record()
    ev = ops.raw_knn_match_prepared(loop.FX, loop.FY, k, idx, d2, 4, timed=timed, ws=loop.kws, row_order=loop.row_order, bounds=bounds if record else None, prune=loop.prune)
    return (e0, e1), ev


# (third variant: the plan of the bounds, but the sweep does NOT record new bounds — what the recording itself costs; timing only: the cache is stale afterwards)
for name, b, rec in (("ball test + cached bounds", loop.bounds, True), ("ball test only", None, True), ("cached bounds, sweep without recording", loop.bounds, False)):
    for _ in range(2):
        replay(b, record=rec)
    _lib.check(L.w3d_knn_mfma_debug_clocks(_lib.ptr(clk), None))
    clk.zero_()
    pe, ev = replay(b, record=rec)
    torch.cuda.synchronize()
    _lib.check(L.w3d_knn_mfma_debug_clocks(None, None))
    # (sets, not positions: the replay takes |x|^2 from the norm kernel, the loop from the featurisation pass — two scores may round to one squared distance under one and not the other)
    assert torch.equal(idx.sort(1).values, loop.nn_idx.sort(1).values)
    c = clk.view(-1, 8)[:nblk].cpu().double()
    tiles = c[:, 3]
    tot = (c[:, 1] - c[:, 0]) / 100.0            # us (100-MHz wall clock)
    pro = (c[:, 6] - c[:, 0]) / 100.0
    loop_us = (c[:, 7] - c[:, 6]) / 100.0
    tail = (c[:, 1] - c[:, 7]) / 100.0
    span = float(c[:, 1].max() - c[:, 0].min()) / 100.0
    A = torch.stack([torch.ones_like(tiles), tiles], 1)
    sol = torch.linalg.lstsq(A, loop_us[:, None]).solution.flatten()
    cnt = ops.knn_prune_counts(N, M, d, k, loop.kws).sum().item() / (nblk * T)
    print(f"[{name}] swept {cnt:.4f}: plan kernel {pe[0].elapsed_time(pe[1]):.3f} ms, sweep + select {ev[0].elapsed_time(ev[1]):.3f} ms; persistent launch (first block start -> last block end) {span / 1e3:.3f} ms")
    print(f"   per row block: tiles mean {tiles.mean():.1f}; prologue {pro.mean():.1f} us, tile loop {loop_us.mean():.1f} us, tail {tail.mean():.1f} us, total {tot.mean():.1f} us; "
          f"sum of block times / 256 CUs = {tot.sum() / 256 / 1e3:.3f} ms")
    print(f"   tile loop fit: {sol[0]:.1f} us + {sol[1]:.3f} us per tile")
    for lo, hi in ((0, 50), (50, 100), (100, 200), (200, 400), (400, 1100)):
        m = (tiles >= lo) & (tiles < hi)
        if m.any():
            print(f"      blocks of {lo:4d}-{hi:4d} tiles: {int(m.sum()):5d}, loop us per tile {float((loop_us[m] / tiles[m].clamp_min(1)).mean()):.3f}")
