"""Round 5: where a wave of the emit sweep spends its cycles — shader-clock stamps at the phase boundaries of every tile (a -DW3D_KNN_STAMP build of the library, selected
with W3D_LIB_PATH): per wave the sums over a row block of M0 / E0 / M1 / E1 (MFMA phase and selection epilogue of the tile's two 32-row halves), the wait at the tile
barrier, by wave group (A = waves 0-3, B = waves 4-7).  The stamps themselves cost ~10 % (s_memtime + lgkmcnt(0)): proportions, not absolute times.
usage: W3D_LIB_PATH=<stamp build> emit_stamp.py [n0=20]"""
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
nblk = (N + 511) // 512
L = _lib.lib()
clk = torch.zeros(nblk * 8 + 2 * nblk * 64, dtype=torch.int64, device="cuda")
idx = torch.empty_like(loop.nn_idx); d2 = torch.empty_like(loop.nn_d2)
for name, b in (("ball test + cached bounds", loop.bounds), ("ball test only", None)):
    for rep in range(2):
        ops.raw_knn_prepare(loop.FX, loop.FY, k, prev_idx=loop.nn_idx, ws=loop.kws, row_order=loop.row_order)
        ops.raw_knn_prune_plan(loop.FX, k, loop.prune, loop.kws, row_order=loop.row_order, bounds=b)
        if rep == 1:
            _lib.check(L.w3d_knn_mfma_debug_clocks(_lib.ptr(clk), None)); clk.zero_()
        ops.raw_knn_match_prepared(loop.FX, loop.FY, k, idx, d2, 4, ws=loop.kws, row_order=loop.row_order, bounds=b, prune=loop.prune)
    torch.cuda.synchronize()
    _lib.check(L.w3d_knn_mfma_debug_clocks(None, None))
    assert torch.equal(idx.sort(1).values, loop.nn_idx.sort(1).values)
    st = clk[nblk * 8:nblk * 8 + nblk * 64].view(nblk, 8, 8).cpu().double()
    sq = clk[nblk * 8 + nblk * 64:].view(nblk, 8, 8).cpu().double()
    tiles = st[:, :, 6].clamp_min(1)
    per = st[:, :, :6] / tiles[:, :, None]                     # cycles per tile, per (block, wave)
    blk = clk[: nblk * 8].view(nblk, 8).cpu().double()
    cyc_per_us = ((blk[:, 5] - blk[:, 4]) / ((blk[:, 1] - blk[:, 0]) / 100.0)).median()
    print(f"[{name}] in-kernel clock {cyc_per_us:.0f} cycles/us; candidates per wave and tile {float((st[:, :, 7] / tiles).mean()):.2f}")
    for g, sl in (("A (waves 0-3)", slice(0, 4)), ("B (waves 4-7)", slice(4, 8))):
        m = per[:, sl].mean((0, 1))
        print(f"   {g}: cycles per tile  M0 {m[0]:.0f}  E0 {m[1]:.0f}  M1 {m[2]:.0f}  E1 {m[3]:.0f}  barrier wait {m[4]:.0f}  sum {m[:5].sum():.0f}")
    big = st[:, 0, 6] >= 60
    fl = sq[big][:, :, :4].mean((0, 1)); span = sq[big][:, :, 4].mean(); nt = st[big][:, 0, 6].mean()
    mid = (st[big][:, :, :5].sum(2).mean(1) / st[big][:, 0, 6]).mean()
    print(f"   blocks of >= 60 tiles: whole tiles by position, cycles: first {fl[0]:.0f}, second {fl[1]:.0f}, last but one {fl[2]:.0f}, last {fl[3]:.0f}; mean tile {mid:.0f}; a wave's item {span:.0f} cycles = {nt:.1f} tiles x mean tile + {span - nt * mid:.0f}")
    # by candidate load of the wave
    cand = st[:, :, 7] / tiles
    for lo, hi in ((0, 4), (4, 8), (8, 12), (12, 20), (20, 1e9)):
        msk = (cand >= lo) & (cand < hi)
        if msk.any():
            m = per[msk].mean(0)
            print(f"   waves with {lo}-{hi} candidates per tile ({int(msk.sum())}): M0 {m[0]:.0f}  E0 {m[1]:.0f}  M1 {m[2]:.0f}  E1 {m[3]:.0f}  barrier {m[4]:.0f}")
