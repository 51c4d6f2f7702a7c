"""Cold emit sweep at BASELINE config C2's shape: shader-clock stamps per wave (a -DW3D_KNN_STAMP build of the (bf16, d) leaf, selected with W3D_LIB_PATH): cycles per
tile in the MFMA phases / epilogues / at the tile barrier, entries written to the region per row, what is left in the region at the end.
usage: W3D_LIB_PATH=<stamp build> python scripts/cold_stamp.py [N=100000] [M=100000] [d=128]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops, _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
d = int(sys.argv[3]) if len(sys.argv) > 3 else 128
k = 20
if os.environ.get("STAMP_CLUSTERS"):                                # style rows stored cluster by cluster, content rows sorted by cluster: candidates come in storms
    g = torch.Generator().manual_seed(47); nc = int(os.environ["STAMP_CLUSTERS"])
    cen = torch.randn(nc, d, generator=g)
    Y = (cen[torch.arange(M) * nc // M] + 0.05 * torch.randn(M, d, generator=g)).to(torch.bfloat16).cuda()
    X = (cen[torch.randint(0, nc, (N,), generator=g).sort().values] + 0.05 * torch.randn(N, d, generator=g)).to(torch.bfloat16).cuda()
else:
    X = torch.randn(N, d, generator=torch.Generator().manual_seed(10)).to(torch.bfloat16).cuda()
    Y = torch.randn(M, d, generator=torch.Generator().manual_seed(11)).to(torch.bfloat16).cuda()
ws = ops.KnnWorkspace()
nblk = (N + 511) // 512
L = _lib.lib()
clk = torch.zeros(nblk * 8 + 2 * nblk * 64, dtype=torch.int64, device="cuda")
idx = torch.empty(N, k, dtype=torch.int32, device="cuda"); d2 = torch.empty(N, k, device="cuda")
WARM = os.environ.get("STAMP_WARM", "0") == "1"                    # the warm emit sweep of the same rows, one item per row block (tuning bits 28-30 = 1 slice), for comparison
if WARM:
    idx0, _ = ops.raw_knn_match(X, Y, k)
    pr = ops.KnnPrune(Y, meb_iters=10)
    ops.knn_set_tuning(mode=1 << 28)
for rep in range(2):
    if WARM:
        ops.raw_knn_prepare(X, Y, k, prev_idx=idx0, ws=ws)
        ops.raw_knn_prune_plan(X, k, pr, ws)
        w = 4
    else:
        w = ops.raw_knn_prepare(X, Y, k, ws=ws)
    if rep == 1:
        _lib.check(L.w3d_knn_mfma_debug_clocks(_lib.ptr(clk), None)); clk.zero_()
    ops.raw_knn_match_prepared(X, Y, k, idx, d2, w, ws=ws)
torch.cuda.synchronize()
_lib.check(L.w3d_knn_mfma_debug_clocks(None, None))
ops.knn_set_tuning(mode=0)
print("WARM emit sweep (bound from the previous lists), one item per row block" if WARM else "COLD emit sweep")
st = clk[nblk * 8:nblk * 8 + nblk * 64].view(nblk, 8, 8).cpu().double()
sq = clk[nblk * 8 + nblk * 64:].view(nblk, 8, 8).cpu().double()
blk = clk[: nblk * 8].view(nblk, 8).cpu().double()
tiles = st[:, :, 6].clamp_min(1)
cyc_per_us = ((blk[:, 5] - blk[:, 4]) / ((blk[:, 1] - blk[:, 0]) / 100.0)).median()
span_us = (blk[:, 1] - blk[:, 0]) / 100.0
print(f"{N} x {M} x {d}: in-kernel clock {cyc_per_us:.0f} cycles/us; a block's item {span_us.mean():.0f} us (max {span_us.max():.0f}); tiles per item {tiles.mean():.0f}")
print("   rows redone by the repair:", ops.knn_repair_count(X, Y, k, ws=ws), " waves whose region ended past its capacity:", int((st[:, :, 7] > 8192).sum()), "of", st[:, :, 7].numel())
fin = st[:, :, 7].flatten(); wr = st[:, :, 5].flatten()
print("   entries left in a region at the end: max", int(fin.max()), " written to a region: max", int(wr.max()), " tiles:", int(tiles.max()))
print(f"   entries written per row {float(st[:, :, 5].mean()) / 64:.1f} (max wave {float(st[:, :, 5].max()) / 64:.1f}); left in the region at the end per row {float(st[:, :, 7].mean()) / 64:.1f}")
per = st[:, :, :5] / tiles[:, :, None]
for g, sl in (("A (waves 0-3)", slice(0, 4)), ("B (waves 4-7)", slice(4, 8))):
    m = per[:, sl].mean((0, 1))
    print(f"   {g}: cycles per tile  M0 {m[0]:.0f}  E0 {m[1]:.0f}  M1 {m[2]:.0f}  E1 {m[3]:.0f}  barrier wait {m[4]:.0f}  sum {m[:5].sum():.0f}")
fl = sq[:, :, :4].mean((0, 1)); span = sq[:, :, 4].mean(); nt = tiles.mean(); mid = (st[:, :, :5].sum(2) / tiles).mean()
print(f"   whole tiles by position, cycles: first {fl[0]:.0f}, second {fl[1]:.0f}, last but one {fl[2]:.0f}, last {fl[3]:.0f}; mean tile {mid:.0f}; a wave's item {span:.0f} cycles = {nt:.0f} tiles x mean tile + {span - nt * mid:.0f}")
