"""Where does a sweep tile's time go at d = 128 (BASELINE config C2: 100k x 100k x 128, warm sweeps)?  In-kernel ablation switches of a -DW3D_KNN_DEBUG build
(W3D_LIB_PATH=scripts/_ab/libwast3d_hip_dbg.so): 1 = no selection epilogue, 64 = the next tile is never fetched (no LDS-DMA), 256 = no tile barrier.  Results are
wrong under any of them; only the times matter.  usage: W3D_LIB_PATH=... python scripts/c2_ablate.py [d=128]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
d = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = M = 100_000; k = 20
X = torch.randn(N, d, generator=torch.Generator().manual_seed(10)).to(torch.bfloat16).cuda()
Y = torch.randn(M, d, generator=torch.Generator().manual_seed(11)).to(torch.bfloat16).cuda()
ws = ops.KnnWorkspace()
idx, d2 = ops.raw_knn_match(X, Y, k)
def warm(mode):
    ops.knn_set_tuning(mode=mode)
    i2 = torch.empty_like(idx); dd = torch.empty_like(d2); evs = []
    for _ in range(5):
        w = ops.raw_knn_prepare(X, Y, k, prev_idx=idx, ws=ws)
        evs.append(ops.raw_knn_match_prepared(X, Y, k, i2, dd, w, timed=True, ws=ws))
    torch.cuda.synchronize()
    ops.knn_set_tuning(mode=0)
    return sorted(a.elapsed_time(b) for a, b in evs)[2]
flop = 2.0 * N * M * d
for name, mode in (("as shipped", 0), ("no selection (1)", 1), ("no DMA (64)", 64), ("no barrier (256)", 256), ("no selection, no DMA (65)", 65), ("no selection, no DMA, no barrier (321)", 321),
                   ("one launch, unsplit (16)", 16), ("unsplit, no selection, no DMA, no barrier", 16 | 321),
                   ("LOCK STEP (0x4000000): as shipped", 0x4000000), ("lock step, no selection, no DMA", 0x4000000 | 65), ("lock step, no selection, no DMA, no barrier", 0x4000000 | 321),
                   ("32-row tiles (8): as shipped", 8), ("32-row tiles, no selection, no DMA", 8 | 65), ("32-row tiles, no selection, no DMA, no barrier", 8 | 321),
                   ("no MFMA priorities (0x4000), no selection, no DMA", 0x4000 | 65), ("no MFMA priorities, no selection, no DMA, no barrier", 0x4000 | 321)):
    ms = warm(mode)
    print(f"d = {d}  {name:45s} {ms:7.3f} ms  {flop / ms / 1e9:7.0f} TFLOP/s = {flop / ms / 1e9 / 2500:.3f} of peak")
