"""What the kNN kernel's GEMM skeleton waits for (needs a -DW3D_KNN_DEBUG library: W3D_LIB_PATH=scripts/_ab/libwast3d_dbg.so).  Debug mode bits switch parts off
(results are wrong, only the time is read): 1 selection epilogue, 256 tile barrier, 64 LDS-DMA of the next tile, 0x4000 no priority raise, 0x2000 no barrier skew."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
from wast3d_amd.style_loop import make_synthetic_c3

def t_knn(FX, FY, k, prev, mode, reps=4, row_order=None):
    ops.knn_set_tuning(mode=mode)
    ws = ops.KnnWorkspace()
    idx = torch.empty(FX.shape[0], k, dtype=torch.int32, device="cuda"); d2 = torch.empty(FX.shape[0], k, device="cuda")
    ts = []
    for _ in range(reps):
        w = ops.raw_knn_prepare(FX, FY, k, prev_idx=prev, ws=ws, row_order=row_order)
        e = ops.raw_knn_match_prepared(FX, FY, k, idx, d2, w, timed=True, ws=ws, row_order=row_order)
        torch.cuda.synchronize(); ts.append(e[0].elapsed_time(e[1]))
    ops.knn_set_tuning(mode=0)
    return min(ts)

loop = make_synthetic_c3(style_density="matched", center=True, lists="knn")
for _ in range(4):
    loop.step()
torch.cuda.synchronize()
FX, FY, prev, ro = loop.FX.clone(), loop.FY, loop.nn_idx.clone(), loop.row_order
for name, mode in (("full", 0), ("no selection", 1), ("no selection, no barrier", 1 | 256), ("no selection, no DMA", 1 | 64), ("no selection, no barrier, no DMA", 1 | 64 | 256),
                   ("full, no barrier", 256), ("no selection, no setprio", 1 | 0x4000), ("full again", 0)):
    print(f"{name:40s} {t_knn(FX, FY, loop.k, prev, mode, row_order=ro):.2f} ms", flush=True)
