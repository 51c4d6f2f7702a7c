"""Where the kNN kernel's time goes, by data set (needs a -DW3D_KNN_DEBUG library: W3D_LIB_PATH=scripts/_ab/libwast3d_dbg.so):
full kernel (warm, previous neighbours as the bound) vs the same sweep with the selection compiled out at run time (mode bit 1 = GEMM skeleton + pretest-free
epilogue), on the feature rows of the C3 loop after a few iterations — hard / easy style patch, centred / uncentred rows."""
import os, sys, time
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

for style in ("matched", "sparse"):
    for center in (True, False):
        loop = make_synthetic_c3(style_density=style, center=center, lists="knn")
        for _ in range(4):
            loop.step()
        torch.cuda.synchronize()
        FX, FY, prev, ro = loop.FX.clone(), loop.FY, loop.nn_idx.clone(), loop.row_order
        full = t_knn(FX, FY, loop.k, prev, 0, row_order=ro)
        skel = t_knn(FX, FY, loop.k, prev, 1, row_order=ro)          # bit 1: selection epilogue skipped
        scan = t_knn(FX, FY, loop.k, prev, 2, row_order=ro)          # bit 2: queue never merged (flush = reset): scan + append cost only
        rnd = torch.randn_like(FX.float()).to(FX.dtype)
        skel_rnd = t_knn(rnd, torch.randn_like(FY.float()).to(FY.dtype), loop.k, None, 1)
        print(f"{style} centred={center}: full {full:.2f} ms  skeleton {skel:.2f}  no-merge {scan:.2f}  skeleton on random-normal rows {skel_rnd:.2f}", flush=True)
        del loop; torch.cuda.empty_cache()
