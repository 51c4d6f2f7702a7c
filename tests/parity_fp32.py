"""Checker (test infrastructure): the MFMA match on 16-bit / fp8 feature rows against the reference's fp32 path.

The reference computes the match in fp32: descriptors (NB11:994-1009) -> torch.cdist (matmul form, NB11:1016) -> k-th-neighbour mask
(NB11:1018-1020) -> mean of the selected squared distances (NB11:1021).  The oracle restates exactly that on UNROUNDED fp32 rows; the product
path featurises the same points with its HIP kernels (centred, rounded to bf16 / f16, optionally quantised to fp8) and matches them on the matrix
cores.  Reported for a random sample of content rows against the FULL style set:
    recall_at_k   mean over rows of |kNN_hip ∩ kNN_fp32| / k
    loss_rel_err  |sum d2_hip - sum d2_fp32| / sum d2_fp32   (the NN loss restricted to the sampled rows)
and, for scale, the same two figures for the fp32 reference itself against an fp64 evaluation of the same rows (the reference's own noise floor:
its matmul form cancels |x|^2 + |y|^2 - 2 x.y with neighbours ~50x closer than the rows are long).
Used by tests/test_gpu_parity_fp32.py and by bench.py's cpu_baseline leg (where the oracle's run is also the timed CPU sample).
"""
import time

import numpy as np
import torch

from oracle import geometry_ref as O


def hip_match_of_current_points(loop):
    """The product path's features + cold match for the loop's CURRENT points (does not touch the loop's own state)."""
    from wast3d_amd import ops
    xyz = loop.pc._xyz.data
    X4 = ops.raw_pack_xyz4(xyz)
    FX = torch.empty_like(loop.FX)
    if loop.csr is None:
        ops.raw_descriptor_fwd_window(X4, loop.idx, FX, center=loop.center)
    else:
        ops.raw_descriptor_fwd_x4(X4, loop.idx, FX, center=loop.center)
    if loop.gemm == "fp8":
        qX, sX = ops.quantize_rows_fp8(FX); qY, sY = ops.quantize_rows_fp8(loop.FY)
        idx, d2 = ops.raw_knn_match_fp8(qX, sX, qY, sY, loop.k)
    else:
        idx, d2 = ops.raw_knn_match(FX, loop.FY, loop.k)
    torch.cuda.synchronize()
    return idx, d2


def fp32_parity(loop, n_sample=16384, seed=0, chunk=4096, with_fp64=True):
    """-> dict(recall_at_k, loss_rel_err, ref_recall_vs_fp64, ref_loss_rel_err_vs_fp64, rows, seconds_oracle_knn)."""
    k = loop.k
    idx_h, d2_h = hip_match_of_current_points(loop)
    X = loop.pc._xyz.data.cpu().numpy().astype(np.float32)
    lists = loop.idx.cpu().numpy()
    N = X.shape[0]
    rng = np.random.default_rng(seed)
    rows = np.sort(rng.choice(N, size=min(n_sample, N), replace=False))
    FXo = O.get_descriptors(X, lists[rows].astype(np.int64))                                    # fp32, unrounded, uncentred: the reference's rows
    FYo = O.get_descriptors(loop.style_xyz.cpu().numpy().astype(np.float32), loop.style_idx.cpu().numpy().astype(np.int64))
    ih = idx_h[torch.as_tensor(rows, device=idx_h.device)].long()
    if getattr(loop, "style_perm", None) is not None:                                           # the loop stores the style rows in its own order
        ih = loop.style_perm[ih]
    ih = ih.cpu().numpy().astype(np.int64)
    dh = d2_h[torch.as_tensor(rows, device=d2_h.device)].cpu().numpy().astype(np.float64)
    hit = 0; s_ref = 0.0; hit64 = 0; s64 = 0.0
    t_knn = 0.0
    FY64 = FYo.astype(np.float64) if with_fp64 else None
    yn64 = (FY64 ** 2).sum(1) if with_fp64 else None
    for c0 in range(0, len(rows), chunk):
        sl = slice(c0, min(c0 + chunk, len(rows)))
        t0 = time.perf_counter()
        _, _, ir, vr = O.knn_match_loss(FXo[sl], FYo, k, partition=True)                        # the reference's fp32 path on these rows
        t_knn += time.perf_counter() - t0
        s_ref += float(vr.astype(np.float64).sum())
        hit += sum(len(np.intersect1d(a, b, assume_unique=True)) for a, b in zip(ih[sl], ir))
        if with_fp64:
            x64 = FXo[sl].astype(np.float64)
            d64 = (x64 ** 2).sum(1)[:, None] + yn64[None, :] - 2.0 * (x64 @ FY64.T)
            p = np.argpartition(d64, k - 1, axis=1)[:, :k]
            s64 += float(np.take_along_axis(d64, p, 1).sum())
            hit64 += sum(len(np.intersect1d(a, b, assume_unique=True)) for a, b in zip(ir, p))
    n = len(rows)
    out = {"rows": int(n), "k": int(k), "recall_at_k": hit / (n * k), "loss_rel_err": abs(float(dh.sum()) - s_ref) / s_ref, "seconds_oracle_knn": t_knn}
    if with_fp64:
        out["ref_fp32_recall_vs_fp64"] = hit64 / (n * k)
        out["ref_fp32_loss_rel_err_vs_fp64"] = abs(s_ref - s64) / s64
        out["loss_rel_err_vs_fp64"] = abs(float(dh.sum()) - s64) / s64
    return out
