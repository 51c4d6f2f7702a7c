"""The MFMA match (bf16 / f16 / fp8 rows) against the reference's fp32 path on the bench workload's own generator (SURVEY 7 'hard parts':
parity of a reduced-precision match = recall@k + loss tolerance, not bit equality; north_star: 'within a stated fp32 tolerance').

Scale of the problem (CPU simulation of the bench generator, scripts/sim_parity.py): the k-th neighbour is 1.4 % of a row's length away, so the
reference's OWN fp32 matmul-form cdist only reaches recall@20 = 0.992 against an fp64 evaluation of the same rows — that is the ceiling for "agrees
with the fp32 path".  Uncentred bf16 rows reach 0.949 / 2.6 % loss error (round 1's rows); shifted by the style set's mean row (distances unchanged,
stored values 17x smaller) 0.991 / 1.2e-4 against fp32 and 0.996 against fp64 — closer to the truth than the fp32 reference is.

Tolerances, sampled content rows against the full style set (M = 65 536, d = 256, k = 20), both synthetic style patches, after 3 iterations:
    bf16 rows centred on the style mean : recall@20 >= 0.98 and NN loss within 1e-3 of the fp32 path; recall vs fp64 not below the fp32 path's own - 0.01
    f16 rows centred                    : recall@20 >= 0.985, NN loss within 2e-4
    fp8 e4m3 operands (config C5)       : recall@20 >= 0.85,  NN loss within 1.2e-1   (a coarse match by design: straight-through gradient on the f16 rows)
    bf16 rows NOT centred (round 1)     : recorded for comparison, required to be worse than centred
"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu

N, M, D, K = 120_000, 65_536, 256, 20


def _loop(style, dtype, gemm="same", center=True):
    from wast3d_amd.style_loop import make_synthetic_c3
    loop = make_synthetic_c3(N=N, M=M, d=D, k=K, H=64, W=64, style_density=style, feat_dtype=dtype, gemm=gemm, center=center)
    for _ in range(3):                                   # a few iterations: the rows checked are rows the loop has moved, not the initial ones
        loop.step()
    torch.cuda.synchronize()
    return loop


@pytest.mark.parametrize("style", ["matched", "sparse"])
def test_bf16_centred_match_is_within_tolerance_of_fp32_path(style):
    from tests.parity_fp32 import fp32_parity
    r = fp32_parity(_loop(style, torch.bfloat16), n_sample=8192)
    print("bf16 centred", style, r)
    assert r["recall_at_k"] >= 0.98, r
    assert r["loss_rel_err"] <= 1e-3, r
    # against the fp64 evaluation the centred bf16 match is about as good as the fp32 reference itself
    assert r["loss_rel_err_vs_fp64"] <= 1e-3, r


def test_centring_is_what_buys_the_tolerance():
    from tests.parity_fp32 import fp32_parity
    a = fp32_parity(_loop("matched", torch.bfloat16, center=True), n_sample=4096, with_fp64=False)
    b = fp32_parity(_loop("matched", torch.bfloat16, center=False), n_sample=4096, with_fp64=False)
    print("bf16 centred", a, "uncentred", b)
    assert a["recall_at_k"] > b["recall_at_k"] and a["loss_rel_err"] < b["loss_rel_err"]


def test_f16_centred_match(style="matched"):
    from tests.parity_fp32 import fp32_parity
    r = fp32_parity(_loop(style, torch.float16), n_sample=4096, with_fp64=False)
    print("f16 centred", r)
    assert r["recall_at_k"] >= 0.985 and r["loss_rel_err"] <= 2e-4, r


def test_fp8_match_is_coarse_but_bounded(style="matched"):
    from tests.parity_fp32 import fp32_parity
    r = fp32_parity(_loop(style, torch.float16, gemm="fp8"), n_sample=4096, with_fp64=False)
    print("fp8", r)
    assert r["recall_at_k"] >= 0.85 and r["loss_rel_err"] <= 1.2e-1, r


def test_c5_per_gpu_shard_loop():
    """BASELINE config C5 as ONE of its eight GPUs sees it: 625 000 Gaussians (5M / 8), 4 views of 800 x 800 per iteration, f16 feature rows, fp8 e4m3 distance-GEMM with
    per-row E8M0 scales, the whole style set (cached gather).  Four iterations of the loop, then size-independent properties of its last (warm, pruned) sweep:
    it equals a cold sweep of the same quantised rows bit for bit; distances ascend, no index repeats, sampled distances equal a float64 evaluation of the
    dequantised rows; the match is within the stated tolerance of the reference's fp32 path; the repair stayed within its capacity; every parameter is finite."""
    from tests.parity_fp32 import fp32_parity
    from wast3d_amd import ops
    from wast3d_amd.style_loop import make_synthetic_c3
    loop = make_synthetic_c3(N=625_000, M=M, d=D, k=K, H=800, W=800, style_density="matched", lists="exact", feat_dtype=torch.float16, gemm="fp8", views=4)
    for _ in range(4):
        loop.step()
    torch.cuda.synchronize()
    assert loop._warm == 4 and loop.bounds is not None                 # round 5: pruned (plan on the dequantised rows), the emit kernel's fp8 form, cached lower bounds
    idx_c, d2_c = ops.raw_knn_match_fp8(loop.qX, loop.sX, loop.qY, loop.sY, K)
    assert torch.equal(loop.nn_idx, idx_c) and torch.equal(loop.nn_d2, d2_c)
    d2 = loop.nn_d2
    assert bool((d2[:, 1:] >= d2[:, :-1]).all())
    srt = loop.nn_idx.sort(dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all())
    rows = torch.arange(0, loop.N, 997, device="cuda")
    Xd = ops.dequantize_rows_fp8(loop.qX[rows], loop.sX[rows]).double(); Yd = ops.dequantize_rows_fp8(loop.qY, loop.sY).double()
    true = ((Xd[:, None, :] - Yd[loop.nn_idx[rows].long()]) ** 2).sum(-1)
    scale = float((Xd ** 2).sum(1).max() + (Yd ** 2).sum(1).max())
    assert float((true - d2[rows].double()).abs().max()) <= 2e-5 * scale
    full = (Xd ** 2).sum(1, keepdim=True) + (Yd ** 2).sum(1)[None, :] - 2.0 * Xd @ Yd.t()
    kth = torch.topk(full, K, dim=1, largest=False).values[:, -1]
    assert float((kth - d2[rows, -1].double()).abs().max()) <= 2e-5 * scale      # the k-th distance is the true k-th distance of the quantised problem
    r = fp32_parity(loop, n_sample=4096, with_fp64=False)
    print("C5 shard, fp8 operands on f16 rows:", r)
    assert r["recall_at_k"] >= 0.85 and r["loss_rel_err"] <= 1.2e-1, r
    assert loop.repaired_rows_max <= loop.kws.cap
    assert all(bool(torch.isfinite(p.data).all()) for p in loop.params) and bool(torch.isfinite(loop.loss).all())


def test_c4_per_gpu_share_loop():
    """BASELINE config C4 as ONE of its eight GPUs computes it with the cached style gather: 125 000 Gaussians (1M / 8) against ALL 10^6 style rows (15 625 style tiles —
    the plan kernel's constants table is refilled once), bf16 rows, one 800 x 800 view.  Three iterations, then size-independent properties of the last (warm, pruned,
    sliced: 245 row blocks) sweep: it equals a cold sweep of the same rows bit for bit; distances ascend, no index repeats; sampled k-th distances equal a float64
    evaluation over the whole style set; most tiles were pruned; nothing was redone; every parameter is finite."""
    from wast3d_amd import ops
    from wast3d_amd.style_loop import make_synthetic_c3
    Mc4 = 1_000_000
    loop = make_synthetic_c3(N=125_000, M=Mc4, d=D, k=K, H=800, W=800, style_density="matched", lists="exact", timers=True)
    for _ in range(3):
        loop.step()
    torch.cuda.synchronize()
    assert loop._warm == 4 and loop.bounds is not None                 # pruned + sliced, emit selection, cached lower bounds (round 4)
    # cold sweep of the same rows with the same |x|^2 (the loop takes them from its featurisation pass)
    ws = ops.KnnWorkspace()
    assert ops.raw_knn_prepare(loop.FX, loop.FY, K, xnorm=loop.xnorm, ws=ws) == 0
    idx_c = torch.empty_like(loop.nn_idx); d2_c = torch.empty_like(loop.nn_d2)
    ops.raw_knn_match_prepared(loop.FX, loop.FY, K, idx_c, d2_c, 0, ws=ws)
    print("rows whose lists differ from the cold sweep:", int((loop.nn_idx != idx_c).any(1).sum()), "max |d2 - d2_cold|", float((loop.nn_d2 - d2_c).abs().max()))
    assert torch.equal(loop.nn_idx, idx_c) and torch.equal(loop.nn_d2, d2_c)
    d2 = loop.nn_d2
    assert bool((d2[:, 1:] >= d2[:, :-1]).all())
    srt = loop.nn_idx.sort(dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all())
    rows = torch.arange(0, loop.N, 1999, device="cuda")
    Xd = loop.FX[rows].double(); Yd = loop.FY.double()
    full = (Xd ** 2).sum(1, keepdim=True) + (Yd ** 2).sum(1)[None, :] - 2.0 * Xd @ Yd.t()
    kth = torch.topk(full, K, dim=1, largest=False).values[:, -1]
    scale = float((Xd ** 2).sum(1).max() + (Yd ** 2).sum(1).max())
    assert float((kth - d2[rows, -1].double()).abs().max()) <= 2e-5 * scale
    swept = loop.swept_pairs_per_step()[-1] / (((loop.N + 511) // 512) * ((Mc4 + 63) // 64))
    print("C4 share: swept fraction", swept, "kNN ms", loop.knn_ms_per_step()[-1])
    assert swept < 0.6, swept
    assert loop.repaired_rows_max == 0
    assert all(bool(torch.isfinite(p.data).all()) for p in loop.params) and bool(torch.isfinite(loop.loss).all())
