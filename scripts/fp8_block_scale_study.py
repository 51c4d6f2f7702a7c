"""CPU study for VERDICT r05 item 6: would the fp8 distance-GEMM's recall against the reference's fp32 path improve with the block-scaled MFMA's NATIVE scales — one E8M0
scale per 32 elements of a row (what v_mfma_scale_f32_32x32x64_f8f6f4 consumes per k-block) — instead of one per row (csrc/quantize.hip)?

The C3 rows at reduced size, built on the CPU exactly as make_synthetic_c3 builds them (uniform points at the content's density, descriptors = distances to the exact 256
nearest neighbours, style patch of the same density, rows shifted by the style set's mean row): recall@20 and NN-loss error of the exact kNN of the QUANTISED rows against the
fp32 path, for bf16 / f16 rounding, e4m3 with a scale per row, per 32 elements and per 16 elements, and (as the ceiling of any scaling scheme) e4m3 with an ideal real-valued
scale per ELEMENT position... all through float64 distances of the dequantised rows, so that only the operand format differs.

    python scripts/fp8_block_scale_study.py [N_content] [M_style]
"""
import sys

import numpy as np
import torch
from scipy.spatial import cKDTree

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
d, k = 256, 20
rng = np.random.default_rng(20)
side = 2.6 * (N / 1e6) ** (1 / 3)                      # the C3 scene's density: 1e6 points in [-1.3, 1.3]^3
X = rng.uniform(0, side, (N, 3)).astype(np.float32)
Y = rng.uniform(0, side * (M / N) ** (1 / 3), (M, 3)).astype(np.float32)


def descriptors(P):
    t = cKDTree(P)
    dist, _ = t.query(P, k=d + 1)
    return dist[:, 1:].astype(np.float32)               # sorted distances to the 256 nearest neighbours (get_descriptors on exact lists, NB11:994-1009)


FX32, FY32 = descriptors(X), descriptors(Y)
# interior rows only (the box's faces distort the descriptors of both sets alike, but the sample should look like the scene's bulk)
mu = FY32.mean(0)


def knn(A, B):
    A = A.astype(np.float64); B = B.astype(np.float64)
    D = (A * A).sum(1)[:, None] + (B * B).sum(1)[None, :] - 2.0 * A @ B.T
    idx = np.argpartition(D, k - 1, axis=1)[:, :k]
    return idx, np.take_along_axis(D, idx, 1).sum()


sample = rng.choice(N, 6000, replace=False)
ref_idx, ref_sum = knn(FX32[sample], FY32)


def report(name, A, B):
    idx, s = knn(A[sample], B)
    rec = np.mean([len(np.intersect1d(a, b)) for a, b in zip(idx, ref_idx)]) / k
    print(f"{name:58s} recall@20 {rec:.4f}   NN-loss rel err {abs(s - ref_sum) / ref_sum:.2e}")


def e4m3(x):
    return torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(torch.float8_e4m3fn).to(torch.float32).numpy()


def quant(F, block):
    """e4m3 with one power-of-two scale per `block` consecutive elements of a row (block = d: the per-row scale of csrc/quantize.hip): the smallest power of two that
    brings the block into e4m3's range (max 448), round to nearest even — dequantised values."""
    R = F.reshape(F.shape[0], -1, block)
    amax = np.abs(R).max(2, keepdims=True)
    e = np.ceil(np.log2(np.maximum(amax, 1e-30) / 448.0))
    s = np.exp2(e)
    return (e4m3(R / s) * s).reshape(F.shape)


cX, cY = FX32 - mu, FY32 - mu
report("fp32 rows, float64 distances (the reference's sets)", FX32, FY32)
report("bf16, centred (the headline's operands)", torch.from_numpy(cX).to(torch.bfloat16).float().numpy(), torch.from_numpy(cY).to(torch.bfloat16).float().numpy())
h = lambda a: torch.from_numpy(a).to(torch.float16).float().numpy()
report("f16, centred (C5's feature rows)", h(cX), h(cY))
for blk in (256, 64, 32, 16):
    report(f"e4m3 of the f16 rows, one E8M0 scale per {blk:3d} elements", quant(h(cX), blk), quant(h(cY), blk))
# what limits any per-block power-of-two scaling: e4m3 keeps 4 significant bits (relative step 2^-4 ... 2^-3) on EVERY element, whatever its scale
rel = np.abs(quant(h(cX), 32) - h(cX)).sum() / np.abs(h(cX)).sum()
print(f"mean relative rounding error of an element, block 32: {rel:.3f}; k-th / (k+1)-th squared-distance gap of the fp32 sets, median relative: ", end="")
A = FX32[sample].astype(np.float64); B = FY32.astype(np.float64)
D = np.sort((A * A).sum(1)[:, None] + (B * B).sum(1)[None, :] - 2.0 * A @ B.T, axis=1)
print(f"{np.median((D[:, k] - D[:, k - 1]) / D[:, k - 1]):.4f}")
print("spread of |element| along a centred style row (90th / 10th percentile of |value|), median over rows:", float(np.median(np.percentile(np.abs(cY), 90, axis=1) / np.maximum(np.percentile(np.abs(cY), 10, axis=1), 1e-12))))
