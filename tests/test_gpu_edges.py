"""Edge cases of the hot-path entry points on the GPU: empty inputs, the smallest and largest supported sizes, k = M, rows that are all equal,
argument errors reported as error codes (W3dError) rather than crashes.  Run with `-m gpu` on an MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from wast3d_amd import ops as o
    return o


def _ref_topk(X, Y, k):
    D = (X.double()[:, None, :] - Y.double()[None, :, :]).pow(2).sum(-1)
    v, i = torch.topk(D, k, dim=1, largest=False)
    return v, i, D


@pytest.mark.parametrize("N,M,k", [(1, 1, 1), (1, 20, 20), (3, 5, 5), (64, 32, 32), (513, 65, 1), (2, 4097, 32)])
def test_knn_mfma_extreme_shapes(ops, N, M, k):
    """One row, one style row, k = M (every style row selected), k = 1, the largest k whose per-row lists fit the LDS at d = 64 (32): the selected SET and its squared
    distances against an f64 brute force on the same bf16 values."""
    g = torch.Generator().manual_seed(N * 1000 + M + k)
    X = torch.randn(N, 64, generator=g).to(torch.bfloat16).cuda(); Y = torch.randn(M, 64, generator=g).to(torch.bfloat16).cuda()
    idx, d2 = ops.raw_knn_match(X, Y, k)
    v, i, D = _ref_topk(X.float().cpu(), Y.float().cpu(), k)
    got = torch.gather(D, 1, idx.cpu().long())
    assert (idx >= 0).all() and (idx < M).all()
    assert all(len(set(r.tolist())) == k for r in idx.cpu())                       # k distinct neighbours per row
    assert torch.allclose(got.sort(1).values, v, rtol=0, atol=1e-3 * float(D.max()) + 1e-6)     # the same k distances as the brute force
    assert torch.allclose(d2.cpu().double(), got, rtol=0, atol=2e-3 * float(D.max()) + 1e-5)   # reported distance = distance of the reported index
    assert (d2[:, 1:] >= d2[:, :-1]).all()                                          # ascending, as torch.topk(largest=False) returns them


def test_knn_rejects_unsupported_arguments(ops):
    from wast3d_amd._lib import W3dError
    X = torch.randn(8, 64).to(torch.bfloat16).cuda(); Y = torch.randn(16, 64).to(torch.bfloat16).cuda()
    with pytest.raises(W3dError):
        ops.raw_knn_match(X, Y, 17)                                                 # k > M
    with pytest.raises(W3dError):
        ops.raw_knn_match(X, Y, 0)
    with pytest.raises(W3dError):
        ops.raw_knn_match(torch.randn(8, 96).to(torch.bfloat16).cuda(), torch.randn(16, 96).to(torch.bfloat16).cuda(), 4)   # d not in {64,128,256} on the MFMA path
    Yb = torch.randn(200, 64).to(torch.bfloat16).cuda()
    with pytest.raises(W3dError):
        ops.raw_knn_match(X, Yb, 33)                                                # per-row lists no longer fit the LDS (d = 64: k <= 32)
    X2 = torch.randn(8, 256).to(torch.bfloat16).cuda(); Y2 = torch.randn(200, 256).to(torch.bfloat16).cuda()
    ops.raw_knn_match(X2, Y2, 24)                                                   # d = 256: k <= 24
    with pytest.raises(W3dError):
        ops.raw_knn_match(X2, Y2, 25)


def test_empty_inputs_are_no_ops(ops):
    """Zero rows / zero Gaussians: every entry point returns without touching its outputs (the reference's torch ops accept empty tensors)."""
    z3 = torch.zeros(0, 3, device="cuda")
    X = torch.randn(10, 3, device="cuda")
    idx0 = torch.zeros(0, 5, dtype=torch.int32, device="cuda")
    F = ops.raw_descriptor_fwd(X, idx0)
    assert F.shape == (0, 4)
    gX = torch.zeros_like(X)
    ops.raw_descriptor_bwd(X, idx0, torch.zeros(0, 4, device="cuda"), gX)
    assert (gX == 0).all()
    Fv = ops.get_descriptors_vec(X, torch.zeros(0, 4, device="cuda"), idx0)
    assert Fv.shape == (0, 16)
    # Adam on an empty group next to a non-empty one
    p = torch.randn(7, device="cuda"); g = torch.randn(7, device="cuda"); m = torch.zeros(7, device="cuda"); v = torch.zeros(7, device="cuda")
    e = torch.zeros(0, device="cuda")
    p0 = p.clone()
    ops.raw_adam_step([(e, e.clone(), e.clone(), e.clone(), 1e-3), (p, g, m, v, 1e-3)], 1, 0.9, 0.999, 1e-8, zero_grad=False)
    assert torch.allclose(p, p0 - 1e-3 * torch.sign(g), atol=1e-6)                  # first Adam step = lr * sign(g)
    assert z3.numel() == 0


def test_rasteriser_no_visible_gaussians_and_single_pixel_image():
    """All Gaussians behind the camera: background image, zero tile instances, zero gradients.  A 1x1 image: one tile, one pixel."""
    import math
    from wast3d_amd.camera import Camera
    from wast3d_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    for (H, W, z) in ((33, 47, -10.0), (1, 1, 0.0)):
        cam = Camera.look_at_origin(0.0, 0.0, 4.0, 0.7, W, H, device="cuda")
        P = 50
        g = torch.Generator().manual_seed(3)
        means = (torch.randn(P, 3, generator=g) * 0.05).cuda()
        if z != 0.0:
            means = means + cam.camera_center[None, :] * 3.0                        # three camera distances from the origin, beyond the camera: behind it
        means.requires_grad_(True)
        bg = torch.tensor([0.2, 0.4, 0.6], device="cuda")
        rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2), bg=bg, scale_modifier=1.0,
                                           viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=0, campos=cam.camera_center,
                                           prefiltered=False, debug=False)
        col = torch.rand(P, 3, generator=g).cuda().requires_grad_(True)
        opa = torch.full((P, 1), 0.5, device="cuda", requires_grad=True)
        sc = torch.full((P, 3), 0.05, device="cuda", requires_grad=True)
        rot = torch.tensor([[1.0, 0, 0, 0]], device="cuda").repeat(P, 1).requires_grad_(True)
        m2d = torch.zeros(P, 3, device="cuda", requires_grad=True)
        offs = -torch.rand(H, W, 2, generator=g).cuda()
        color, depth, radii = GaussianRasterizer(rs)(means3D=means, means2D=m2d, shs=None, colors_precomp=col, opacities=opa, scales=sc, rotations=rot,
                                                     cov3D_precomp=None, cam_view_depth=None, sampling_offsets=offs)
        (color.sum() + depth.sum()).backward()
        assert color.shape == (3, H, W) and depth.shape[-2:] == (H, W) and torch.isfinite(color).all()
        if z != 0.0:
            assert (radii == 0).all()
            assert torch.allclose(color, bg[:, None, None].expand(3, H, W))
            assert (means.grad == 0).all() and (col.grad == 0).all() and (opa.grad == 0).all()
        else:
            assert (radii > 0).any() and torch.isfinite(means.grad).all()


@pytest.mark.parametrize("N,M,d,k,dt", [(1, 64, 64, 1, torch.bfloat16), (65, 100, 128, 3, torch.float16), (700, 1000, 64, 32, torch.bfloat16), (700, 1000, 128, 24, torch.float16),
                                        (513, 64 * 8 + 1, 64, 24, torch.bfloat16), (5000, 20000, 128, 20, torch.bfloat16)])
def test_emit_sweep_edge_shapes(N, M, d, k, dt):
    """Round 5 — the emit kernel (warm = 4) at the edges: a single row, fewer style rows than two tiles, a style set one row past a tile boundary, k = 1 and the largest
    k the rank sort takes (32: the select kernel's general path then has no room for its chunk and sends an overflowing region to the repair), a NaN row, duplicate rows,
    with and without cached bounds: the lists equal the cold sweep's bit for bit."""
    from wast3d_amd import ops
    g = torch.Generator().manual_seed(N * 7 + k)
    Xf = torch.randn(N, d, generator=g); Yf = torch.randn(M, d, generator=g)
    if N > 10:
        Xf[3] = Xf[2]                                                    # duplicate content rows
        Xf[7] = float("nan")
    if M > 80:
        Yf[70] = Yf[5]                                                   # duplicate style rows: an exact tie, ordered by index
    X = Xf.to(dt).cuda(); Y = Yf.to(dt).cuda()
    ref_i, ref_d = ops.raw_knn_match(X, Y, k)
    pr = ops.KnnPrune(Y)
    for use_bounds in (False, True):
        ws = ops.KnnWorkspace()
        bounds = ops.KnnBounds(N, M, "cuda") if use_bounds else None
        prev = ref_i
        for it in range(3):
            ops.raw_knn_prepare(X, Y, k, prev_idx=prev, ws=ws)
            ops.raw_knn_prune_plan(X, k, pr, ws, bounds=bounds)
            oi = torch.empty_like(ref_i); od = torch.empty_like(ref_d)
            ops.raw_knn_match_prepared(X, Y, k, oi, od, 4, ws=ws, bounds=bounds, prune=pr if use_bounds else None)
            assert torch.equal(oi, ref_i) and torch.equal(od, ref_d), (use_bounds, it, int((oi != ref_i).sum()))
            if bounds is not None:
                bounds.advance(torch.zeros(N, device="cuda"))


def test_lean_adam_equals_the_plain_kernel_bit_for_bit():
    """Round 5: raw_adam_step(lean=True) — the 32-VGPR form of the Adam kernel that runs under the persistent emit sweep (one launch per group, float2 per thread) — does the
    arithmetic of the plain kernel: parameters, both moments and the zeroed gradients are bit-equal, with a second gradient buffer, odd sizes, sizes under one pair."""
    from wast3d_amd import ops
    g = torch.Generator().manual_seed(12)
    sizes = [3 * 100_001, 45 * 7001, 4, 1, 2 * 65_537 + 1]
    def make():
        gg = torch.Generator().manual_seed(13)
        out = []
        for i, n in enumerate(sizes):
            p = torch.randn(n, generator=gg).cuda(); gr = torch.randn(n, generator=gg).cuda(); m = (0.1 * torch.randn(n, generator=gg)).cuda(); v = (0.01 * torch.rand(n, generator=gg)).cuda()
            g2 = torch.randn(n, generator=gg).cuda() if i % 2 == 0 else None
            out.append((p, gr, m, v, 1e-3 * (i + 1), g2))
        return out
    a, b = make(), make()
    for step in (1, 2, 7):
        ops.raw_adam_step(a, step, 0.9, 0.999, 1e-15, zero_grad=(step == 2))
        ops.raw_adam_step(b, step, 0.9, 0.999, 1e-15, zero_grad=(step == 2), lean=True)
        torch.cuda.synchronize()
        for ga, gb in zip(a, b):
            for ta, tb in zip(ga, gb):
                if torch.is_tensor(ta):
                    assert torch.equal(ta, tb), step
