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
