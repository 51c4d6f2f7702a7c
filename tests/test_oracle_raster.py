"""Pins for the oracle's rasteriser restatement (oracle/raster_ref.c).  CPU only.
The reference rasteriser is CUDA-only (unbuildable here), so the pins are: the reference's Python fallbacks for
SH->RGB and covariance (golden sh_cov_cam.npz), the reference's camera matrices, closed-form cases, and an
independent fp64 autograd statement of the forward (tests/raster_torch_ref.py) for every gradient."""
import math

import numpy as np
import pytest
import torch

from oracle import raster_ref as RR
from tests import raster_torch_ref as TR
from tests.scene_util import make_scene, np_
from wast3d_amd.camera import Camera, getProjectionMatrix, getWorld2View2


def run_oracle(s, deg=3, mod=1.0, cov3D=None):
    r = RR.Raster(np_(s["bg"]), np_(s["means3D"]), np_(s["opacities"]), np_(s["viewmatrix"]), np_(s["projmatrix"]), np_(s["campos"]),
                  s["tan_fovx"], s["tan_fovy"], s["H"], s["W"], np_(s["offsets"]), shs=np_(s.get("shs")), sh_degree=deg if "shs" in s else 0,
                  colors_precomp=np_(s.get("colors")), scales=None if cov3D is not None else np_(s["scales"]),
                  rotations=None if cov3D is not None else np_(s["rotations"]), cov3D_precomp=cov3D, scale_modifier=mod)
    return r, r.forward()


def test_camera_matrices_match_reference(golden):
    z = golden("sh_cov_cam")
    cam = Camera(z["cam_R"], z["cam_T"], float(z["fovx"]), float(z["fovy"]), 800, 800, device="cpu")
    assert np.abs(cam.world_view_transform.numpy() - z["world_view_transform"]).max() < 1e-6
    assert np.abs(cam.projection_matrix.numpy() - z["projection_matrix"]).max() < 1e-6
    assert np.abs(cam.full_proj_transform.numpy() - z["full_proj_transform"]).max() < 1e-5
    assert np.abs(cam.camera_center.numpy() - z["camera_center"]).max() < 1e-5


def test_sh_and_cov3d_match_reference_python_fallbacks(golden):
    z = golden("sh_cov_cam")
    P = z["xyz"].shape[0]
    cam = Camera(z["cam_R"], z["cam_T"], float(z["fovx"]), float(z["fovy"]), 64, 64, device="cpu")
    # the SH direction uses campos only; put every point in front of the camera so that all are preprocessed
    for deg in range(4):
        r = RR.Raster(np.zeros(3, np.float32), z["xyz"], np.full((P, 1), 0.5, np.float32), np_(cam.world_view_transform), np_(cam.full_proj_transform),
                      z["campos"], 0.36, 0.36, 64, 64, np.zeros((64, 64, 2), np.float32), shs=z["sh"], sh_degree=deg, scales=z["scaling"],
                      rotations=z["rot"] / np.linalg.norm(z["rot"], axis=1, keepdims=True), scale_modifier=1.3)
        _, _, radii = r.forward()
        vis = radii > 0
        assert vis.sum() > P // 4
        rgb = r.geom("rgb", (P, 3))
        assert np.abs(rgb[vis] - z[f"rgb_deg{deg}"][vis]).max() < 2e-6
        cov = r.geom("cov3D", (P, 6))
        assert np.abs(cov - z["cov6_mod1p3"]).max() / np.abs(z["cov6_mod1p3"]).max() < 1e-6


def test_single_gaussian_closed_form():
    """One isotropic Gaussian on the optical axis: alpha(p) = o*exp(-|p-c|^2/(2 s^2)), colour = a*c + (1-a)*bg, depth = a*z."""
    H = W = 33
    cam = Camera.look_at_origin(0.0, 0.0, 5.0, 0.8, W, H, device="cpu")
    sig_world, op, col = 0.1, 0.8, np.array([[0.2, 0.6, 0.9]], np.float32)
    r = RR.Raster(np.array([0.1, 0.2, 0.3], np.float32), np.zeros((1, 3), np.float32), np.array([[op]], np.float32), np_(cam.world_view_transform),
                  np_(cam.full_proj_transform), np_(cam.camera_center), math.tan(0.4), math.tan(0.4), H, W, np.zeros((H, W, 2), np.float32),
                  colors_precomp=col, scales=np.full((1, 3), sig_world, np.float32), rotations=np.array([[1, 0, 0, 0]], np.float32))
    color, depth, radii = r.forward()
    f = W / (2 * math.tan(0.4))
    var = (sig_world * f / 5.0) ** 2 + 0.3
    yy, xx = np.mgrid[0:H, 0:W]
    cx = cy = (W - 1) / 2.0
    a = np.minimum(0.99, op * np.exp(-0.5 * ((xx - cx) ** 2 + (yy - cy) ** 2) / var))
    a[a < 1 / 255.0] = 0
    inside = (np.abs(xx - cx) <= 16 + radii[0]) & (np.abs(yy - cy) <= 16 + radii[0])   # tiles touched cover the image here
    exp_c = a[None] * col.reshape(3, 1, 1) + (1 - a[None]) * np.array([0.1, 0.2, 0.3]).reshape(3, 1, 1)
    assert radii[0] == math.ceil(3 * math.sqrt(var + math.sqrt(0.1)))      # forward.cu:229-232: lambda = mid + sqrt(max(0.1, mid^2 - det))
    assert np.abs(color - exp_c)[:, inside].max() < 2e-5
    assert np.abs(depth - a * 5.0)[inside].max() < 1e-4


@pytest.mark.parametrize("sh,mod,seed", [(True, 1.0, 0), (False, 1.0, 1), (True, 0.7, 2)])
def test_forward_and_gradients_vs_fp64_autograd(sh, mod, seed):
    s = make_scene(P=150, H=37, W=50, seed=seed, sh=sh)       # H, W not multiples of 16: ragged edge tiles
    r, (color, depth, radii) = run_oracle(s, mod=mod)
    P, H, W = s["P"], s["H"], s["W"]
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    plist = r.geom("point_list", (max(r.R, 1),), np.uint32)
    ranges = r.geom("ranges", (tiles, 2), np.uint32)
    dd = lambda t: t.double().clone().requires_grad_(True)
    means, op, sc, rot = dd(s["means3D"]), dd(s["opacities"][:, 0]), dd(s["scales"]), dd(s["rotations"])
    feat = dd(s["shs"]) if sh else dd(s["colors"])
    c64, d64 = TR.render(means, op, s["viewmatrix"].double().reshape(-1), s["projmatrix"].double().reshape(-1), s["campos"].double(), s["bg"].double(),
                         s["tan_fovx"], s["tan_fovy"], H, W, s["offsets"].double(), plist, ranges, radii > 0,
                         shs=feat if sh else None, deg=3, colors=None if sh else feat, scales=sc, rots=rot, mod=mod)
    assert np.abs(color - c64.detach().numpy()).max() < 5e-5
    assert np.abs(depth - d64.detach().numpy()).max() < 2e-4
    g = torch.Generator().manual_seed(100 + seed)
    gc = torch.randn(3, H, W, generator=g).double(); gd = torch.randn(H, W, generator=g).double()
    ((c64 * gc).sum() + (d64 * gd).sum()).backward()
    out = r.backward(gc.numpy(), gd.numpy())

    def close(name, got, ref, tol=2e-3):
        ref = ref.numpy() if torch.is_tensor(ref) else ref
        scale = np.abs(ref).max() + 1e-12
        assert np.abs(got - ref).max() / scale < tol, (name, np.abs(got - ref).max() / scale)
    close("means3D", out["dL_dmeans3D"], means.grad)
    close("opacity", out["dL_dopacity"][:, 0], op.grad)
    close("rot", out["dL_drots"], rot.grad)
    # the reference returns d/d(mod*scale) without the modifier factor (backward.cu:323-325): autograd is w.r.t. scale
    close("scale", out["dL_dscales"] * mod, sc.grad)
    if sh:
        close("sh", out["dL_dsh"], feat.grad)
    else:
        close("colors", out["dL_dcolors"], feat.grad)


def test_precomputed_cov3d_path_and_culling():
    s = make_scene(P=120, H=32, W=32, seed=5, sh=False)
    s["means3D"][:10] += torch.tensor([50.0, 0, 0])           # far outside the frustum sides (still in front)
    s["means3D"][10:20] = s["campos"] + 0.05                  # behind the near plane (p_view.z <= 0.2)
    r1, (c1, d1, rad1) = run_oracle(s)
    cov = r1.geom("cov3D", (s["P"], 6))
    r2, (c2, d2, rad2) = run_oracle(s, cov3D=cov)
    assert (rad1[10:20] == 0).all() and (rad1 > 0).sum() > 50
    vis = rad1 > 0
    assert np.array_equal(rad1[vis], rad2[vis]) and np.abs(c1 - c2).max() < 1e-6
    present = RR.mark_visible(np_(s["means3D"]), np_(s["viewmatrix"]), np_(s["projmatrix"]))
    assert (~present[10:20]).all() and present[20:].all()


def test_empty_scene_and_dense_overlap():
    s = make_scene(P=1, H=20, W=20, seed=6, sh=False)
    s["means3D"][:] = s["campos"] + 0.01                      # nothing visible
    r, (c, d, rad) = run_oracle(s)
    assert r.R == 0 and np.allclose(c, np_(s["bg"]).reshape(3, 1, 1)) and np.all(d == 0)
    # many opaque splats: early termination (T < 1e-4) must leave final_T > 0 and n_contrib < list length
    s = make_scene(P=400, H=32, W=32, seed=7, sh=False, scale=0.5, opacity=(0.9, 0.99), spread=0.3)
    r, (c, d, rad) = run_oracle(s)
    nc = r.geom("n_contrib", (32 * 32,), np.uint32); ranges = r.geom("ranges", (4, 2), np.uint32)
    assert (nc < (ranges[:, 1] - ranges[:, 0]).max()).any() and r.geom("final_T", (32 * 32,)).min() >= 1e-4 * 0.01
