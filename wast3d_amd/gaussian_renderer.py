"""`render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color)` on the HIP rasteriser.

Interface contract (what callers of gaussian_renderer/__init__.py:18-115 rely on): the argument list above, and a dict with
`render` [3,H,W], `depth` [H,W], `viewspace_points` [P,3] (a leaf whose `.grad` receives dL/dmeans2D), `visibility_filter` [P] bool
and `radii` [P] int32.  The body is written against that contract and this package's rasteriser, not after the reference's body:
the reference's own `render()` can equally stay in place and import `wast3d_amd.diff_gaussian_rasterization`.

Extension: `sampling_offsets` [H,W,2] may be passed.  The reference draws `-U[0,1)` offsets on every call (line 31), which makes
renders irreproducible; None does the same here, a tensor makes the call deterministic.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from .diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from .sh_utils import eval_sh


def _settings(cam, pc, pipe, bg_color, scaling_modifier) -> GaussianRasterizationSettings:
    return GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width),
        tanfovx=math.tan(0.5 * cam.FoVx), tanfovy=math.tan(0.5 * cam.FoVy),
        bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
        sh_degree=pc.active_sh_degree, campos=cam.camera_center,
        prefiltered=False, debug=bool(getattr(pipe, "debug", False)))


def _colour_inputs(cam, pc, pipe, override_color):
    """(shs, colors_precomp): exactly one is not None.  `pipe.convert_SHs_python` evaluates the SH basis with torch (the reference's Python
    fallback, pinned by tests/golden/sh_cov_cam.npz) instead of inside the projection kernel."""
    if override_color is not None:
        return None, override_color
    if not getattr(pipe, "convert_SHs_python", False):
        return pc.get_features, None
    n_coeff = (pc.max_sh_degree + 1) ** 2
    coeffs = pc.get_features.transpose(1, 2).reshape(-1, 3, n_coeff)
    view_dir = torch.nn.functional.normalize(pc.get_xyz - cam.camera_center.reshape(1, 3), dim=1)
    return None, (eval_sh(pc.active_sh_degree, coeffs, view_dir) + 0.5).clamp_min(0.0)


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None, sampling_offsets: Optional[torch.Tensor] = None):
    cam = viewpoint_camera
    xyz = pc.get_xyz
    # the handle through which callers read dL/d(screen-space mean): a fresh leaf per call
    viewspace_points = torch.zeros_like(xyz).requires_grad_(True)
    if sampling_offsets is None:
        sampling_offsets = -torch.rand(int(cam.image_height), int(cam.image_width), 2, device=xyz.device)
    shs, colors_precomp = _colour_inputs(cam, pc, pipe, override_color)
    geometry = ({"cov3D_precomp": pc.get_covariance(scaling_modifier)} if getattr(pipe, "compute_cov3D_python", False)
                else {"scales": pc.get_scaling, "rotations": pc.get_rotation})
    rasterizer = GaussianRasterizer(raster_settings=_settings(cam, pc, pipe, bg_color, scaling_modifier))
    image, depth, radii = rasterizer(means3D=xyz, means2D=viewspace_points, opacities=pc.get_opacity, shs=shs, colors_precomp=colors_precomp,
                                     sampling_offsets=sampling_offsets, **geometry)
    return {"render": image, "depth": depth, "viewspace_points": viewspace_points, "visibility_filter": radii > 0, "radii": radii}
