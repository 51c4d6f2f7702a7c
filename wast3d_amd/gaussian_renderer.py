"""Drop-in for gaussian_renderer/__init__.py::render (lines 18-115) on the HIP rasteriser.

One deliberate extension: `sampling_offsets` may be passed explicitly.  The reference draws
`torch.rand(H, W, 2) * -1` inside render() on every call (line 31), which makes renders irreproducible; with
`sampling_offsets=None` this function does exactly that, with a tensor it is deterministic."""
from __future__ import annotations

import math

import torch

from .diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from .sh_utils import eval_sh


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None, sampling_offsets=None):
    dev = pc.get_xyz.device
    screenspace_points = torch.zeros_like(pc.get_xyz, dtype=pc.get_xyz.dtype, requires_grad=True, device=dev) + 0
    cam_view_depth = torch.zeros(pc.get_xyz.shape[:-1] + (1,), dtype=pc.get_xyz.dtype, requires_grad=True, device=dev) + 0   # dead input, kept for parity
    if sampling_offsets is None:
        sampling_offsets = torch.rand(viewpoint_camera.image_height, viewpoint_camera.image_width, 2, device=dev) * -1
    try:
        screenspace_points.retain_grad()
        cam_view_depth.retain_grad()
    except Exception:
        pass
    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width), tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center, prefiltered=False, debug=getattr(pipe, "debug", False))
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    means3D = pc.get_xyz
    means2D = screenspace_points
    opacity = pc.get_opacity
    scales = rotations = cov3D_precomp = None
    if getattr(pipe, "compute_cov3D_python", False):
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales = pc.get_scaling
        rotations = pc.get_rotation
    shs = colors_precomp = None
    if override_color is None:
        if getattr(pipe, "convert_SHs_python", False):
            shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = (pc.get_xyz - viewpoint_camera.camera_center.repeat(pc.get_features.shape[0], 1))
            dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            sh2rgb = eval_sh(pc.active_sh_degree, shs_view, dir_pp_normalized)
            colors_precomp = torch.clamp_min(sh2rgb + 0.5, 0.0)
        else:
            shs = pc.get_features
    else:
        colors_precomp = override_color
    rendered_image, rendered_depth, radii = rasterizer(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp, opacities=opacity,
                                                       scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp, cam_view_depth=cam_view_depth,
                                                       sampling_offsets=sampling_offsets)
    return {"render": rendered_image, "depth": rendered_depth, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}
