"""Seeded synthetic scenes for the rasteriser tests (inputs only)."""
import math

import numpy as np
import torch

from wast3d_amd.camera import Camera


def make_scene(P=200, H=40, W=56, seed=0, sh=True, scale=0.06, opacity=(0.2, 0.95), spread=1.0, fovx=0.9):
    g = torch.Generator().manual_seed(seed)
    cam = Camera.look_at_origin(0.7, 0.5, 4.0, fovx, W, H, device="cpu")
    means = (torch.rand(P, 3, generator=g) * 2 - 1) * spread
    shs = torch.randn(P, 16, 3, generator=g) * 0.3
    shs[:, 0] += 0.5
    colors = torch.rand(P, 3, generator=g)
    op = torch.rand(P, 1, generator=g) * (opacity[1] - opacity[0]) + opacity[0]
    scales = torch.exp(torch.randn(P, 3, generator=g) * 0.4) * scale
    rots = torch.nn.functional.normalize(torch.randn(P, 4, generator=g), dim=1)
    offsets = -torch.rand(H, W, 2, generator=g)                       # gaussian_renderer/__init__.py:31
    bg = torch.tensor([1.0, 0.5, 0.25])
    d = dict(P=P, H=H, W=W, bg=bg, means3D=means, opacities=op, scales=scales, rotations=rots, offsets=offsets,
             viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, campos=cam.camera_center,
             tan_fovx=math.tan(cam.FoVx * 0.5), tan_fovy=math.tan(cam.FoVy * 0.5))
    if sh:
        d["shs"] = shs
    else:
        d["colors"] = colors
    return d


def np_(t):
    return None if t is None else t.detach().cpu().numpy().astype(np.float32)
