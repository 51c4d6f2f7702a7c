"""Camera matrices in the reference's conventions (row-vector / transposed 4x4s):
utils/graphics_utils.py:38-71 (getWorld2View2, getProjectionMatrix), scene/cameras.py:48-57 (Camera),
scene/cameras.py:59-70 (MiniCam).  Host-side float maths only; no kernels."""
from __future__ import annotations

import math

import numpy as np
import torch


def getWorld2View2(R, t, translate=np.array([0.0, 0.0, 0.0]), scale=1.0):
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    C2W[:3, 3] = (C2W[:3, 3] + translate) * scale
    return np.float32(np.linalg.inv(C2W))


def getProjectionMatrix(znear, zfar, fovX, fovY):
    ty = math.tan(fovY / 2); tx = math.tan(fovX / 2)
    top, right = ty * znear, tx * znear
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (2 * right)
    P[1, 1] = 2.0 * znear / (2 * top)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


class Camera:
    """The fields render() reads from the reference's Camera/MiniCam: image_height/width, FoVx/FoVy,
    world_view_transform, full_proj_transform, camera_center."""

    def __init__(self, R, T, FoVx, FoVy, width, height, znear=0.01, zfar=100.0, device="cuda"):
        self.R, self.T, self.FoVx, self.FoVy = R, T, FoVx, FoVy
        self.image_width, self.image_height, self.znear, self.zfar = int(width), int(height), znear, zfar
        wvt = torch.tensor(getWorld2View2(R, T)).transpose(0, 1)
        proj = getProjectionMatrix(znear, zfar, FoVx, FoVy).transpose(0, 1)
        self.world_view_transform = wvt.to(device)
        self.projection_matrix = proj.to(device)
        self.full_proj_transform = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).to(device)
        self.camera_center = wvt.inverse()[3, :3].to(device)

    @staticmethod
    def from_c2w_blender(c2w, fovx, width, height, device="cuda"):
        """nerf_synthetic convention (scene/dataset_readers.py:199-213): flip Y/Z of the OpenGL camera-to-world."""
        c2w = np.array(c2w, dtype=np.float64).copy()
        c2w[:3, 1:3] *= -1
        w2c = np.linalg.inv(c2w)
        R = np.transpose(w2c[:3, :3]); T = w2c[:3, 3]
        fovy = focal2fov(fov2focal(fovx, width), height)
        return Camera(R, T, fovx, fovy, width, height, device=device)

    @staticmethod
    def look_at_origin(theta, phi, radius, fovx, width, height, device="cuda"):
        """Camera on a sphere looking at the origin (nerf_synthetic pose_spherical), OpenGL c2w."""
        eye = radius * np.array([math.cos(phi) * math.sin(theta), -math.cos(phi) * math.cos(theta), math.sin(phi)])
        fwd = -eye / np.linalg.norm(eye)                # camera looks along -z in OpenGL
        up = np.array([0.0, 0.0, 1.0])
        right = np.cross(fwd, up); right /= np.linalg.norm(right)
        upv = np.cross(right, fwd)
        c2w = np.eye(4)
        c2w[:3, 0] = right; c2w[:3, 1] = upv; c2w[:3, 2] = -fwd; c2w[:3, 3] = eye
        return Camera.from_c2w_blender(c2w, fovx, width, height, device=device)
