"""Scene state of the style loop: the six learnable tensors of the reference's GaussianModel
(scene/gaussian_model.py:44-59), their activations (:33-41,95-115), the Adam parameter groups (:149-167, here one
fused HIP launch), the xyz learning-rate schedule (:169-175, utils/general_utils.py:29-62) and the checkpoint tuple
(:61-93), and the point-cloud initialisation (:124-146, simple-knn's distCUDA2 through csrc/pointcloud.hip).  Densification is outside
the hot path."""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

from . import ops


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Learning-rate schedule of the positions (same call signature and values as utils/general_utils.py:29-62; golden: tests/golden/adam.npz `lr_sched`):
    geometric interpolation lr_init^(1-t) * lr_final^t with t = step / max_steps clamped to [0, 1], multiplied during the first `lr_delay_steps` steps by a
    quarter-sine ramp from `lr_delay_mult` to 1.  Negative steps and an all-zero schedule give 0.
    Exactly ONE rate of 0 (e.g. position_lr_final = 0): the reference's np.log gives -inf on that side, so its schedule is 0 wherever the infinite term has a
    non-zero weight and nan (0 x inf, with a numpy warning) at the one end where the weight is exactly 0.  Here the construction never raises; the schedule is 0
    where the reference's is 0 and the FINITE rate where the reference returns nan."""
    off = lr_init == 0.0 and lr_final == 0.0
    one_zero = (lr_init <= 0.0) != (lr_final <= 0.0)
    log_init, log_span = (0.0, 0.0) if (off or one_zero) else (math.log(lr_init), math.log(lr_final) - math.log(lr_init))

    def schedule(step):
        if off or step < 0:
            return 0.0
        ramp = 1.0
        if lr_delay_steps > 0:
            ramp = lr_delay_mult + (1.0 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0.0), 1.0))
        t = min(max(step / max_steps, 0.0), 1.0)
        if one_zero:                                   # log(0) = -inf on one side: the finite rate where the infinite term has weight exactly 0, else 0
            if lr_final <= 0.0:
                return ramp * lr_init if t == 0.0 else 0.0
            return ramp * lr_final if t == 1.0 else 0.0
        return ramp * math.exp(log_init + t * log_span)
    return schedule


def default_opt():
    """OptimizationParams defaults (arguments/__init__.py:71-89)."""
    return SimpleNamespace(iterations=30000, position_lr_init=1.6e-4, position_lr_final=1.6e-6, position_lr_delay_mult=0.01, position_lr_max_steps=30000,
                           feature_lr=2.5e-3, opacity_lr=0.05, scaling_lr=5e-3, rotation_lr=1e-3, percent_dense=0.1)


class GaussianModel:
    def __init__(self, sh_degree: int):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        e = torch.empty(0)
        self._xyz = self._features_dc = self._features_rest = self._scaling = self._rotation = self._opacity = e
        self.max_radii2D = self.xyz_gradient_accum = self.denom = e
        self.optimizer = None
        self.percent_dense = 0
        self.spatial_lr_scale = 0
        self.scaling_activation = torch.exp
        self.scaling_inverse_activation = torch.log
        self.opacity_activation = torch.sigmoid
        self.inverse_opacity_activation = inverse_sigmoid
        self.rotation_activation = torch.nn.functional.normalize

    # ---- construction from raw arrays (the per-cluster .npz layout of aux_save_clusters_clean.py:161-164) ----
    @classmethod
    def from_tensors(cls, xyz, features_dc, features_rest, scaling, rotation, opacity, sh_degree=3, device="cuda", spatial_lr_scale=1.0):
        m = cls(sh_degree)
        m.active_sh_degree = sh_degree
        mk = lambda t: nn.Parameter(torch.as_tensor(np.asarray(t) if not torch.is_tensor(t) else t, dtype=torch.float32).to(device).contiguous().requires_grad_(True))
        m._xyz, m._features_dc, m._features_rest = mk(xyz), mk(features_dc), mk(features_rest)
        m._scaling, m._rotation, m._opacity = mk(scaling), mk(rotation), mk(opacity)
        m.max_radii2D = torch.zeros(m._xyz.shape[0], device=device)
        m.spatial_lr_scale = spatial_lr_scale
        return m

    @classmethod
    def create_from_pcd(cls, points, colors, spatial_lr_scale: float = 1.0, sh_degree: int = 3, device="cuda"):
        """scene/gaussian_model.py:124-146: xyz = the points, DC colour = RGB2SH(colors), higher SH = 0, log-scale = log sqrt of the mean
        squared distance to the 3 nearest neighbours (clamped at 1e-7), identity rotations, opacity = inverse_sigmoid(0.1)."""
        from .simple_knn import distCUDA2
        xyz = torch.as_tensor(np.asarray(points), dtype=torch.float32).to(device)
        rgb = torch.as_tensor(np.asarray(colors), dtype=torch.float32).to(device)
        P = xyz.shape[0]
        f_dc = ((rgb - 0.5) / 0.28209479177387814).view(P, 1, 3)                      # RGB2SH (utils/sh_utils.py:114-115)
        f_rest = torch.zeros(P, (sh_degree + 1) ** 2 - 1, 3, device=device)
        dist2 = torch.clamp_min(distCUDA2(xyz), 0.0000001)
        scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3)
        rots = torch.zeros(P, 4, device=device); rots[:, 0] = 1
        opac = inverse_sigmoid(0.1 * torch.ones(P, 1, device=device))
        m = cls.from_tensors(xyz, f_dc, f_rest, scales, rots, opac, sh_degree=sh_degree, device=device, spatial_lr_scale=spatial_lr_scale)
        m.active_sh_degree = 0                                                        # the reference starts at degree 0 and raises it during training
        return m

    @classmethod
    def synthetic(cls, P, seed=20, device="cuda", scale=0.005, opacity=0.1, extent=1.3, sh_degree=3):
        """The synthetic scene of SURVEY §8d C3: uniform points in [-extent, extent]^3 (scene/dataset_readers.py:240),
        isotropic scale, identity rotation, opacity 0.1 (scene/gaussian_model.py:136-139), random DC colour, zero rest."""
        g = torch.Generator().manual_seed(seed)
        xyz = torch.rand(P, 3, generator=g) * 2 * extent - extent
        C0 = 0.28209479177387814
        dc = ((torch.rand(P, 1, 3, generator=g)) - 0.5) / C0
        rest = torch.zeros(P, (sh_degree + 1) ** 2 - 1, 3)
        scaling = torch.full((P, 3), math.log(scale))
        rot = torch.zeros(P, 4); rot[:, 0] = 1
        op = inverse_sigmoid(opacity * torch.ones(P, 1))
        return cls.from_tensors(xyz, dc, rest, scaling, rot, op, sh_degree, device)

    # ---- reference accessors ----
    @property
    def get_scaling(self):
        return self.scaling_activation(self._scaling)

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity)

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    def training_setup(self, training_args):
        self.percent_dense = training_args.percent_dense
        dev = self._xyz.device
        self.xyz_gradient_accum = torch.zeros((self._xyz.shape[0], 1), device=dev)
        self.denom = torch.zeros((self._xyz.shape[0], 1), device=dev)
        l = [
            {'params': [self._xyz], 'lr': training_args.position_lr_init * self.spatial_lr_scale, "name": "xyz"},
            {'params': [self._features_dc], 'lr': training_args.feature_lr, "name": "f_dc"},
            {'params': [self._features_rest], 'lr': training_args.feature_lr / 20.0, "name": "f_rest"},
            {'params': [self._opacity], 'lr': training_args.opacity_lr, "name": "opacity"},
            {'params': [self._scaling], 'lr': training_args.scaling_lr, "name": "scaling"},
            {'params': [self._rotation], 'lr': training_args.rotation_lr, "name": "rotation"},
        ]
        self.optimizer = ops.FusedAdam(l, lr=0.0, eps=1e-15)
        self.xyz_scheduler_args = get_expon_lr_func(lr_init=training_args.position_lr_init * self.spatial_lr_scale,
                                                    lr_final=training_args.position_lr_final * self.spatial_lr_scale,
                                                    lr_delay_mult=training_args.position_lr_delay_mult, max_steps=training_args.position_lr_max_steps)

    def update_learning_rate(self, iteration):
        for param_group in self.optimizer.param_groups:
            if param_group["name"] == "xyz":
                lr = self.xyz_scheduler_args(iteration)
                param_group['lr'] = lr
                return lr

    def capture(self):
        return (self.active_sh_degree, self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation, self._opacity, self.max_radii2D,
                self.xyz_gradient_accum, self.denom, self.optimizer.state_dict(), self.spatial_lr_scale)

    def restore(self, model_args, training_args):
        (self.active_sh_degree, self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation, self._opacity, self.max_radii2D,
         xyz_gradient_accum, denom, opt_dict, self.spatial_lr_scale) = model_args
        self.training_setup(training_args)
        self.xyz_gradient_accum = xyz_gradient_accum
        self.denom = denom
        self.optimizer.load_state_dict(opt_dict)
