"""Drop-in for the reference's `diff_gaussian_rasterization` package (the fork with depth output and per-pixel
sampling offsets): same names, argument order, return values and error behaviour as
RAST/diff_gaussian_rasterization/__init__.py:21-238, on top of libwast3d_hip.so instead of the CUDA `_C` module.

  rasterize_gaussians(...)           == __init__.py:21-46
  _RasterizeGaussians                == __init__.py:48-171   (forward returns color, depth, radii)
  GaussianRasterizationSettings      == __init__.py:173-185
  GaussianRasterizer                 == __init__.py:187-238  (markVisible, forward)
  _C.rasterize_gaussians / _backward / mark_visible are the functions of the same name below
  (RAST/ext.cpp:15-19, RAST/rasterize_points.h:18-70).
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import RasterGradIn, RasterGradOut, RasterIn, check, lib, ptr, require_cuda, stream_ptr

NUM_CHANNELS = 3


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.contiguous().float()


def _raster_in(bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, H, W, sh,
               degree, campos, prefiltered, sampling_offsets):
    P = means3D.shape[0]
    M = sh.shape[1] if sh.numel() != 0 else 0
    keep = [_f32c(t) for t in (bg, means3D, sh, colors, opacity, scales, rotations, cov3D_precomp, viewmatrix, projmatrix, campos, sampling_offsets)]
    bg, means3D, sh, colors, opacity, scales, rotations, cov3D_precomp, viewmatrix, projmatrix, campos, sampling_offsets = keep
    rin = RasterIn(P, degree, M, H, W, ptr(bg), ptr(means3D), ptr(sh), ptr(colors), ptr(opacity), ptr(scales), ptr(rotations), ptr(cov3D_precomp),
                   ptr(viewmatrix), ptr(projmatrix), ptr(campos), ptr(sampling_offsets), float(scale_modifier), float(tan_fovx), float(tan_fovy),
                   int(bool(prefiltered)))
    return rin, keep


def rasterize_gaussians_fwd(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx,
                            tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, debug, sampling_offsets):
    """`_C.rasterize_gaussians` (RasterizeGaussiansCUDA, RAST/rasterize_points.cu:35-119): returns
    (num_rendered, color[3,H,W], depth[H,W], radii[P] int32, geomBuffer, binningBuffer, imgBuffer)."""
    if means3D.dim() != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")          # AT_ERROR, rasterize_points.cu:58-60
    require_cuda(means3D, background, opacity, viewmatrix, projmatrix, sampling_offsets)
    dev = means3D.device
    P, H, W = means3D.shape[0], int(image_height), int(image_width)
    out_color = torch.zeros(NUM_CHANNELS, H, W, dtype=torch.float32, device=dev)
    out_depth = torch.zeros(H, W, dtype=torch.float32, device=dev)
    radii = torch.zeros(P, dtype=torch.int32, device=dev)
    geom = torch.empty(0, dtype=torch.uint8, device=dev)
    binning = torch.empty(0, dtype=torch.uint8, device=dev)
    img = torch.empty(0, dtype=torch.uint8, device=dev)
    rendered = 0
    if P != 0:
        if sampling_offsets.shape[0] != H or sampling_offsets.shape[1] != W:
            raise RuntimeError("sampling_offsets must be [H, W, 2]")
        rin, keep = _raster_in(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx,
                               tan_fovy, H, W, sh, degree, campos, prefiltered, sampling_offsets)
        geom = torch.empty(lib().w3d_raster_geom_bytes(P), dtype=torch.uint8, device=dev)
        img = torch.empty(lib().w3d_raster_image_bytes(H, W), dtype=torch.uint8, device=dev)
        holder = {}

        def grow(_user, nbytes):              # the resizeFunctional of rasterize_points.cu:27-33
            holder["t"] = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
            return holder["t"].data_ptr()

        cb = _lib.ALLOC_FN(grow)
        n = C.c_int(0)
        check(lib().w3d_raster_forward(C.byref(rin), ptr(geom), ptr(img), cb, None, ptr(out_color), ptr(out_depth), ptr(radii), C.byref(n), stream_ptr()),
              "w3d_raster_forward")
        rendered = n.value
        binning = holder["t"]
        del keep
    return rendered, out_color, out_depth, radii, geom, binning, img


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx,
                                 tan_fovy, dL_dout_color, dL_dout_depth, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug,
                                 sampling_offsets, opacity=None):
    """`_C.rasterize_gaussians_backward` (RasterizeGaussiansBackwardCUDA, RAST/rasterize_points.cu:121-206): returns
    (dL_dmeans2D[P,3], dL_dcolors[P,3], dL_dopacity[P,1], dL_dmeans3D[P,3], dL_dcov3D[P,6], dL_dsh[P,M,3], dL_dscales[P,3], dL_drotations[P,4])."""
    dev = means3D.device
    P = means3D.shape[0]
    H, W = dL_dout_color.shape[1], dL_dout_color.shape[2]
    M = sh.shape[1] if sh.numel() != 0 else 0
    z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
    g_m3, g_m2, g_col, g_op, g_cov, g_sh, g_sc, g_rot = z(P, 3), z(P, 3), z(P, NUM_CHANNELS), z(P, 1), z(P, 6), z(P, M, 3), z(P, 3), z(P, 4)
    if P != 0:
        accum = z(P, 12)
        op = opacity if opacity is not None else z(P, 1)          # opacities are not read by the backward (they live in geomBuffer)
        rin, keep = _raster_in(background, means3D, colors, op, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy,
                               H, W, sh, degree, campos, False, sampling_offsets)
        gc, gd = _f32c(dL_dout_color), _f32c(dL_dout_depth)
        gin = RasterGradIn(ptr(gc), ptr(gd))
        gout = RasterGradOut(ptr(g_m2), ptr(g_col), ptr(g_op), ptr(g_m3), ptr(g_cov), ptr(g_sh), ptr(g_sc), ptr(g_rot), ptr(accum))
        check(lib().w3d_raster_backward(C.byref(rin), ptr(radii), C.byref(gin), C.byref(gout), ptr(geomBuffer), ptr(binningBuffer), ptr(imageBuffer), int(R),
                                        stream_ptr()), "w3d_raster_backward")
        del keep
    return g_m2, g_col, g_op, g_m3, g_cov, g_sh, g_sc, g_rot


def mark_visible(means3D, viewmatrix, projmatrix):
    require_cuda(means3D, viewmatrix, projmatrix)
    P = means3D.shape[0]
    present = torch.zeros(P, dtype=torch.bool, device=means3D.device)
    if P != 0:
        m, v, p = _f32c(means3D), _f32c(viewmatrix), _f32c(projmatrix)
        check(lib().w3d_mark_visible(ptr(m), ptr(v), ptr(p), P, ptr(present), stream_ptr()), "w3d_mark_visible")
    return present


def _debug_call(fn, args, dump_path: str, where: str):
    """`raster_settings.debug` (pipe.debug / --debug_from; RAST/diff_gaussian_rasterization/__init__.py:90-97, 141-148 and CHECK_CUDA, auxiliary.h:166-173): the
    arguments are copied to the CPU BEFORE the call (so that a fault cannot corrupt them), the device is synchronised right after it (an asynchronous fault surfaces
    here, not at some later call), and on any failure the copies are written to `dump_path` — a file a maintainer can replay — before the exception travels on."""
    snapshot = tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)
    try:
        out = fn(*args)
        torch.cuda.synchronize()
        return out
    except Exception:
        torch.save(snapshot, dump_path)
        print(f"\nAn error occured in {where}. Please forward {dump_path} for debugging.")
        raise


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings, cam_view_depth,
                        sampling_offsets):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings, cam_view_depth,
                                     sampling_offsets)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings, cam_view_depth, sampling_offsets):
        rs = raster_settings
        dev = means3D.device
        sh, colors_precomp, scales, rotations, cov3Ds_precomp = [t.to(dev) if t.device != dev else t for t in (sh, colors_precomp, scales, rotations, cov3Ds_precomp)]
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
                rs.tanfovy, rs.image_height, rs.image_width, sh, rs.sh_degree, rs.campos, rs.prefiltered, rs.debug, sampling_offsets)
        if rs.debug:
            num_rendered, color, depth, radii, geomBuffer, binningBuffer, imgBuffer = _debug_call(rasterize_gaussians_fwd, args, "snapshot_fw.dump", "forward")
        else:
            num_rendered, color, depth, radii, geomBuffer, binningBuffer, imgBuffer = rasterize_gaussians_fwd(*args)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer, imgBuffer, sampling_offsets)
        ctx.mark_non_differentiable(radii)
        return color, depth, radii

    @staticmethod
    def backward(ctx, grad_out_color, grad_out_depth, _):
        rs = ctx.raster_settings
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer, imgBuffer, sampling_offsets = ctx.saved_tensors
        args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy,
                grad_out_color, grad_out_depth, sh, rs.sh_degree, rs.campos, geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer, rs.debug, sampling_offsets)
        if rs.debug:
            g_m2, g_col, g_op, g_m3, g_cov, g_sh, g_sc, g_rot = _debug_call(rasterize_gaussians_backward, args, "snapshot_bw.dump", "backward")
        else:
            g_m2, g_col, g_op, g_m3, g_cov, g_sh, g_sc, g_rot = rasterize_gaussians_backward(*args)
        # same slots as __init__.py:152-164: (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, None, None, None)
        return g_m3, g_m2, g_sh, g_col, g_op, g_sc, g_rot, g_cov, None, None, None


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            return mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, cam_view_depth=None,
                sampling_offsets=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        e = lambda: torch.empty(0, dtype=torch.float32, device=means3D.device)
        shs = e() if shs is None else shs
        colors_precomp = e() if colors_precomp is None else colors_precomp
        scales = e() if scales is None else scales
        rotations = e() if rotations is None else rotations
        cov3D_precomp = e() if cov3D_precomp is None else cov3D_precomp
        if sampling_offsets is None:
            # the reference's render() always passes offsets (gaussian_renderer/__init__.py:31); zeros = pixel centres
            sampling_offsets = torch.zeros(rs.image_height, rs.image_width, 2, dtype=torch.float32, device=means3D.device)
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, rs, cam_view_depth, sampling_offsets)
