"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes wrapper of oracle/raster_ref.c (CPU restatement of the
reference rasteriser, RAST/cuda_rasterizer/*.cu).  Only tests/, smoke() and bench.py's cpu_baseline leg
may import this."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libraster_ref.so")
_lib = None
fp = C.POINTER(C.c_float)


class RIn(C.Structure):
    _fields_ = [("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("bg", fp), ("means3D", fp), ("shs", fp), ("colors_precomp", fp), ("opacities", fp), ("scales", fp),
                ("rotations", fp), ("cov3D_precomp", fp), ("view", fp), ("proj", fp), ("campos", fp), ("offsets", fp),
                ("scale_modifier", C.c_float), ("tan_fovx", C.c_float), ("tan_fovy", C.c_float)]


class RGrad(C.Structure):
    _fields_ = [(n, fp) for n in ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drots")]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "raster_ref.c")):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        l = C.CDLL(_SO)
        l.raster_ref_forward.restype = C.c_void_p
        l.raster_ref_forward.argtypes = [C.POINTER(RIn), fp, fp, C.POINTER(C.c_int)]
        l.raster_ref_backward.restype = None
        l.raster_ref_backward.argtypes = [C.POINTER(RIn), C.c_void_p, fp, fp, C.POINTER(RGrad)]
        l.raster_ref_free.argtypes = [C.c_void_p]
        l.raster_ref_num_rendered.restype = C.c_int64
        l.raster_ref_num_rendered.argtypes = [C.c_void_p]
        for n, t in (("radii", C.c_int), ("rgb", C.c_float), ("cov3D", C.c_float), ("means2D", C.c_float), ("conic_opacity", C.c_float),
                     ("depths", C.c_float), ("point_list", C.c_uint32), ("ranges", C.c_uint32), ("n_contrib", C.c_uint32), ("final_T", C.c_float)):
            f = getattr(l, "raster_ref_" + n)
            f.restype = C.POINTER(t)
            f.argtypes = [C.c_void_p]
        l.raster_ref_mark_visible.argtypes = [fp, fp, fp, C.c_int, C.POINTER(C.c_uint8)]
        _lib = l
    return _lib


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(fp)


class Raster:
    """One forward (+ optional backward) of the CPU restatement.  Arguments follow
    RAST/rasterize_points.h:18-43 (same names, numpy arrays, None for absent optionals)."""

    def __init__(self, bg, means3D, opacities, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, H, W, sampling_offsets,
                 shs=None, sh_degree=0, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, scale_modifier=1.0):
        self.keep = [_f(x) for x in (bg, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, viewmatrix, projmatrix, campos,
                                     sampling_offsets)]
        (bg, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, viewmatrix, projmatrix, campos, offs) = self.keep
        self.P = means3D.shape[0]; self.H = H; self.W = W
        self.M = 0 if shs is None else shs.shape[1]
        self.inp = RIn(self.P, sh_degree, self.M, H, W, _p(bg), _p(means3D), _p(shs), _p(colors_precomp), _p(opacities), _p(scales), _p(rotations),
                       _p(cov3D_precomp), _p(viewmatrix), _p(projmatrix), _p(campos), _p(offs), scale_modifier, tan_fovx, tan_fovy)
        self.state = None

    def forward(self):
        color = np.zeros((3, self.H, self.W), np.float32); depth = np.zeros((self.H, self.W), np.float32)
        radii = np.zeros(self.P, np.int32)
        self.state = lib().raster_ref_forward(C.byref(self.inp), _p(color), _p(depth), radii.ctypes.data_as(C.POINTER(C.c_int)))
        self.R = lib().raster_ref_num_rendered(self.state)
        return color, depth, radii

    def geom(self, name, shape, dtype=np.float32):
        ptr = getattr(lib(), "raster_ref_" + name)(self.state)
        return np.ctypeslib.as_array(ptr, shape=shape).astype(dtype, copy=True)

    def backward(self, dL_dcolor, dL_ddepth):
        P, M = self.P, self.M
        out = dict(dL_dmeans2D=np.zeros((P, 3), np.float32), dL_dcolors=np.zeros((P, 3), np.float32), dL_dopacity=np.zeros((P, 1), np.float32),
                   dL_dmeans3D=np.zeros((P, 3), np.float32), dL_dcov3D=np.zeros((P, 6), np.float32), dL_dsh=np.zeros((P, max(M, 1), 3), np.float32),
                   dL_dscales=np.zeros((P, 3), np.float32), dL_drots=np.zeros((P, 4), np.float32))
        g = RGrad(*[_p(out[n]) for n, _ in RGrad._fields_])
        a = _f(dL_dcolor); b = _f(dL_ddepth)
        lib().raster_ref_backward(C.byref(self.inp), self.state, _p(a), _p(b), C.byref(g))
        if M == 0:
            out["dL_dsh"] = np.zeros((P, 0, 3), np.float32)
        return out

    def __del__(self):
        if self.state is not None and _lib is not None:
            _lib.raster_ref_free(self.state)
            self.state = None


def mark_visible(means3D, viewmatrix, projmatrix):
    m = _f(means3D); out = np.zeros(m.shape[0], np.uint8)
    lib().raster_ref_mark_visible(_p(m), _p(_f(viewmatrix)), _p(_f(projmatrix)), m.shape[0], out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out.astype(bool)
