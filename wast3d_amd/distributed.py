"""Gaussian-sharded style loop over the GPUs of one node (BASELINE config C4): one process per GPU,
torch.distributed with the "nccl" backend (= RCCL over xGMI on ROCm).  The reference is single-GPU; this is new.

Sharding:
  * geometry leg — every content row is independent given the style set: rank r owns rows/Gaussians
    [r*P, (r+1)*P) with their Adam state.  The style feature matrix is sharded M/n per rank and all-gathered
    (once: it is constant across iterations; `regather_each_step=True` puts the all-gather inside the step to
    measure it).  No other collective on this leg.
  * render leg — alpha blending is order dependent per pixel, so partial images cannot be summed.  Each rank
    projects its own Gaussians to 48-byte splat records, the records are ALL-GATHERED, each rank bins/sorts/blends a
    contiguous band of tile rows from all records, the (small) image bands are all-gathered so that the image losses
    are evaluated exactly as on one GPU, each rank back-propagates its band into per-splat accumulators for ALL
    Gaussians, and a REDUCE-SCATTER(sum) returns each rank the accumulators of the Gaussians it owns.
    Per iteration and rank: 48 B x P_total in, 48 B x P_total out, + 16 B x H x W.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Tuple

import torch
import torch.distributed as dist

from . import _lib, ops
from . import diff_gaussian_rasterization as dgr
from ._lib import RasterGradOut, check, lib, ptr, stream_ptr

SPLAT = 12   # W3D_SPLAT_FLOATS


# ------------------------------------------------------------------------------------------------ host-side plan (pure python, CPU-testable)
class ShardPlan:
    """Who owns which Gaussians / style rows / tile rows."""

    def __init__(self, world: int, rank: int, gaussians_per_rank: int, style_rows: int, H: int, W: int):
        self.world, self.rank, self.P = world, rank, gaussians_per_rank
        self.P_total = world * gaussians_per_rank
        self.gauss_range = (rank * gaussians_per_rank, (rank + 1) * gaussians_per_rank)
        self.style_ranges = split_even(style_rows, world)
        self.tile_rows = (H + 15) // 16
        self.bands = split_even(self.tile_rows, world)           # contiguous bands of tile rows, sizes differ by at most one
        self.band = self.bands[rank]
        self.H, self.W = H, W

    def band_pixel_rows(self, r=None) -> Tuple[int, int]:
        b = self.bands[self.rank if r is None else r]
        return b[0] * 16, min(b[1] * 16, self.H)


def split_even(n: int, parts: int) -> List[Tuple[int, int]]:
    base, rem = divmod(n, parts)
    out, s = [], 0
    for i in range(parts):
        e = s + base + (1 if i < rem else 0)
        out.append((s, e))
        s = e
    return out


def all_gather_rows(local: torch.Tensor, group=None) -> torch.Tensor:
    """Concatenate equally sized row blocks of every rank (rank order)."""
    world = dist.get_world_size(group)
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out


def all_gather_ragged_rows(local: torch.Tensor, sizes: List[int], group=None) -> torch.Tensor:
    """Row blocks whose sizes differ (image bands, ragged style shards): pad to the largest, gather, strip."""
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    full = all_gather_rows(pad, group).view((len(sizes), mx) + tuple(local.shape[1:]))
    return torch.cat([full[i, : sizes[i]] for i in range(len(sizes))], 0)


def reduce_scatter_rows(full: torch.Tensor, group=None) -> torch.Tensor:
    """Sum `full` over ranks and return this rank's equal row block.  gloo (CPU tests) has no reduce_scatter: all_reduce + slice."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    rows = full.shape[0] // world
    if dist.get_backend(group) == "gloo":
        tmp = full.clone()
        dist.all_reduce(tmp, group=group)
        return tmp[rank * rows:(rank + 1) * rows].clone()
    out = torch.empty((rows,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
    dist.reduce_scatter_tensor(out, full.contiguous(), group=group)
    return out


# ------------------------------------------------------------------------------------------------ staged rasteriser calls
class BandRaster:
    """Stages A-D of include/wast3d_hip.h on one device; the caller moves records / accumulators between ranks."""

    def __init__(self, H: int, W: int, device):
        self.H, self.W, self.dev = H, W, device
        self.img = torch.empty(lib().w3d_raster_image_bytes(H, W), dtype=torch.uint8, device=device)
        self._keep = None

    def preprocess(self, rin_args):
        """rin_args: the tensors/scalars of rasterize_gaussians_fwd for the LOCAL Gaussians. Returns (records view [P,12], radii, geom)."""
        rin, keep = dgr._raster_in(*rin_args)
        P = rin.P
        geom = torch.empty(lib().w3d_raster_geom_bytes(P), dtype=torch.uint8, device=self.dev)
        radii = torch.zeros(P, dtype=torch.int32, device=self.dev)
        check(lib().w3d_raster_preprocess(C.byref(rin), ptr(geom), ptr(radii), stream_ptr()), "w3d_raster_preprocess")
        records = geom[: P * SPLAT * 4].view(torch.float32).view(P, SPLAT)
        self._keep = (rin, keep)
        return records, radii, geom

    def blend_forward(self, records, bg, offsets, band, out_color, out_depth):
        P = records.shape[0]
        scratch = torch.empty(lib().w3d_raster_band_scratch_bytes(P), dtype=torch.uint8, device=self.dev)
        holder = {}

        def grow(_u, nbytes):
            holder["t"] = torch.empty(int(nbytes), dtype=torch.uint8, device=self.dev)
            return holder["t"].data_ptr()

        cb = _lib.ALLOC_FN(grow)
        n = C.c_int(0)
        check(lib().w3d_raster_blend_forward(ptr(records), P, self.H, self.W, band[0], band[1], ptr(bg), ptr(offsets), ptr(scratch), ptr(self.img), cb, None,
                                             ptr(out_color), ptr(out_depth), C.byref(n), stream_ptr()), "w3d_raster_blend_forward")
        return n.value, holder["t"]

    def blend_backward(self, records, bg, offsets, band, binning, R, g_color, g_depth, accum):
        check(lib().w3d_raster_blend_backward(ptr(records), records.shape[0], self.H, self.W, band[0], band[1], ptr(bg), ptr(offsets), ptr(self.img),
                                              ptr(binning), int(R), ptr(g_color), ptr(g_depth), ptr(accum), stream_ptr()), "w3d_raster_blend_backward")

    def preprocess_backward(self, radii, geom, accum_local, M):
        rin, keep = self._keep
        P = rin.P
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.dev)
        g_m3, g_m2, g_col, g_op, g_cov, g_sh, g_sc, g_rot = z(P, 3), z(P, 3), z(P, 3), z(P, 1), z(P, 6), z(P, M, 3), z(P, 3), z(P, 4)
        gout = RasterGradOut(ptr(g_m2), ptr(g_col), ptr(g_op), ptr(g_m3), ptr(g_cov), ptr(g_sh), ptr(g_sc), ptr(g_rot), ptr(accum_local))
        check(lib().w3d_raster_preprocess_backward(C.byref(rin), ptr(radii), ptr(geom), C.byref(gout), stream_ptr()), "w3d_raster_preprocess_backward")
        return g_m2, g_col, g_op, g_m3, g_cov, g_sh, g_sc, g_rot


# ------------------------------------------------------------------------------------------------ the sharded loop
class DistributedStyleLoop:
    """Same iteration as style_loop.GaussianStyleLoop, Gaussians sharded over the ranks of the default process group."""

    def __init__(self, base_loop, plan: ShardPlan, style_shard: torch.Tensor, regather_each_step: bool = False):
        self.b = base_loop                    # a GaussianStyleLoop over the LOCAL Gaussians (owns params, Adam state, buffers)
        self.plan = plan
        self.regather = regather_each_step
        self.style_shard = style_shard.contiguous()
        self.style_sizes = [e - s for s, e in plan.style_ranges]
        self.b.FY = all_gather_ragged_rows(self.style_shard, self.style_sizes)       # cached: the style set is constant
        self.b.M = self.b.FY.shape[0]
        self.b.N_total = plan.P_total
        self.raster = BandRaster(plan.H, plan.W, self.b.pc._xyz.device)
        self.loss = self.b.loss
        self.last_R = 0

    @property
    def timers(self):
        return self.b.timers

    @property
    def knn_events(self):
        return self.b.knn_events

    @knn_events.setter
    def knn_events(self, v):
        self.b.knn_events = v

    def knn_ms(self):
        return self.b.knn_ms()

    def _render_view_sharded(self, xyz, shs, cam, depth_target, wv):
        """One view through the three exchanges (records all-gather, image-band all-gather, accumulator reduce-scatter); returns the raw
        per-Gaussian gradients of the LOCAL Gaussians."""
        b, plan = self.b, self.plan
        pc = b.pc
        dev = xyz.device
        offs = b._offsets()                                       # same seed on every rank -> identical offsets
        e = torch.empty(0, device=dev)
        tx, ty = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
        records, radii, geom = self.raster.preprocess((b.bg, xyz, e, b.act_o, b.act_s, b.act_q, 1.0, e, cam.world_view_transform, cam.full_proj_transform,
                                                       tx, ty, b.H, b.W, shs, pc.active_sh_degree, cam.camera_center, False, offs))
        all_records = all_gather_rows(records)                    # [P_total, 12]  (RCCL all-gather over xGMI)
        color = torch.zeros(3, b.H, b.W, device=dev); depth = torch.zeros(b.H, b.W, device=dev)
        R, binning = self.raster.blend_forward(all_records, b.bg, offs, plan.band, color, depth)
        self.last_R += R
        # image bands -> full image on every rank (16 B per pixel), so the image losses are those of the single-GPU run
        y0, y1 = plan.band_pixel_rows()
        rows = [plan.band_pixel_rows(r)[1] - plan.band_pixel_rows(r)[0] for r in range(plan.world)]
        band_img = torch.cat([color[:, y0:y1], depth[None, y0:y1]], 0).permute(1, 0, 2).contiguous()      # [rows, 4, W]
        full = all_gather_ragged_rows(band_img, rows).permute(1, 0, 2).contiguous()                        # [4, H, W]
        color_full, depth_full = full[:3].contiguous(), full[3].contiguous()
        ops.raw_mse_loss(depth_full, depth_target, b.w_depth * wv, b.loss[1:2], b.g_depth, accumulate=False)
        ops.raw_tv_loss(color_full, b.w_tv * wv, b.loss[2:3], b.g_color, accumulate=False)
        accum = torch.zeros(plan.P_total, SPLAT, device=dev)
        self.raster.blend_backward(all_records, b.bg, offs, plan.band, binning, R, b.g_color, b.g_depth, accum)
        accum_local = reduce_scatter_rows(accum)                  # [P, 12] sums over the ranks' bands
        return self.raster.preprocess_backward(radii, geom, accum_local, shs.shape[1])

    def _render_leg_sharded(self, xyz):
        """Render leg of every view; writes grads[1..5] and b.g_xyz_render like the single-GPU render leg."""
        b = self.b
        pc = b.pc
        N = b.N
        L = lib(); s = stream_ptr()
        check(L.w3d_activations_fwd(ptr(pc._scaling.data), ptr(pc._rotation.data), ptr(pc._opacity.data), N, ptr(b.act_s), ptr(b.act_q), ptr(b.act_o), s))
        shs = torch.cat((pc._features_dc.data, pc._features_rest.data), dim=1)
        self.last_R = 0
        V = len(b.cams)
        tot = None
        for cam, tgt in zip(b.cams, b.depth_targets):
            g = self._render_view_sharded(xyz, shs, cam, tgt, 1.0 / V)
            if tot is None:
                tot = g
            else:
                for a, c in zip(tot, g):
                    a.add_(c)
        g_m2, g_col, g_op, g_m3, g_cov, g_sh, g_sc, g_rot = tot
        check(L.w3d_activations_bwd(ptr(pc._scaling.data), ptr(pc._rotation.data), ptr(pc._opacity.data), N, ptr(g_sc), ptr(g_rot), ptr(g_op),
                                    ptr(b.grads[4]), ptr(b.grads[5]), ptr(b.grads[3]), s))
        b.g_xyz_render.copy_(g_m3)
        b.grads[1].copy_(g_sh[:, :1]); b.grads[2].copy_(g_sh[:, 1:])

    def step(self):
        """Geometry leg (no collective, 70 % of the step) on the main stream; render leg with its three collectives on a second stream,
        so the exchanges over xGMI and the small render kernels overlap the distance-GEMM.  The legs meet in Adam."""
        b = self.b
        xyz = b.pc._xyz.data
        main = torch.cuda.current_stream()
        b.loss.zero_()
        if self.regather:
            b.FY = all_gather_ragged_rows(self.style_shard, self.style_sizes)
        b._geometry_leg(xyz)
        b.side.wait_event(b.ev_feat)
        with torch.cuda.stream(b.side):
            self._render_leg_sharded(xyz)
        main.wait_stream(b.side)
        b.t += 1
        b.lrs[0] = float(b.xyz_sched(b.t))
        groups = [(p.data, g, m, v, lr) for p, g, m, v, lr in zip(b.params, b.grads, b.m, b.v, b.lrs)]
        groups[0] = groups[0] + (b.g_xyz_render,)                 # geometry-leg + render-leg dL/dxyz are summed inside the Adam step
        ops.raw_adam_step(groups, b.t, 0.9, 0.999, 1e-15, zero_grad=True)


def make_synthetic_c4(n_per_rank=1_000_000, M=65536, d=256, k=20, H=800, W=800, timers=False, regather_each_step=False, lists="knn", **kw):
    """Weak-scaling workload: every rank owns `n_per_rank` Gaussians of a world*n_per_rank scene (seeds offset by rank),
    the M style rows are sharded M/world per rank."""
    from .style_loop import make_synthetic_c3
    world, rank = dist.get_world_size(), dist.get_rank()
    plan = ShardPlan(world, rank, n_per_rank, M, H, W)
    s0, s1 = plan.style_ranges[rank]
    base = make_synthetic_c3(N=n_per_rank, M=max(s1 - s0, 1), d=d, k=k, H=H, W=W, seeds=(20 + 100 * rank, 21, 22 + 100 * rank, 23 + 100 * rank), timers=timers, lists=lists, **kw)
    shard = base.FY[: s1 - s0].clone()
    # the depth targets must be the same on every rank: take rank 0's
    for t in base.depth_targets:
        dist.broadcast(t, src=0)
    return DistributedStyleLoop(base, plan, shard, regather_each_step=regather_each_step)
