"""The render-coupled style loop at Gaussian-scene scale (BASELINE config C3; reference skeleton:
notebooks/13.Optimize_guassians_depth.ipynb cell 42, NB13:1463-1475 and train_st_depth_hotdog.py:39-336).

One iteration = a fixed sequence of HIP kernels on one stream (no autograd graph, no host sync except the
rasteriser's tile-instance count):
    featurise      F_X = descriptors(xyz, frozen neighbour lists)            [N, d] bf16        (a3)
    match          k nearest style rows of F_Y per content row, fused MFMA   [N, k]             (a1+a2)
    NN loss head   loss_nn = w_nn * mean_{N*M}(D^2 * mask)  + dL/dF_X                            (a4)
    descriptor bwd dL/dxyz += scatter(dL/dF_X)                                                    (a3)
    activations    exp / normalize / sigmoid                                                      (a10)
    render         one view, colour + depth                                                       (a6)
    image losses   w_depth * MSE(depth, target) + w_tv * tv_loss(render)  + their gradients       (a8)
    render bwd     -> dL/d{xyz, SH, opacity, scale, rotation}                                      (a7)
    Adam           59 floats per Gaussian, six groups, one launch, gradients zeroed in the same pass (a9)
"""
from __future__ import annotations

import ctypes as C
import math
from contextlib import contextmanager
from dataclasses import dataclass, replace
from typing import Optional

import torch

from . import _lib, ops
from . import diff_gaussian_rasterization as dgr
from ._lib import check, lib, ptr, stream_ptr
from .gaussian_model import GaussianModel, default_opt


@dataclass(frozen=True)
class LoopConfig:
    """Every tunable of GaussianStyleLoop in ONE place.  The defaults are the measured best; the loop itself reads no environment variable — A/B scripts and
    `bench.py --loop-config '{...}'` build variants with `LoopConfig(**fields)` (the library reads W3D_KNN_MODE / W3D_KNN_SLACK / W3D_RASTER_BIN once at load, _lib
    W3D_LIB_PATH: diagnostics).  None of these changes a result: they are layout hints, bounds and schedules."""
    # warm start of the fused kNN (exact either way).  "prev": bound = the CURRENT distances of last iteration's k neighbours (k gathered style rows per content row,
    # 0.7 ms at C3; the tightest bound: the prune plan keeps the fewest tiles under it, and the emit sweep + cached lower bounds of round 4 need it).  "motion": last k-th distance + row motion (triangle inequality; touches no style row — the
    # only bound a chunked sweep over a style set that is still arriving can use; it decays over a long run)
    warm: str = "prev"
    prune: bool = True                 # pruned sweeps (16-bit rows of 64 / 128 / 256 features; either warm-start bound gives a valid plan): style tiles an exact ball test rules out are skipped
    adam_under_sweep: bool = True      # schedule "front": the Adam step of the 56 floats only the render leg feeds, in its 32-VGPR form (ops.raw_adam_step(lean=True)), goes to the
                                       # side stream the moment the sweep is launched: a wave of it fits on every SIMD next to the sweep's two 240-VGPR waves, and the sweep leaves HBM
                                       # idle — the step loses the 0.3 ms that Adam cost next to the fused backward, the sweep lasts 0.1 ms longer (round 5: 7.85 -> 7.57 ms per step)
    side_priority: int = -1            # priority of the render leg's stream (-1 = high: its short kernels are dispatched ahead of the main stream's)
    adam_late: bool = True             # schedule "front": the Adam step of the 56 floats only the render leg feeds (a pure HBM stream) runs next to the fused backward's Infinity-Cache
                                       # gathers, after the sweep, instead of at the end of the render leg next to the geometry prologue: 9.39 -> 9.26 ms per step (same device, A/B x2)
    cold_order: bool = True            # round 4: the loop's FIRST sweep (no bound yet) takes every block's tiles nearest-first (the content rows' tree order is built before it
                                       # instead of after): the running k-th best collapses at once instead of in storms (ops.raw_knn_plan_nearest_first)
    emit: bool = True                  # round 4, "prev" bound only: the pruned sweep keeps no k-lists — candidates are emitted to a buffer and selected by a second kernel (warm = 4)
    bound_motion_ratio: float = 0.0    # round 4 (w3d_knn_prepare_hybrid), OFF: rows that moved less than this fraction of their previous k-th distance take the motion bound and
                                       # skip the 20 gathers of the bound kernel.  Measured at 0.02 / 0.05 (profiles/r04g_hybrid_bound.log): the bound kernel alone 0.78 -> 0.32 /
                                       # 0.14 ms, the STEP unchanged (9.96 -> 9.94 / 10.12 ms: the prologue runs next to the render leg, which is the longer of the two), and a
                                       # few content rows far from the style set admit thousands of candidates under a bound 2 % looser — 38-54 regions of the emit sweep
                                       # overflow per sweep and go through the cold repair (late phase: 11.2 -> 19.6 ms per step)
    bounds_summaries: bool = True      # round 4, with bounds_cache: block summaries of the bounds — every plan but the first after a reset drops most tiles on one float per
                                       # (block, tile) and skips the ball test's pass over the centres (ops.KnnBounds)
    bounds_cache: bool = True          # round 4, with emit: cached lower bounds (ops.KnnBounds) — the plan also drops tiles whose rows were all farther away than the
                                       # bound + their motion since the block last swept the tile; reset at every row-order refresh
    multiview_direct: bool = True     # round 4: V > 1 views per iteration run on the one-view path's persistent buffers with an accumulating backward (False: the generic
                                       # per-view wrappers, gradient arrays summed on the host side)
    nn_sum_cache: bool = True          # round 4: the fused backward keeps every row's sum of its k style rows and the list it came from; rows whose neighbour SET is unchanged
                                       # (75-96 % per iteration at C3) read the 1-KB sum instead of gathering k rows (ops.NeighbourSumCache; +1 GB at C3)
    bounds_carry: bool = False         # round 4, OFF: at a row-order refresh the cached bounds follow their rows (KnnBounds.reorder) instead of being reset.  Exact, and the
                                       # refresh iteration's spike (a sweep of everything the ball test keeps: 20 instead of 11.5 ms at iteration 258) is gone — but a refresh
                                       # scatters 40 % of the rows (20 distinct 32-B sectors of the old array per 64 new positions instead of 4), the reorder pass takes 4.8-5.5 ms
                                       # for 2 GB in + 2 GB out, and the all-fresh bounds after a reset are worth a few iterations at 0.18-0.22 instead of 0.25: 97.9 (carry) vs
                                       # 98.2 (reset) it/s over 1001 iterations, refresh every 16: 97.4 vs 97.1 (profiles/r04q_bounds_carry.log).  Costs a second bounds array.
                                       # A sticky bucket key (a row keeps its bucket while any of its k neighbours is in that tile: fewer rows move) made the ORDER worse: 95.6
    cache_budget_bytes: int = 0        # cap on each of the loop's pure performance caches (KnnBounds: 2 GB at C3, 32 GB against M = 1e6 style rows; NeighbourSumCache: 1 GB at
                                       # C3): 0 = at most half of the device memory free at the time of the allocation, > 0 = bytes, < 0 = no caches.  A cache over its budget
                                       # (or whose allocation fails) is left out: the loop runs the plain pruned emit sweep / the gathering backward — slower, same results
    row_deal: bool = True              # round 5: within every block of 512 positions of the row order, the 32 groups of 16 consecutive rows are DEALT to the sweep's
                                       # eight waves round-robin (wave w takes groups w, w + 8, w + 16, w + 24) instead of four neighbouring groups each: the block's
                                       # row set — and so its tile plan — is unchanged, but every wave gets the same mix of candidate-rich and candidate-poor rows, and
                                       # the eight waves of a block meet at every tile barrier (the wait there is 15-20 % of a wave's time: scripts/emit_stamp.py).  Sweep 3.74 -> 3.66 ms, step
                                       # 8.10 -> 8.03 ms (same device, A/B x2, profiles/r05_ab.log)
    row_order: bool = True             # content rows taken by the kNN kernels in a descriptor-tree order (results land in place)
    raster_selfclean: bool = True      # (round 6) the render leg's persistent gradient buffers without a fill per iteration (w3d_raster_backward_selfclean) and the next forward's
                                       # sampling offsets drawn behind the backward: an 80-MB fill and two tiny launches less on the leg's serial chain.  False: rounds 4-6's form
    straggler_rows: int = 16           # (round 6) a kept (block, tile) pair that at most this many of the block's 512 rows still need is not swept by the block: those
                                       # (row, tile) pairs run on the side (knn_straggler_kernel) — exact, same lists (step 6.27 -> 5.78 ms, whole run 135 -> 167 it/s;
                                       # 8 / 24 / 32: 5.82 / 5.82 / 5.89 ms, profiles/r06ay_straggler_ab.log).  0: off
    row_order_refresh: int = 64        # ... refreshed every so many iterations (0: built once; rounds 3-6: 32 — with the straggler pairs a refresh's reset sweep weighs more: whole run 167 -> 174 it/s at 64,
                                       # 48 / 96 / 192 / never: 173 / 175 / 168 / 140; profiles/r06bj, r06bk).  The order AGES: 300 iterations into the C3 loop a pruned sweep under
                                       # the order of iteration 1 takes 18.8 ms, under a fresh one 13.5-14.1 ms (profiles/r03b_order.log)
    row_order_kind: str = "tree"       # (round 6: "tree" — with the dense builder the tree itself is the cheapest good refresh; rounds 3-5: "tile")  what a refresh builds (the FIRST order is always the tree): "tree" = the tree again (126 ms at C3); "nn0" = rows sorted by the
                                       # position of their nearest style row (0.25 ms; the style rows are stored in tree order); "tile" = the current order stably
                                       # re-bucketed by the 64-row style TILE of the nearest style row (rows that stayed in their tile keep the tree's order inside it)
    row_order_proj: int = 3            # principal components the content rows' tree is built on (0: all d).  Round 6: 3 instead of 16 — on fewer components the blocks come out
                                       # compact in the directions that carry the distance: swept fraction 0.146 -> 0.135 in the headline window (profiles/r06x_tree_headline_ab.log)
    row_tree_leaf: int = 256           # the content rows' tree is bisected down to leaves of at most this many rows (the plan's unit is the block of 512)
    row_tree_iters: int = 2            # (row_tree_fast = False only) power iterations per node and level
    row_tree_fast: bool = True         # round 6: content_tree_order — the tree built densely (~10 ms at C3 instead of 126): cheap enough to BE the refresh
    style_order: bool = True           # style rows stored in the order of a principal-axis tree over them (what makes a kernel tile a compact ball)
    style_order_proj: int = 3          # principal components the STYLE rows' tree (whose leaves are the kernel's 64-row tiles) is built on (0: all d).  Round 6: 3 instead of all —
                                       # with tiles and content blocks both cut in the SAME few directions a block's rows need far fewer tiles: swept fraction 0.134 -> 0.106,
                                       # sweep 3.55 -> 2.93 ms (profiles/r06ab_style_order_ab.log); other pairings lose (content 2 / 4 with style 3: 0.129 / 0.120)
                                       # (the dense builder for the style rows, and the content tree on the style rows' own principal basis: no further gain, r06ad)
    slot_rows: int = -1                # rows per block of the slot plan (featurisation + fused backward): -1 = by row count (1024 at C3), 0 = off (index-window kernels)
    featurise_window: bool = True      # False: the L2-gather featurisation kernel (A/B)
    schedule: str = "front"            # one view: "front" (default since round 3) = the WHOLE render leg (forward, backward, Adam of its five groups) from t = 0 next to featurise /
                                       # bound / plan, the sweep starts when it is through and has the chip to itself, and so has the fused backward; "early" (round 2's default) =
                                       # render forward from t = 0, render backward gated on the GEMM's end (next to the fused backward): 14.33 vs 14.23 ms per step, same device;
                                       # "overlap" (multi-view and sharded loops) = the render leg after the featurisation


class GaussianStyleLoop:
    def __init__(self, pc: GaussianModel, desc_idx: torch.Tensor, style_feats: torch.Tensor, camera, depth_target: torch.Tensor, bg: torch.Tensor,
                 k: int = 20, w_nn: float = 1.0, w_depth: float = 1.0, w_tv: float = 1e-2, opt=None, sampling_seed: int = 21,
                 deterministic_scatter: Optional[bool] = None, timers: bool = False, gemm: str = "same", warm: str = "auto",
                 feat_center: Optional[torch.Tensor] = None, row0: int = 0, n_points_total: Optional[int] = None, exchange=None, order_style: bool = True,
                 config: Optional[LoopConfig] = None):
        self.pc = pc
        cfg = config or LoopConfig()
        if warm != "auto":                      # the constructor argument is the short form of config.warm
            cfg = replace(cfg, warm=warm)
        if cfg.warm not in ("prev", "motion") or cfg.schedule not in ("early", "overlap", "front") or cfg.row_order_kind not in ("tree", "nn0", "tile"):
            raise ValueError(f"unknown loop configuration: {cfg}")
        self.cfg = cfg
        dev = pc._xyz.device
        # Sharded scenes (wast3d_amd.distributed): this loop owns rows [row0, row0 + N) of a scene of n_points_total Gaussians stored along ONE curve; the
        # descriptor lists hold GLOBAL indices (neighbours across the shard boundary belong to other ranks).  `exchange` supplies the two collectives
        # of the geometry leg: positions of all Gaussians in (all-gather), dL/dxyz of foreign Gaussians out (reduce-scatter).
        self.row0 = int(row0)
        self.P_total = int(n_points_total) if n_points_total is not None else pc._xyz.shape[0]
        self.exchange = exchange
        # processing order of the content rows in the kNN kernels (a layout hint, results unchanged): rows with similar features in one MFMA block share
        # the style tiles their candidates sit in.  Built from the feature rows after the first iteration, rebuilt every cfg.row_order_refresh iterations
        # (the rows drift slowly)
        self.row_order = None
        self._row_seq = None
        self._sweeps = 0
        self.prune = None
        self.bounds = None                        # ops.KnnBounds of the pruned emit sweeps (built at the first pruned sweep, reset with the row order)
        self.style_stream = None                  # distributed.StyleStream: the style rows of other ranks arrive during the sweep (BASELINE config C4, in-loop all-gather)
        # Round 6 — the in-loop all-gather on the PRUNED emit path (distributed.DistributedStyleLoop(style="inloop")): the other ranks' shards are re-delivered every iteration
        # into the padded layout of self.FY while featurisation and the render leg run; the bound, the plan and ONE pruned emit sweep follow once they are all there.
        #   style_token     identity of the style rows' CONTENT when the tensor is rewritten every iteration with the same rows (the reference's style set is constant,
        #                   NB11:1057-1063): stands in for (storage, in-place version, shape) in every per-style-set cache key; whoever changes the rows bumps it
        #   style_wait      callable run on the main stream right before the step's first use of self.FY (waits for the shards, refreshes their norms)
        #   style_external_norms  the -0.5 |y|^2 array of the kNN workspace is maintained by style_wait (w3d_knn_chunk_norms: padding rows = -inf): the prepare calls keep it
        #   style_geometry_index  [M] long: the row whose values stand for each stored row in the prune plan's tile geometry (padding rows -> a real row of their shard)
        self.style_token = None
        self.style_wait = None
        self.style_external_norms = False
        self.style_geometry_index = None
        # feat_center [d] f32: the shift the style rows were stored with (normally their mean row); the content rows get the same shift before
        # they are rounded to 16 bits.  Distances are unchanged, the rounding error of the match drops ~10x (tests/test_gpu_parity_fp32.py)
        self.center = None if feat_center is None else feat_center.to(dev).float().contiguous()
        self.kws = ops.KnnWorkspace()             # this loop's own kNN workspace (bounds, flags): not shared with other loops on the device
        self._sum_cache = None; self._sum_cache_of = None
        self._style_seen = None                   # key of the style rows the last iteration matched against
        self._bounds_refused = False; self._sum_cache_refused = False      # a performance cache that did not fit its budget (LoopConfig.cache_budget_bytes) is not retried
        self._ynorm_of = None                     # key of the style matrix whose norms the workspace holds (_style_key)
        self.N = pc._xyz.shape[0]
        self.N_total = self.P_total           # rows the NN loss averages over (all ranks' rows when the scene is sharded)
        self.idx = ops.as_i32(desc_idx).contiguous()
        assert self.idx.shape[0] == self.N and self.row0 + self.N <= self.P_total
        self.d = self.idx.shape[1] - 1
        self.FY = style_feats.contiguous()
        assert self.FY.dtype in (torch.bfloat16, torch.float16) and self.FY.shape[1] == self.d
        # the style rows are stored in the order of a principal-axis tree over them (descriptor_tree_order): nn_idx numbers rows in THAT order,
        # style_perm[j] is the caller's row number of stored row j
        self.style_perm = None
        if order_style and cfg.style_order and self.FY.shape[0] > 128:
            self.style_perm = descriptor_tree_order(self.FY, proj_dim=cfg.style_order_proj)
            self.FY = self.FY[self.style_perm].contiguous()
        # gemm = "same": the distance-GEMM runs on the feature dtype (bf16 / f16);  "fp8": on e4m3 rows with a power-of-two scale per
        # row (BASELINE config C5) — the matches are the exact kNN of the quantised rows, the gradient uses the 16-bit features
        assert gemm in ("same", "fp8")
        self.gemm = gemm
        self._warm_mode = cfg.warm                # what the last prepare used ("motion" under a chunked sweep whatever the configuration)
        self.M = self.FY.shape[0]
        self.M_true = self.M                      # style rows that exist (a sharded style set is stored with padding rows between the shards)
        self.k, self.w_nn, self.w_depth, self.w_tv = k, w_nn, w_depth, w_tv
        # one view per iteration (the reference's loops) or V fixed views per iteration (BASELINE config C5: 4): a camera / a list of cameras,
        # depth targets [H,W] / [V,H,W]; the image losses are averaged over the views
        self.cams = list(camera) if isinstance(camera, (list, tuple)) else [camera]
        self.cam, self.bg = self.cams[0], bg.to(dev).float()
        self.H, self.W = self.cam.image_height, self.cam.image_width
        dt = depth_target.to(dev).float().contiguous()
        self.depth_targets = [dt] * len(self.cams) if dt.dim() == 2 else [dt[v].contiguous() for v in range(len(self.cams))]
        self.opt = opt or default_opt()
        self.gen = torch.Generator(device=dev).manual_seed(sampling_seed)
        self.params = [pc._xyz, pc._features_dc, pc._features_rest, pc._opacity, pc._scaling, pc._rotation]
        self.names = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
        self.lrs = [self.opt.position_lr_init * max(pc.spatial_lr_scale, 1e-12), self.opt.feature_lr, self.opt.feature_lr / 20.0, self.opt.opacity_lr,
                    self.opt.scaling_lr, self.opt.rotation_lr]
        from .gaussian_model import get_expon_lr_func
        self.xyz_sched = get_expon_lr_func(self.lrs[0], self.opt.position_lr_final * max(pc.spatial_lr_scale, 1e-12),
                                           lr_delay_mult=self.opt.position_lr_delay_mult, max_steps=self.opt.position_lr_max_steps)
        z = lambda p: torch.zeros_like(p.data)
        self.grads = [z(p) for p in self.params]
        self.m = [z(p) for p in self.params]
        self.v = [z(p) for p in self.params]
        N, d = self.N, self.d
        self.FX = torch.empty(N, d, dtype=self.FY.dtype, device=dev)
        if gemm == "fp8":
            self.qX = torch.zeros(N, d, dtype=torch.uint8, device=dev); self.sX = torch.full((N,), 127, dtype=torch.int32, device=dev)
            self.motion2_q = torch.zeros(N, dtype=torch.float32, device=dev)
            self._qY_of = None
        self.xyz4 = torch.empty(self.P_total, 4, dtype=torch.float32, device=dev)      # positions of ALL Gaussians of the scene (own rows packed locally, the rest gathered)
        # own rows: packed straight into the scene buffer when the scene is whole; into a separate send buffer when it is sharded (the all-gather then
        # writes every rank's block, this rank's included — no aliasing of send and receive buffers)
        self.xyz4_own = self.xyz4[self.row0:self.row0 + N] if self.P_total == N else torch.empty(N, 4, dtype=torch.float32, device=dev)
        # dL/dxyz of the geometry leg: straight into the Adam gradient when the scene is whole, else into a scene-sized buffer that is reduce-scattered
        self.gx_scene = None if self.P_total == N else torch.zeros(self.P_total, 3, dtype=torch.float32, device=dev)
        self.g_xyz_render = torch.zeros(N, 3, dtype=torch.float32, device=dev)
        # high priority: the render leg's short kernels take the CUs the distance-GEMM frees at its block boundaries first, so
        # the whole leg finishes inside the GEMM's shadow instead of queueing behind it
        self.side = torch.cuda.Stream(device=dev, priority=cfg.side_priority)
        self.schedule = cfg.schedule              # see step()
        self.cu_reserve = 0                       # CUs a pruned sweep leaves free for the other stream (DistributedStyleLoop sets it for short sweeps)
        self.ev_knn = torch.cuda.Event(); self.ev_rfwd = torch.cuda.Event(); self.ev_pre_sweep = torch.cuda.Event()
        self.ev_feat = torch.cuda.Event()          # recorded once the feature rows are written: the render leg starts after the (memory-bound)
                                                   # featurisation instead of competing with it, and then runs in the distance-GEMM's shadow
        # deterministic_scatter = True: NN-loss backward -> [N,d] gradient -> atomic-free CSR gather (bit-reproducible);
        # False: one fused row-order pass with an LDS accumulation window (dL/dF_X never goes through HBM);
        # None (default): the window pass if the lists are local enough for it (entries within +-4096 rows of their own row: the window's
        # reach; everything else is a global float atomic there, 771 M of them for random lists), else the CSR gather
        if deterministic_scatter is None:
            rows = torch.arange(self.row0, self.row0 + self.N, device=dev, dtype=torch.int32)[:, None]
            sample = slice(0, self.N, max(1, self.N // 4096))
            local = float(((self.idx[sample].int() - rows[sample]).abs() <= 4096).float().mean())
            deterministic_scatter = local < 0.6
        self.gFX = torch.empty(N, d, dtype=torch.float32, device=dev) if deterministic_scatter else None
        self.nn_idx = torch.empty(N, k, dtype=torch.int32, device=dev)
        self.nn_prev = torch.empty(N, k, dtype=torch.int32, device=dev)     # last iteration's neighbours seed the selection bound
        self.have_prev = False
        self.motion2 = torch.zeros(N, dtype=torch.float32, device=dev)
        # |x|^2 of the rounded feature rows, written by the featurisation pass STRAIGHT into the kNN workspace's own slot (16-bit rows with a resident style set): the kNN
        # calls then neither recompute nor copy it (the copy alone was 0.1 ms per step next to the render leg)
        self.bind_xnorm()
        self.nn_d2 = torch.empty(N, k, dtype=torch.float32, device=dev)
        self.act_s = torch.empty(N, 3, device=dev); self.act_q = torch.empty(N, 4, device=dev); self.act_o = torch.empty(N, 1, device=dev)
        self.g_color = torch.empty(3, self.H, self.W, device=dev); self.g_depth = torch.empty(self.H, self.W, device=dev)
        self.loss = torch.zeros(3, device=dev)        # nn, depth, tv  (weighted)
        self._swept_on_side = False; self._swept_pending = False; self.strag_log = []
        self._next_offs = None
        self.csr = ops.DescriptorCSR(self.idx, self.P_total) if deterministic_scatter else None
        self.max_fanin = ops.list_max_fanin(self.idx, self.P_total) if not deterministic_scatter else 0       # frozen lists: once
        # Slot plan: the window of the featurisation / fused backward is the SET of points each row block names instead of an index range — no entry leaves LDS (exact
        # lists of a Z-ordered scene: 16 % of the entries fall outside an index window), no index rows read.  Used when every block's set fits (outside fraction 0:
        # any spatially local list); otherwise the index-window kernels (out-of-window entries = global float atomics) or, for lists without locality, the CSR gather.
        self.slots = None
        slot_rows = cfg.slot_rows
        if slot_rows < 0:
            # 1024: sets of <= 6.8 k points — accumulators AND positions of the set fit LDS; a rank's share of a sharded scene (125 k rows at 8 GPUs)
            # would be 123 blocks of 1 024 rows on 256 CUs: at least ~2 blocks per CU instead
            slot_rows = int(min(1024, max(128, -(-self.idx.shape[0] // 512))))
        if not deterministic_scatter and slot_rows > 0 and self.d % 8 == 0 and self.d <= 256:
            sp = ops.SlotPlan(self.idx, self.P_total, rows_per_block=slot_rows)
            if sp.outside_fraction == 0.0:
                self.slots = sp
        self._fy_absmax_of = None
        self.t = 0
        self.last_R = 0
        self.timers = timers
        self.knn_events = []
        self.knn_kernel_events = []
        self.swept_log = []                       # timers: per sweep, the number of (row block, style tile) pairs its plan kept — a device scalar, read by bench.py after the run
        self.stage_events = {}                    # timers: name -> [(start, end)] HIP events on the stream the stage runs on (bench.py's per-stage rooflines)

    # the reference draws fresh offsets every render (gaussian_renderer/__init__.py:31); here they come from a seeded
    # device generator so that a run is reproducible
    @property
    def depth_target(self):
        return self.depth_targets[0]

    @depth_target.setter
    def depth_target(self, t):
        t = t.float().contiguous()
        self.depth_targets = [t] * len(self.cams) if t.dim() == 2 else [t[v].contiguous() for v in range(len(self.cams))]

    def _offsets(self):
        return -torch.rand(self.H, self.W, 2, device=self.pc._xyz.device, generator=self.gen)

    def _fy_absmax(self):
        # max |F_Y| (bounds the fixed-point scale of the fused backward); F_Y is constant, re-evaluated only if the tensor is replaced or rewritten in place
        # (keyed like every other per-style-set cache — storage, in-place version, shape: loop.FY.copy_(larger rows) used to leave a stale, too small
        # range here and the fixed-point accumulators of the fused backward could saturate silently)
        key = self._fy_key()
        if self._fy_absmax_of != key:
            self._fy_absmax_val = ops.absmax_16(self.FY); self._fy_absmax_of = key
        return self._fy_absmax_val

    @contextmanager
    def _stage(self, name):
        """With timers on: HIP events (on the current stream) around one stage of the iteration, in the loop as it runs — i.e. with whatever the other
        stream does next to it."""
        if not self.timers:
            yield
            return
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        yield
        e1.record()
        self.stage_events.setdefault(name, []).append((e0, e1))

    def stage_ms(self):
        """name -> average duration (ms) of the recorded stages; clears the records."""
        torch.cuda.synchronize()
        out = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in self.stage_events.items() if v}
        self.stage_events = {}
        return out

    def stream_bytes_per_row(self):
        """Algorithmic HBM bytes per content row of (featurisation, fused backward) in this loop's configuration — what bench.py prices the two stages with:
        index row 4K B or, under a slot plan that covers every entry, 2 B slots per entry; feature row 2d B written (+ 2d B re-read when a bound needs the
        row motion); backward: the same list bytes + own row 2d + neighbour list and distances 8k + position 16 + gradient 12."""
        K, d, k = self.idx.shape[1], self.d, self.k
        lst = 2 * d + 4 if (self.slots is not None and self.slots.root is not None) else 4 * K
        need_motion = self.cfg.warm == "motion"
        fb = self.FX.element_size()
        return lst + fb * d * (2 if need_motion else 1) + 16 + 8, lst + fb * d + 8 * k + 16 + 12

    def _fy_key(self):
        """Identity of the style rows' content: the tensor's (storage, in-place version, shape), or the owner's token when the rows are re-delivered every iteration."""
        return self.style_token if self.style_token is not None else _tensor_key(self.FY)

    def _content_tree(self):
        """The content rows' tree order from the current feature rows (LoopConfig.row_tree_fast: the dense builder of round 6, else rounds 3-5's)."""
        c = self.cfg
        if c.row_tree_fast:
            # the principal basis of the first build is kept for 8 builds (the rows' low-dimensional view changes slowly; the basis is a layout hint of a layout hint)
            n = getattr(self, "_tree_builds", 0)
            basis = getattr(self, "_tree_basis", None) if n % 8 else None
            order, self._tree_basis = content_tree_order(self.FX, leaf=c.row_tree_leaf, proj_dim=c.row_order_proj, basis=basis, return_basis=True)
            self._tree_builds = n + 1
            return order
        return descriptor_tree_order(self.FX, leaf=c.row_tree_leaf, iters=c.row_tree_iters, proj_dim=c.row_order_proj)

    def _await_style(self):
        if self.style_wait is not None:
            self.style_wait()

    def _make_prune(self):
        """Tile geometry of the current style rows (padding rows of a sharded layout stand in as copies of a real row of their shard: a valid, tight ball)."""
        rows = self.FY if self.style_geometry_index is None else self.FY[self.style_geometry_index].contiguous()
        pr = ops.KnnPrune(rows)
        pr.of, pr.key = self.FY, self._fy_key()
        return pr

    def _style_key(self):
        """Identity of (style rows, workspace) for "their norms are still in the workspace": the tensor, its in-place modification counter (loop.FY.copy_(...) bumps it)
        and the workspace buffer itself (a regrow zero-fills the norms)."""
        return (self._fy_key(), None if self.kws.t is None else self.kws.t.data_ptr())

    def _alloc_cache(self, nbytes: int, make):
        """A pure performance cache (cached lower bounds, neighbour sums) or None: None when it exceeds LoopConfig.cache_budget_bytes or its allocation fails — the loop
        then runs without it (slower, same results)."""
        def refused(why):
            if not getattr(self, "_cache_warned", False):            # once per loop: a refused cache is a permanent slowdown the user should hear about
                import warnings
                warnings.warn(f"wast3d_amd: a {nbytes / 2 ** 30:.2f}-GiB performance cache of the style loop was left out ({why}); the loop runs without it — same results, slower "
                              f"(LoopConfig.cache_budget_bytes)", RuntimeWarning, stacklevel=3)
                self._cache_warned = True
            return None
        if not ops.cache_budget_ok(nbytes, self.cfg.cache_budget_bytes, self.FX.device):
            return refused("over its budget")
        try:
            return make()
        except torch.OutOfMemoryError:
            torch.cuda.empty_cache()
            return refused("the allocation failed")

    def bind_xnorm(self):
        """(Re)creates self.xnorm for the current style set: the workspace's own slot where the kNN calls can take it in place (16-bit rows, resident style set of a
        supported width), else a buffer of its own.  Call again after replacing self.FY (DistributedStyleLoop does)."""
        if self.style_stream is None and self.gemm != "fp8" and self.d in (64, 128, 256) and feat_dtype_ok(self.FY):
            self.xnorm = self.kws.xnorm_view(self.N, self.FY.shape[0], self.d, self.k, ops._dtype_code(self.FY), self.pc._xyz.device)
        else:
            self.xnorm = torch.empty(self.N, dtype=torch.float32, device=self.pc._xyz.device)

    def _featurise(self, motion2=None, xnorm=None):
        """F_X = descriptors(xyz, frozen lists) as 16-bit rows.  Curve-local lists: positions of each block's window staged in LDS (the neighbour
        fetches are LDS reads); otherwise the L2-gather kernel."""
        with self._stage("featurise"):
            if self.csr is None and self.cfg.featurise_window:
                ops.raw_descriptor_fwd_window(self.xyz4, self.idx, self.FX, motion2, xnorm, center=self.center, slots=self.slots)
            elif motion2 is not None:
                ops.raw_descriptor_fwd_x4_motion(self.xyz4, self.idx, self.FX, motion2, xnorm, center=self.center)
            else:
                ops.raw_descriptor_fwd_x4(self.xyz4, self.idx, self.FX, center=self.center)

    def _geometry_prepare(self, xyz):
        """Geometry leg up to (not including) the distance-GEMM: featurise (+ row motion), quantise (fp8), row norms and warm-start bound."""
        k = self.k
        ops.raw_pack_xyz4(xyz, self.xyz4_own)                   # one 16-B load per gather in the three gather passes below
        if self.exchange is not None and self.gx_scene is not None:
            self.exchange.gather_xyz4(self.xyz4, self.xyz4_own)  # positions of the other ranks' Gaussians (16 B each; the lists reach across shard boundaries)
        if self.style_stream is None:
            # the style rows changed since the last iteration — another tensor, or the same one rewritten in place: last iteration's neighbours name rows that are no
            # longer there (any k rows give a VALID bound, but a bound from k random rows is so loose that the emit sweep's candidate regions overflow): one cold sweep
            key = self._fy_key()
            if self._style_seen is not None and self._style_seen != key:
                self.have_prev = False
                self._bounds_refused = self._sum_cache_refused = False      # another style set (other size, other moment): a cache that did not fit then may fit now
            self._style_seen = key
        warm_src = self.nn_d2 if self.have_prev else None
        if self.gemm == "fp8":
            if self.style_external_norms:
                raise _lib.W3dError("fp8 distance-GEMM over a padded, re-delivered style layout is not supported: use the chunked in-loop sweep or the cached gather")
            if self._qY_of != self._fy_key():                   # the style set is constant: quantised once
                self.qY, self.sY = ops.quantize_rows_fp8(self.FY); self._qY_of = self._fy_key()
            self._featurise()
            self.ev_feat.record()
            # quantise in place; motion of the DEQUANTISED rows feeds the exact triangle-inequality warm start
            ops.quantize_rows_fp8(self.FX, self.qX, self.sX, motion2=self.motion2_q if self.have_prev else None)
            self._warm = int(self.have_prev)
            self._warm_mode = self.cfg.warm
            # fp8 rows take the gathered bound too.  In a loop's first ~20 iterations the motion bound gives the faster sweep (19.9 vs 22.6 ms at C3: on quantised rows
            # ~500 rows per sweep tie at the k-th place under the gathered bound and are redone cold) — but it decays with the run: 44 ms per sweep after 60
            # iterations, 52 ms after 400, against a steady 22.4-23.4 ms (scripts/late_warm_ab.py with GEMM=fp8)
            if self.have_prev and self._warm_mode == "prev":
                ops.raw_knn_fp8_phase(self.qX, self.sX, self.qY, self.sY, k, 1, 1, prev_idx=self.nn_idx, ws=self.kws, row_order=self.row_order)
                self._warm = 1
                if self.cfg.prune and self.d in (64, 128, 256):
                    # pruned sweep of the QUANTISED problem: the plan works on the dequantised rows (q x 2^e: exact in bf16), the bounds are the fp8 prepare's
                    # Round 5: with LoopConfig.emit the sweep is the emit kernel's fp8 form (warm = 4) and, with bounds_cache, records / uses the cached lower bounds
                    # exactly as the 16-bit loop does (motion of the DEQUANTISED rows from the quantisation pass); steady plans do not read the rows at all
                    use_bounds = self.cfg.emit and self.cfg.bounds_cache and not self._bounds_refused
                    with self._stage("prune"):
                        if self.prune is None or self.prune.key != _tensor_key(self.qY):
                            self.prune = ops.KnnPrune(_dequant_bf16(self.qY, self.sY), fp8=True); self.prune.of = self.qY; self.prune.key = _tensor_key(self.qY)
                            self.bounds = None
                        if use_bounds:
                            if self.bounds is None or not self.bounds.fits(self.N, self.M):
                                self.bounds = self._alloc_cache(ops.KnnBounds.bytes_needed(self.N, self.M, self.cfg.bounds_summaries, False),
                                                                lambda: ops.KnnBounds(self.N, self.M, self.FX.device, summaries=self.cfg.bounds_summaries))
                                self._bounds_refused = self.bounds is None
                            else:
                                self.bounds.advance(self.motion2_q)
                        else:
                            self.bounds = None
                        Xd = None if ops.knn_plan_is_steady(self.bounds) else _dequant_bf16(self.qX, self.sX)
                        ops.raw_knn_prune_plan(Xd, k, self.prune, self.kws, row_order=self.row_order, bounds=self.bounds, shape=(self.N, self.d, ops.BF16))
                        self._warm = 4 if self.cfg.emit else 3
                    if self.timers:
                        self.swept_log.append(ops.knn_prune_counts(self.N, self.M, self.d, k, self.kws).sum())
            else:
                ops.raw_knn_fp8_phase(self.qX, self.sX, self.qY, self.sY, k, 1, self._warm, prev_dist2=warm_src, motion2=self.motion2_q if self.have_prev else None, ws=self.kws)
        elif self.style_stream is not None:
            # the style set arrives shard by shard during the sweep: only bounds that need no style row (cold, or k-th distance + motion)
            if self.have_prev:
                self._featurise(self.motion2, self.xnorm)
                self.ev_feat.record()
                self._warm = ops.raw_knn_prepare_chunked(self.FX, self.M, k, prev_dist2=self.nn_d2, motion2=self.motion2, xnorm=self.xnorm, ws=self.kws)
            else:
                self._featurise()
                self.ev_feat.record()
                self._warm = ops.raw_knn_prepare_chunked(self.FX, self.M, k, ws=self.kws)
            self._warm_mode = "motion"
        elif self.have_prev:
            # feature rows updated in place + |x|^2 of the rounded rows in the same pass; how far each row moved only for the motion bound (the default, the
            # gathered bound, does not need it: the old rows are then not re-read)
            self._warm_mode = self.cfg.warm
            # (measured and dropped: emit + cached bounds under the MOTION bound — no gather of the previous neighbours' rows — admit (1 + |dx| / D)^16 times the candidates; in a
            # run's first iterations |dx| ~ D, 820 000 of 1M rows overflowed their regions of the candidate buffer: profiles/r04e_motion_emit.log)
            tight = self._warm_mode == "prev"
            use_bounds = tight and self.cfg.prune and self.cfg.emit and self.cfg.bounds_cache and feat_dtype_ok(self.FY) and self.gemm == "same"
            # (the cached lower bounds need every row's motion since the previous sweep: the featurisation pass reports it while it overwrites the rows)
            self._featurise(self.motion2 if (self._warm_mode == "motion" or use_bounds) else None, self.xnorm)
            self.ev_feat.record()
            self._await_style()                                  # (in-loop gather: the shards arrived while the rows were featurised; their norms are refreshed)
            with self._stage("bound"):
                if self._warm_mode == "prev":
                    # bound = the previous neighbours' CURRENT distances (exact result; the prepare phase has read nn_idx before the kernel rewrites it); |x|^2 comes
                    # from the featurisation pass (the norm pass over the 512 MB of rows it saves was 0.14 ms per step)
                    key = self._style_key()
                    hyb = use_bounds and self.cfg.bound_motion_ratio > 0.0
                    self._warm = ops.raw_knn_prepare(self.FX, self.FY, k, prev_idx=self.nn_idx, ws=self.kws, row_order=self.row_order, xnorm=self.xnorm,
                                                     style_unchanged=self.style_external_norms or self._ynorm_of == key, prev_dist2=self.nn_d2 if hyb else None, motion2=self.motion2 if hyb else None,
                                                     motion_ratio=self.cfg.bound_motion_ratio if hyb else 0.0)
                    self._ynorm_of = self._style_key()       # (the style set of a loop is constant: its norms stay in the workspace; taken AFTER the call: it may have grown the workspace)
                else:
                    # bound = last iteration's k-th distances + the motion (triangle inequality) — exact result, no gather of style rows
                    # (external norms: no style matrix is passed, so the prepare leaves the norm array alone — the chunked sweeps' convention)
                    self._warm = ops.raw_knn_prepare(self.FX, None if self.style_external_norms else self.FY, k, prev_dist2=self.nn_d2, motion2=self.motion2, xnorm=self.xnorm, ws=self.kws,
                                                     row_order=self.row_order, M=self.FY.shape[0])
            if self.cfg.prune and feat_dtype_ok(self.FY):
                # pruned sweep (exact): style tiles whose ball lies beyond every bound of a row block are not swept (w3d_knn_prune_plan)
                with self._stage("prune"):
                    if self.prune is None or self.prune.key != self._fy_key():
                        # a new style set — another tensor, or the same one rewritten in place (loop.FY.copy_(...) bumps its version counter): the tiles' centres
                        # and radii are re-measured, and the cached bounds of the old rows mean nothing (both used to be keyed on tensor identity alone)
                        self.prune = self._make_prune()
                        self.bounds = None
                    if use_bounds and not self._bounds_refused:
                        if self.bounds is None or not self.bounds.fits(self.N, self.M):
                            self.bounds = self._alloc_cache(ops.KnnBounds.bytes_needed(self.N, self.M, self.cfg.bounds_summaries, self.cfg.bounds_carry),
                                                            lambda: ops.KnnBounds(self.N, self.M, self.FX.device, summaries=self.cfg.bounds_summaries))
                            self._bounds_refused = self.bounds is None      # over the budget / out of memory: plain pruned emit sweeps from here on (not retried every step)
                        else:
                            self.bounds.advance(self.motion2)
                    else:
                        self.bounds = None
                    ops.raw_knn_prune_plan(self.FX, k, self.prune, self.kws, row_order=self.row_order, bounds=self.bounds,
                                           straggler_rows=self.cfg.straggler_rows if (self._warm_mode == "prev" and self.cfg.emit and self.gemm == "same") else 0)
                    self._warm = 4 if (tight and self.cfg.emit) else 3
                if self.timers:                # the flop THIS sweep executes (bench.py prices every timed sweep with its own plan); no host synchronisation
                    if self._swept_on_side:    # (the "front" schedule sums the plan's counts on the side stream, under the sweep: two tiny launches less between plan and sweep)
                        self._swept_pending = True
                    else:
                        self.swept_log.append(ops.knn_prune_counts(self.N, self.M, self.d, k, self.kws).sum())
        else:
            self._featurise()
            self.ev_feat.record()
            self._await_style()
            self._warm = ops.raw_knn_prepare(self.FX, self.FY, k, ws=self.kws, style_unchanged=self.style_external_norms)
            if (self.cfg.cold_order and self.cfg.prune and self.cfg.row_order and self.N >= 4096 and self.style_perm is not None and feat_dtype_ok(self.FY)
                    and self.gemm == "same" and self.style_stream is None):
                # the first sweep: no bound, so the ORDER of the tiles is everything (LoopConfig.cold_order).  The content rows' tree order — built after the first sweep
                # until round 4 — is what makes a block of 512 rows a neighbourhood; it only needs the feature rows
                if self.row_order is None:
                    self._row_seq = self._content_tree().to(torch.int32).contiguous()
                    self.row_order = deal_row_groups(self._row_seq) if self.cfg.row_deal else self._row_seq
                if self.prune is None or self.prune.key != self._fy_key():
                    self.prune = self._make_prune()
                    self.bounds = None
                ops.raw_knn_plan_nearest_first(self.FX, k, self.prune, self.kws, row_order=self.row_order)
                self._warm = 3

    def _geometry_match(self):
        """The fused distance-GEMM + top-k, the NN loss, and the fused backward (dL/dxyz of the geometry leg into grads[0])."""
        M, k = self.FY.shape[0], self.k
        if self.style_stream is not None:
            ev = self._match_chunked()
        elif self.gemm == "fp8":
            ev = ops.raw_knn_fp8_phase(self.qX, self.sX, self.qY, self.sY, k, 2, self._warm, idx=self.nn_idx, dist2=self.nn_d2, timed=self.timers, ws=self.kws, row_order=self.row_order,
                                       bounds=self.bounds if self._warm == 4 else None, prune=self.prune)
        else:
            if self.cu_reserve:                 # a sharded loop's rank: CUs left to the render leg for the duration of this launch (the setting is process-wide: reset in the finally)
                ops.knn_set_cu_reserve(self.cu_reserve)
            try:
                ev = ops.raw_knn_match_prepared(self.FX, self.FY, k, self.nn_idx, self.nn_d2, self._warm, timed=self.timers, ws=self.kws, row_order=self.row_order,
                                                bounds=self.bounds if self._warm == 4 else None, prune=self.prune)
            finally:
                if self.cu_reserve:
                    ops.knn_set_cu_reserve(0)
        if ev is not None:
            self.knn_events.append(ev[:2])    # HIP events around the fused distance-GEMM + top-k launch group (bench.py roofline)
            if len(ev) == 4:
                self.knn_kernel_events.append(ev[2:])      # ... and around its emit kernel alone
        # rows this sweep redid with a cold bound (normally 0) and the sticky overflow word: fetched asynchronously after EVERY sweep (bf16, f16
        # and fp8 alike), looked at one iteration late; more flagged rows than the repair takes raises W3dError
        self.kws.check()
        if self._warm:
            self.kws.poll(self.N, M, k)
        self.ev_knn.record()
        if self.cfg.row_order and self.N >= 4096 and (self.row_order is None or (self.cfg.row_order_refresh > 0 and self._sweeps > 0 and self._sweeps % self.cfg.row_order_refresh == 0)):
            # after the sweep (the kernel has read its order): the new order is used from the next iteration on.  First order: the tree on the rows' top principal
            # components (126 ms at C3, once).  Refreshes: see LoopConfig.row_order_kind — a sort of N integers, so they can be frequent
            kind = self.cfg.row_order_kind
            old_order = self.row_order
            # (_row_seq: the order as the refreshes see it; row_order: what the kernels take — the same, or with the 16-row groups of every block dealt to the waves)
            if self.row_order is None or kind == "tree":
                self._row_seq = self._content_tree().to(torch.int32).contiguous()
            elif kind == "nn0":
                self._row_seq = torch.argsort(self.nn_idx[:, 0], stable=True).to(torch.int32).contiguous()
            else:
                cur = self._row_seq.long()
                self._row_seq = cur[torch.argsort(self.nn_idx[cur, 0] >> 6, stable=True)].to(torch.int32).contiguous()
            self.row_order = deal_row_groups(self._row_seq) if self.cfg.row_deal else self._row_seq
            if self.bounds is not None:
                # the cached lower bounds are keyed by position in the order.  bounds_carry: they follow their rows into the new order (one pass over the cache);
                # otherwise a reset — the next sweep takes what the ball test keeps and refills them
                if self.cfg.bounds_carry and old_order is not None:
                    self.bounds.reorder(old_order, self.row_order)
                else:
                    self.bounds.reset()
        self._sweeps += 1
        self.have_prev = True
        scale = self.w_nn / (float(self.N_total) * float(self.M_true))      # mean over ALL content rows x style rows (N_total > N when Gaussians are sharded)
        gx = self.grads[0] if self.gx_scene is None else self.gx_scene
        if self.gx_scene is not None:
            self.gx_scene.zero_()
        if self.csr is not None:
            ops.raw_knn_loss(self.nn_d2, scale, self.loss[0:1])
            ops.raw_knn_match_bwd(self.FX, self.FY, self.nn_idx, scale, self.gFX, accumulate=False, row_order=self.row_order)
            self.csr.backward_x4(self.xyz4, self.gFX, gx, accumulate=True)
        else:                                                     # loss head + NN backward + descriptor scatter: one launch
            fya = self._fy_absmax()
            sc = None
            if self.cfg.nn_sum_cache and self.slots is not None and self.d <= 256 and k <= 64:
                # neighbour-sum cache: keyed by the style rows (tensor, in-place version, shape); rows whose neighbour set did not change skip their k gathers
                key = self._fy_key()
                if not self._sum_cache_refused and (self._sum_cache is None or (self._sum_cache.R, self._sum_cache.d, self._sum_cache.k) != (self.idx.shape[0], self.d, k)):
                    self._sum_cache = self._alloc_cache(ops.NeighbourSumCache.bytes_needed(self.idx.shape[0], self.d, k),
                                                        lambda: ops.NeighbourSumCache(self.idx.shape[0], self.d, k, self.FX.device))
                    self._sum_cache_of = key
                    self._sum_cache_refused = self._sum_cache is None
                elif self._sum_cache is not None and self._sum_cache_of != key:
                    self._sum_cache.invalidate(); self._sum_cache_of = key
                sc = self._sum_cache
            with self._stage("fused_bwd"):
                ops.raw_knn_descriptor_bwd_fused(self.xyz4, self.idx, self.FX, self.FY, fya, self.nn_idx, scale, gx, self.max_fanin,
                                                 nn_dist2=self.nn_d2, loss_accum=self.loss[0:1], slots=self.slots, sum_cache=sc)
        if self.exchange is not None and self.gx_scene is not None:
            self.exchange.scatter_gx(self.gx_scene, self.grads[0])   # every rank's contributions to MY Gaussians (reduce-scatter, 12 B per Gaussian of the scene)

    def _match_chunked(self):
        """The distance-GEMM over a style set that is still arriving (distributed.StyleStream): the own shard is swept at once (no transfer), the shards
        before and after it — by then in place: their events — in one sweep each, every sweep into its own slot of the partial lists; one merge at the
        end returns exactly what the sweep of the whole set returns."""
        st = self.style_stream
        k = self.k
        groups = st.groups()                    # lists of consecutive shards swept together: [own] first (no transfer), then what lies before / after it
        parts = len(groups)
        if getattr(self, "_part_idx", None) is None or self._part_idx.shape[1] != parts:
            self._part_idx = torch.empty(self.N, parts, k, dtype=torch.int32, device=self.FX.device)
            self._part_val = torch.empty(self.N, parts, k, dtype=torch.float32, device=self.FX.device)
        ev = None
        if self.timers:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
        for g, shards in enumerate(groups):
            for c in shards:
                st.wait(c)                                          # the current stream waits for shard c's broadcast (own shard: nothing to wait for)
                m0, m_valid, m1 = st.chunk_rows(c)
                ops.raw_knn_chunk_norms(self.FX, self.FY, k, m0, m_valid, m1, ws=self.kws)
            g0, g1 = st.chunk_rows(shards[0])[0], st.chunk_rows(shards[-1])[2]
            ops.raw_knn_chunk_sweep(self.FX, self.FY, k, self._warm, g0, g1, g, parts, self._part_idx, self._part_val, ws=self.kws, row_order=self.row_order)
        ops.raw_knn_chunk_merge(self.FX, self.FY, k, self._warm, parts, self._part_idx, self._part_val, self.nn_idx, self.nn_d2, ws=self.kws, row_order=self.row_order)
        if self.timers:
            e1.record(); ev = (e0, e1)
        return ev

    def _geometry_leg(self, xyz):
        self._geometry_prepare(xyz)
        self._geometry_match()

    def _render_view(self, xyz, shs, cam, depth_target, wv):
        """Forward + image losses (weights scaled by wv = 1/V) + backward of ONE view; returns the raw rasteriser gradients."""
        pc = self.pc
        offs = self._offsets()
        e = torch.empty(0, device=xyz.device)
        tx, ty = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
        R, color, depth, radii, geom, binning, img = dgr.rasterize_gaussians_fwd(
            self.bg, xyz, e, self.act_o, self.act_s, self.act_q, 1.0, e, cam.world_view_transform, cam.full_proj_transform, tx, ty, self.H, self.W,
            shs, pc.active_sh_degree, cam.camera_center, False, False, offs)
        self.last_R += R
        ops.raw_mse_loss(depth, depth_target, self.w_depth * wv, self.loss[1:2], self.g_depth, accumulate=False)
        ops.raw_tv_loss(color, self.w_tv * wv, self.loss[2:3], self.g_color, accumulate=False)
        return dgr.rasterize_gaussians_backward(
            self.bg, xyz, radii, e, self.act_s, self.act_q, 1.0, e, cam.world_view_transform, cam.full_proj_transform, tx, ty, self.g_color,
            self.g_depth, shs, pc.active_sh_degree, cam.camera_center, geom, R, binning, img, False, offs)

    def _render_leg(self, xyz):
        """Writes the render-leg gradients into grads[1..5] and the xyz part into self.g_xyz_render (summed with grads[0] inside Adam)."""
        if len(self.cams) == 1:
            return self._render_leg_direct(xyz)
        if self.cfg.multiview_direct:
            # V views per iteration (BASELINE C5: 4) on the same persistent buffers as the one-view path: forward + image losses (weights / V) + backward view after
            # view, the backward of views 1 ... V-1 ADDING to the per-Gaussian gradients (w3d_raster_backward_accumulate) — no SH concatenation, no per-view gradient
            # arrays, no sums; one activation backward at the end
            V = len(self.cams)
            R = 0
            for v in range(V):
                self._render_direct_forward(xyz, view=v, wv=1.0 / V)
                R += self.last_R
                self._render_direct_backward(xyz, accumulate=v > 0, activations=v == V - 1)
            self.last_R = R
            return
        pc = self.pc
        N = self.N
        L = lib(); s = stream_ptr()
        check(L.w3d_activations_fwd(ptr(pc._scaling.data), ptr(pc._rotation.data), ptr(pc._opacity.data), N, ptr(self.act_s), ptr(self.act_q), ptr(self.act_o), s))
        shs = torch.cat((pc._features_dc.data, pc._features_rest.data), dim=1)
        self.last_R = 0
        V = len(self.cams)
        tot = None
        for cam, tgt in zip(self.cams, self.depth_targets):
            g = self._render_view(xyz, shs, cam, tgt, 1.0 / V)
            if tot is None:
                tot = g
            else:                                                  # V > 1: the per-view gradients (plain [P,*] arrays) are summed before the activation backward
                for a, b in zip(tot, g):
                    a.add_(b)
        g_m2, g_col, g_op, g_m3, g_cov, g_sh, g_sc, g_rot = tot
        check(L.w3d_activations_bwd(ptr(pc._scaling.data), ptr(pc._rotation.data), ptr(pc._opacity.data), N, ptr(g_sc), ptr(g_rot), ptr(g_op),
                                    ptr(self.grads[4]), ptr(self.grads[5]), ptr(self.grads[3]), s))
        self.g_xyz_render.copy_(g_m3)
        self.grads[1].copy_(g_sh[:, :1]); self.grads[2].copy_(g_sh[:, 1:])

    def _render_leg_direct(self, xyz):
        self._render_direct_forward(xyz)
        self._render_direct_backward(xyz)

    def _render_direct_forward(self, xyz, view: int = 0, wv: float = 1.0):
        """One view per iteration (the reference's loops), straight on the C ABI with persistent buffers: the DC and higher-order SH parameter
        tensors are read and their gradients written in place (W3dRasterIn.shs_rest / dL_dsh_rest: no concatenation, no copies), dL/dxyz lands
        in g_xyz_render (second gradient pointer of the Adam step), and the raw per-Gaussian gradients that must read 0 for culled Gaussians
        live in ONE buffer zeroed by one fill.  Gradient buffers the Adam step zeroes (grads[1], grads[2], g_xyz_render) need no fill."""
        import ctypes as C
        pc, N, H, W = self.pc, self.N, self.H, self.W
        dev = xyz.device
        L = lib(); s = stream_ptr()
        if getattr(self, "_rb", None) is None:
            # g_op 1 | g_sc 3 | g_rot 4 | accum 12 (round 6: dL/dmeans2D, dL/dcolors and dL/dcov3D — 12 floats per Gaussian nobody in this loop reads — are not
            # requested from the per-Gaussian backward any more (NULL), and are neither zero-filled nor written)
            # (round 6: zero-filled ONCE — w3d_raster_backward_selfclean leaves the accumulators zero and writes the three per-Gaussian outputs the activation backward reads
            # for every Gaussian, culled ones as zeros; until then an 80-MB fill sat between the render leg's forward and backward in every iteration)
            z = torch.zeros(N * 20, dtype=torch.float32, device=dev)
            cuts = [0, 1, 4, 8, 20]
            g_op_, g_sc_, g_rot_, accum_ = [z[N * a:N * b] for a, b in zip(cuts[:-1], cuts[1:])]
            parts = [None, None, g_op_, None, g_sc_, g_rot_, accum_]
            self._rb = dict(raw=z, parts=parts, geom=torch.empty(L.w3d_raster_geom_bytes(N), dtype=torch.uint8, device=dev),
                            img=torch.empty(L.w3d_raster_image_bytes(H, W), dtype=torch.uint8, device=dev), color=torch.empty(3, H, W, device=dev),
                            depth=torch.empty(H, W, device=dev), radii=torch.empty(N, dtype=torch.int32, device=dev), binning=None)

            def grow(_u, nbytes):                                                 # binning buffer: kept, regrown only when the tile-instance count outgrows it
                b = self._rb["binning"]
                if b is None or b.numel() < nbytes:
                    self._rb["binning"] = torch.empty(int(nbytes * 1.5), dtype=torch.uint8, device=dev)
                return self._rb["binning"].data_ptr()
            self._rb["cb"] = _lib.ALLOC_FN(grow)
        rb = self._rb
        g_m2, g_col, g_op, g_cov, g_sc, g_rot, accum = rb["parts"]
        cam = self.cams[view]
        if view == 0:
            check(L.w3d_activations_fwd(ptr(pc._scaling.data), ptr(pc._rotation.data), ptr(pc._opacity.data), N, ptr(self.act_s), ptr(self.act_q), ptr(self.act_o), s))
        offs = self._next_offs if self._next_offs is not None else self._offsets()      # (drawn behind the previous backward: two tiny launches less at the head of the leg's chain)
        self._next_offs = None
        mats = rb.setdefault("mats", {})
        if mats.get(view, (None,))[0] is not cam:                  # the camera's matrices are transposed VIEWS (scene/cameras.py:54-57): contiguous copies, once per camera
            mats[view] = (cam,) + tuple(t.detach().to(dev).float().contiguous() for t in (cam.world_view_transform, cam.full_proj_transform, cam.camera_center))
        _, view_m, proj, campos = mats[view]
        rin = _lib.RasterIn(N, pc.active_sh_degree, 1 + pc._features_rest.shape[1], H, W, ptr(self.bg), ptr(xyz), ptr(pc._features_dc.data), None, ptr(self.act_o),
                            ptr(self.act_s), ptr(self.act_q), None, ptr(view_m), ptr(proj), ptr(campos), ptr(offs),
                            1.0, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), 0, ptr(pc._features_rest.data))
        n = C.c_int(0)
        check(L.w3d_raster_forward(C.byref(rin), ptr(rb["geom"]), ptr(rb["img"]), rb["cb"], None, ptr(rb["color"]), ptr(rb["depth"]), ptr(rb["radii"]), C.byref(n), s),
              "w3d_raster_forward")
        self.last_R = n.value
        ops.raw_mse_loss(rb["depth"], self.depth_targets[view], self.w_depth * wv, self.loss[1:2], self.g_depth, accumulate=False)
        ops.raw_tv_loss(rb["color"], self.w_tv * wv, self.loss[2:3], self.g_color, accumulate=False)
        if not self.cfg.raster_selfclean:                            # (A/B: rounds 4-6's fills — everything before the first view, the blend backward's scratch before the others)
            if view == 0:
                rb["raw"].zero_()
            else:
                accum.zero_()
        rb["rin"], rb["R"], rb["offs"] = rin, n.value, offs          # the backward re-reads the offsets through rin: keep them alive

    def _render_direct_backward(self, xyz, accumulate: bool = False, activations: bool = True):
        """Blend + per-Gaussian backward of the view rendered by _render_direct_forward (accumulate: added to the gradients of the iteration's earlier views), and — after
        the last view — the activation backward."""
        import ctypes as C
        pc, N = self.pc, self.N
        L = lib(); s = stream_ptr()
        rb = self._rb
        g_m2, g_col, g_op, g_cov, g_sc, g_rot, accum = rb["parts"]
        rin = rb["rin"]
        gin = _lib.RasterGradIn(ptr(self.g_color), ptr(self.g_depth))
        gout = _lib.RasterGradOut(ptr(g_m2), ptr(g_col), ptr(g_op), ptr(self.g_xyz_render), ptr(g_cov), ptr(self.grads[1]), ptr(g_sc), ptr(g_rot), ptr(accum),
                                  ptr(self.grads[2]))
        if self.cfg.raster_selfclean:
            check(L.w3d_raster_backward_selfclean(C.byref(rin), ptr(rb["radii"]), C.byref(gin), C.byref(gout), ptr(rb["geom"]), ptr(rb["binning"]), ptr(rb["img"]), rb["R"],
                                                  1 if accumulate else 0, s), "w3d_raster_backward_selfclean")
            self._next_offs = self._offsets()                        # the next forward's sampling offsets (same generator, same order of draws)
        else:
            fn = L.w3d_raster_backward_accumulate if accumulate else L.w3d_raster_backward
            check(fn(C.byref(rin), ptr(rb["radii"]), C.byref(gin), C.byref(gout), ptr(rb["geom"]), ptr(rb["binning"]), ptr(rb["img"]), rb["R"], s), "w3d_raster_backward")
        if not activations:
            return
        check(L.w3d_activations_bwd(ptr(pc._scaling.data), ptr(pc._rotation.data), ptr(pc._opacity.data), N, ptr(g_sc), ptr(g_rot), ptr(g_op),
                                    ptr(self.grads[4]), ptr(self.grads[5]), ptr(self.grads[3]), s))

    def step(self, serial: bool = False):
        """Two HIP streams; the legs meet in Adam.  The distance-GEMM's blocks hold every VGPR and all LDS of a CU, so a kernel of the other
        stream only gets in at its block boundaries and, if it is large, stalls whole rounds (19.8 ms alone vs 22.4 ms with the whole render leg
        started together with it).  Schedules (LoopConfig.schedule):
          "front" (one view, the default since round 3): see the branch below.
          "early" (round 2's default): render FORWARD on the high-priority stream from t = 0, next to the memory-bound featurisation — its ~25 short
              launches are through before the GEMM fills the chip, the blend forward overlaps the GEMM's first round; render BACKWARD gated on the
              GEMM's end, next to the fused geometry backward (complementary: L2 gathers vs VALU/atomics).
          "overlap" (multi-view and sharded loops, or on request): the render leg starts after the featurisation; its small launches trickle through the
              GEMM's block boundaries, both blend kernels land next to the fused geometry backward.
        (Round 2 also carried a "phased" schedule and another "front" (the sweep between two halves of the render leg); both were removed in round 3.)"""
        _lib.clear_stale_hip_error()                                 # a stale error of another HIP user of this thread is not this step's
        xyz = self.pc._xyz.data
        main = torch.cuda.current_stream()
        self.loss.zero_()
        if serial:
            # every stage alone on the chip, one stream, in dependency order (a measurement aid: bench.py's stand-alone stage times; same arithmetic)
            assert len(self.cams) == 1
            self._geometry_prepare(xyz)
            with self._stage("render_fwd"):
                self._render_direct_forward(xyz)
            self._geometry_match()
            self.t += 1
            self.lrs[0] = float(self.xyz_sched(self.t))
            groups = [(p.data, g, m, v, lr) for p, g, m, v, lr in zip(self.params, self.grads, self.m, self.v, self.lrs)]
            with self._stage("render_bwd"):
                self._render_direct_backward(xyz)
            with self._stage("adam_render_groups"):
                ops.raw_adam_step(groups[1:], self.t, 0.9, 0.999, 1e-15, zero_grad=True)
            with self._stage("adam_xyz"):
                ops.raw_adam_step([groups[0] + (self.g_xyz_render,)], self.t, 0.9, 0.999, 1e-15, zero_grad=True)
            return
        if len(self.cams) == 1 and self.schedule == "front":
            # Round 3 (the persistent sweep holds every CU for its whole duration, so nothing of the other stream runs next to it any more): the whole render leg —
            # forward, backward, Adam of the five groups only it feeds — runs from t = 0 next to the geometry leg's memory-bound prologue (featurise, bound, plan);
            # the sweep starts when both are through and has the chip to itself, and so has the fused backward after it.
            self.side.wait_stream(main)                              # the last Adam step
            self._swept_on_side = self.cfg.adam_under_sweep
            try:
                self._geometry_prepare(xyz)                          # main: featurise + bound + plan (enqueued; asynchronous)
            finally:
                self._swept_on_side = False
            self.t += 1
            self.lrs[0] = float(self.xyz_sched(self.t))
            groups = [(p.data, g, m, v, lr) for p, g, m, v, lr in zip(self.params, self.grads, self.m, self.v, self.lrs)]
            with torch.cuda.stream(self.side):
                with self._stage("render_fwd"):
                    self._render_direct_forward(xyz)                 # (the host blocks once on the tile-instance count)
                with self._stage("render_bwd"):
                    self._render_direct_backward(xyz)
                if not self.cfg.adam_late and not self.cfg.adam_under_sweep:
                    with self._stage("adam_render_groups"):
                        ops.raw_adam_step(groups[1:], self.t, 0.9, 0.999, 1e-15, zero_grad=True)
            main.wait_stream(self.side)                              # the sweep waits for the render leg (it could not run next to it anyway)
            if self.cfg.adam_under_sweep:
                self.ev_pre_sweep.record(main)
                with torch.cuda.stream(self.side):
                    self.side.wait_event(self.ev_pre_sweep)
                    with self._stage("adam_render_groups"):
                        ops.raw_adam_step(groups[1:], self.t, 0.9, 0.999, 1e-15, zero_grad=True, lean=True)
                    if self._swept_pending:                          # (timers only) the plan's tile counts, summed behind the Adam stream; main joins this stream before the next plan
                        self.swept_log.append(ops.knn_prune_counts(self.N, self.M, self.d, self.k, self.kws).sum())
                        if self.cfg.straggler_rows:
                            self.strag_log.append(ops.knn_straggler_counts(self.N, self.M, self.d, self.k, self.kws).sum())
                        self._swept_pending = False
            self._geometry_match()                                   # main: distance-GEMM + top-k (the lean Adam stream under it), loss, fused backward (alone on the chip)
            if self.cfg.adam_late and not self.cfg.adam_under_sweep:
                with torch.cuda.stream(self.side):
                    self.side.wait_event(self.ev_knn)                # after the sweep: a pure HBM stream next to the fused backward's Infinity-Cache gathers
                    with self._stage("adam_render_groups"):
                        ops.raw_adam_step(groups[1:], self.t, 0.9, 0.999, 1e-15, zero_grad=True)
            with self._stage("adam_xyz"):
                ops.raw_adam_step([groups[0] + (self.g_xyz_render,)], self.t, 0.9, 0.999, 1e-15, zero_grad=True)
            if self.cfg.adam_late or self.cfg.adam_under_sweep:
                # that Adam runs on the (non-blocking) side stream — under the sweep, or (adam_late alone) next to the fused backward: the caller's stream joins it before
                # step() returns, so that whatever is enqueued next on it — save_ply, an eval render, a clone of the parameters, step(serial=True) — is ordered after EVERY
                # parameter update of this step.  Free: it is through long before the fused backward that main runs meanwhile.
                main.wait_stream(self.side)
            return
        if len(self.cams) == 1 and self.schedule == "early":
            # render FORWARD from t = 0 next to the memory-bound featurisation (its ~25 short launches finish before the GEMM fills the chip),
            # render BACKWARD gated on the GEMM's end, next to the fused geometry backward
            self.side.wait_stream(main)                              # the last Adam step
            self._geometry_leg(xyz)                                  # enqueued first (asynchronous); the host then blocks on the tile count below
            with torch.cuda.stream(self.side), self._stage("render_fwd"):
                self._render_direct_forward(xyz)
            self.side.wait_event(self.ev_knn)                        # render backward gated on the GEMM's end
            self.t += 1
            self.lrs[0] = float(self.xyz_sched(self.t))
            groups = [(p.data, g, m, v, lr) for p, g, m, v, lr in zip(self.params, self.grads, self.m, self.v, self.lrs)]
            with torch.cuda.stream(self.side):
                with self._stage("render_bwd"):
                    self._render_direct_backward(xyz)
                # the five groups only the render leg feeds (56 of the 59 floats per Gaussian) are stepped here, in the slack next to the fused
                # geometry backward (0.05 ms per step, same-device A/B); the positions (both legs) after the join
                with self._stage("adam_render_groups"):
                    ops.raw_adam_step(groups[1:], self.t, 0.9, 0.999, 1e-15, zero_grad=True)
            main.wait_stream(self.side)
            with self._stage("adam_xyz"):
                ops.raw_adam_step([groups[0] + (self.g_xyz_render,)], self.t, 0.9, 0.999, 1e-15, zero_grad=True)
            return
        else:
            self._geometry_leg(xyz)                                  # enqueued first: asynchronous, keeps the device busy while the host
            self.side.wait_event(self.ev_feat)
            with torch.cuda.stream(self.side):                       # blocks on the render leg's tile-instance count
                self._render_leg(xyz)
        main.wait_stream(self.side)
        # --- Adam (six groups, one launch, zero_grad fused); dL/dxyz = geometry leg (grads[0]) + render leg (g_xyz_render), summed in the step ---
        self.t += 1
        self.lrs[0] = float(self.xyz_sched(self.t))
        groups = [(p.data, g, m, v, lr) for p, g, m, v, lr in zip(self.params, self.grads, self.m, self.v, self.lrs)]
        groups[0] = groups[0] + (self.g_xyz_render,)
        ops.raw_adam_step(groups, self.t, 0.9, 0.999, 1e-15, zero_grad=True)

    @property
    def nn_idx_original(self):
        """The current neighbours in the caller's numbering of the style rows."""
        return self.nn_idx if self.style_perm is None else self.style_perm[self.nn_idx.long()].to(torch.int32)

    @property
    def repaired_rows_max(self):
        """Largest number of rows any warm-started sweep of this loop had to redo with a cold bound (blocks for the last poll)."""
        self.kws.check(wait=True)
        return self.kws.max_redone

    def knn_ms(self):
        """Average duration of the fused kNN launch group over the recorded steps (HIP events on the launch stream)."""
        ts = self.knn_ms_per_step()
        return sum(ts) / max(len(ts), 1)

    def knn_ms_per_step(self):
        """Duration of the fused kNN launch group of every recorded step, in order; clears the records."""
        torch.cuda.synchronize()
        ts = [a.elapsed_time(b) for a, b in self.knn_events]
        self.knn_events = []
        return ts

    def knn_kernel_ms_per_step(self):
        """Duration of the emit kernel alone inside every recorded step's launch group ([] when the sweeps are not pruned emit sweeps with cached bounds); clears the records."""
        torch.cuda.synchronize()
        ts = [a.elapsed_time(b) for a, b in self.knn_kernel_events]
        self.knn_kernel_events = []
        return ts

    def straggler_pairs_per_step(self):
        """(row, tile) pairs each recorded plan sent the straggler way, in order; clears the records."""
        out = [int(t.item()) for t in self.strag_log]
        self.strag_log = []
        return out

    def swept_pairs_per_step(self):
        """(row block, style tile) pairs each recorded PRUNED sweep kept, in order ([] when the sweeps are not pruned); clears the records."""
        out = [int(t.item()) for t in self.swept_log]
        self.swept_log = []
        return out


def morton_order(xyz: torch.Tensor, bits: int = 10) -> torch.Tensor:
    """Permutation that sorts points along a Z-order curve (setup-time layout choice: neighbours in space become neighbours in memory,
    so the three gather passes of the geometry leg hit L2 instead of HBM)."""
    lo, hi = xyz.min(0).values, xyz.max(0).values
    q = ((xyz - lo) / (hi - lo + 1e-12) * (2 ** bits - 1)).long().clamp_(0, 2 ** bits - 1)
    code = torch.zeros(xyz.shape[0], dtype=torch.int64, device=xyz.device)
    for b in range(bits):
        for c in range(3):
            code |= ((q[:, c] >> b) & 1) << (3 * b + c)
    return torch.argsort(code)


def windowed_knn_lists(xyz: torch.Tensor, K: int, window: int = 4096, chunk: int = 8192) -> torch.Tensor:
    """Neighbour lists for points ALREADY in Morton order restricted to the +-window positions around each row (self first): the
    exact K nearest (ops.knn_lists) of each overlapping slab, cut to the rows whose whole window lies inside the slab.
    The index locality this guarantees is what the LDS window of the fused backward is sized for; `exact_knn_lists` has no such bound."""
    N = xyz.shape[0]
    out = torch.empty(N, K, dtype=torch.int32, device=xyz.device)
    for c0 in range(0, N, chunk):
        c1 = min(c0 + chunk, N)
        w0, w1 = max(0, c0 - window), min(N, c1 + window)
        if w1 - w0 < K:
            w0, w1 = max(0, min(w0, N - K)), min(N, max(w1, K))
        slab = xyz[w0:w1].contiguous()
        loc = ops.knn_lists(xyz[c0:c1].contiguous(), K, Y=slab)
        out[c0:c1] = loc + w0
        # column 0 = the root: its own point is at distance 0 and leads unless an exact duplicate with a lower index precedes it
        rows = torch.arange(c0, c1, device=xyz.device, dtype=torch.int32)
        blk = out[c0:c1]
        bad = torch.nonzero(blk[:, 0] != rows).flatten()
        if bad.numel():
            sub = blk[bad].clone()
            pos = (sub == rows[bad, None]).int().argmax(1)
            head = sub[:, 0].clone()
            sub[torch.arange(sub.shape[0], device=sub.device), pos] = head
            sub[:, 0] = rows[bad]
            blk[bad] = sub
    return out


def exact_knn_lists(xyz: torch.Tensor, K: int) -> torch.Tensor:
    """The exact K nearest neighbours of every point (self first), any order of the points (NB11:1055-1056 at any size)."""
    return ops.knn_lists(xyz, K)


def deal_row_groups(order: torch.Tensor) -> torch.Tensor:
    """LoopConfig.row_deal: positions xb * 128 + w * 16 + r of every full block of 512 move to w * 64 + xb * 16 + r (w < 8, xb < 4, r < 16)."""
    n = (order.shape[0] // 512) * 512
    if n == 0:
        return order
    head = order[:n].view(-1, 4, 8, 16).permute(0, 2, 1, 3).reshape(-1)
    return torch.cat([head, order[n:]]).contiguous()


def _tensor_key(t: torch.Tensor):
    """Identity of a tensor's CONTENT as far as torch can tell without reading it: storage address, in-place modification counter, shape."""
    return (t.data_ptr(), t._version, tuple(t.shape))


def _dequant_bf16(q: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    """e4m3 bytes x per-row power-of-two scale (E8M0 byte) as bf16 rows: exact (3 mantissa bits, a power-of-two factor)."""
    return (q.view(torch.float8_e4m3fn).to(torch.bfloat16) * torch.pow(2.0, (s.float() - 127.0)).to(torch.bfloat16)[:, None]).contiguous()


def feat_dtype_ok(F: torch.Tensor) -> bool:
    return F.dtype in (torch.bfloat16, torch.float16) and F.shape[1] in (64, 128, 256)


def descriptor_tree_order(F: torch.Tensor, leaf: int = 64, iters: int = 6, proj_dim: int = 0) -> torch.Tensor:
    """Permutation that stores similar feature rows next to each other: recursive bisection of the row set along its principal axis (power iteration
    per node, all nodes of a level at once) down to leaves of <= `leaf` rows — the order of the leaves of that tree.  Setup-time layout choice for the
    STYLE rows (they have no order of their own): a content row's k nearest style rows then sit in ~9 of the kNN kernel's 64-row tiles instead of ~20
    (bench generator), so the selection's slow path runs in ~16 % of the 32x32 score blocks instead of ~27 %, and the k gathered rows of the
    bound / backward passes share cache lines' neighbourhoods.  The match itself is unchanged (same rows, other numbering)."""
    X = F.float()
    M, d = X.shape
    dev = X.device
    if proj_dim and d > proj_dim and M > 4 * d:
        # the tree on the rows' top `proj_dim` principal components (one d x d covariance GEMM + eigh, one [M,d] x [d,p] projection): every level then moves
        # p / d of the bytes — 1M x 256 rows: 520 ms -> ~50 ms, which is what makes refreshing the CONTENT rows' order during a run affordable.  The
        # descriptors' spectrum decays fast (singular values 1, 0.20, 0.11, 0.08, ...): the leaves change little.
        Xm = X - X.mean(0, keepdim=True)
        cov = Xm.T @ Xm
        _, vec = torch.linalg.eigh(cov)                               # ascending eigenvalues
        X = (Xm @ vec[:, -proj_dim:]).contiguous()
        d = proj_dim
    order = torch.arange(M, device=dev)
    node = torch.zeros(M, dtype=torch.long, device=dev)             # node id of every row (in `order` numbering), nodes numbered left to right
    n_nodes, largest = 1, M
    g = torch.Generator(device=dev).manual_seed(0)
    def node_sums(V, start, cnt):
        # per-node sums of the rows of V: the rows are sorted by node and the nodes of a level differ in size by at most one row, so a level is a padded
        # [nodes, rows per node, d] block — a gather and a reduction (index_add_'s float atomics on a few hot rows took 0.7-1.3 ms per call at 1M rows, 145 calls
        # per loop set-up, and their summation order changed from run to run)
        width = -(-M // n_nodes)                                     # a split gives ceil(c / 2) and floor(c / 2) rows: the largest node of a level has ceil(M / nodes)
        pos = start[:, None] + torch.arange(width, device=dev)[None, :]
        live = pos < (start + cnt)[:, None]
        return (V[pos.clamp_max(V.shape[0] - 1)] * live[:, :, None]).sum(1)
    while largest > leaf:
        Xo = X[order]
        cnt = torch.bincount(node, minlength=n_nodes).clamp_min(1)
        start = torch.cumsum(cnt, 0) - cnt                           # first position of every node (all nodes are non-empty)
        mean = node_sums(Xo, start, cnt) / cnt[:, None]
        Xc = Xo - mean[node]
        v = torch.randn(n_nodes, d, device=dev, generator=g)
        for _ in range(iters):                                       # top principal direction of every node: v <- Xc^T (Xc v), node-wise
            y = (Xc * v[node]).sum(1)
            v = node_sums(Xc * y[:, None], start, cnt)
            v = v / v.norm(dim=1, keepdim=True).clamp_min(1e-30)
        proj = (Xc * v[node]).sum(1)
        # sort rows by (node, projection): stable sort by node of the projection-sorted rows
        o1 = torch.argsort(proj, stable=True)
        o2 = torch.argsort(node[o1], stable=True)
        o = o1[o2]
        order, node = order[o], node[o]
        rank = torch.arange(M, device=dev) - start[node]
        node = 2 * node + (rank >= (cnt[node] + 1) // 2).long()
        n_nodes *= 2
        largest = (largest + 1) // 2
    return order


def content_tree_order(F: torch.Tensor, leaf: int = 256, proj_dim: int = 3, power_steps: int = 6, basis=None, return_basis: bool = False):
    """The processing order of the CONTENT rows (round 6): the order of the leaves of a principal-axis bisection tree — as descriptor_tree_order — over the rows' top `proj_dim`
    principal components, built DENSELY: the rows are padded to nodes x width, so a level is a [nodes, width] segmented sort plus a handful of dense ops on a [nodes, width, p]
    tensor (no index arithmetic, no two-key global sorts, the exact p x p covariance of every node instead of a power iteration over its rows).  1M x 256 rows: ~10 ms against
    the 126 ms of descriptor_tree_order(proj_dim = 16) — and the order it replaces is the better one: on FEWER components the blocks of 512 rows come out compact in the
    directions that carry the distance (swept fraction of the C3 loop's pruned sweeps 0.146 -> 0.135 at the start of a run, 0.195 -> 0.165 over 1001 iterations with a
    refresh every 32: profiles/r06w_tree_ab.log).  A layout hint: every result is unchanged."""
    N, d = F.shape
    dev = F.device
    p = min(proj_dim, d)
    # top-p principal components from the d x d second-moment matrix (one GEMM on the rows as stored; f32 accumulation)
    # (`basis` = (mean [1, d], V [d, p]) of an earlier build: the rows drift slowly and the tree only needs SOME good low-dimensional view of them — the eigendecomposition of the
    # 256 x 256 matrix is a third of a build)
    Fm = F if F.dtype in (torch.bfloat16, torch.float16) else F.float()
    if basis is None:
        # from a strided sample of <= 65 536 rows, in f32 (a 16-bit GEMM over a million rows would return the second moments in 16 bits — f16 overflows at 65 504)
        Fs = Fm[:: max(1, N // 65536)].float()
        mean = Fs.mean(0, keepdim=True)                                  # [1, d]
        Fc = Fs - mean
        _, vec = torch.linalg.eigh(Fc.T @ Fc)
        V = vec[:, -p:].contiguous()                                     # [d, p]
        basis = (mean, V)
    mean, V = basis
    X = (Fm @ V.to(Fm.dtype)).float() - mean @ V                        # [N, p]
    levels = 0
    while (N + (1 << levels) - 1) >> levels > leaf:
        levels += 1
    nodes_final = 1 << levels
    width0 = -(-N // nodes_final) * nodes_final                          # padded row count: a multiple of the final node count
    pad = width0 - N
    order = torch.arange(width0, device=dev)
    live = torch.ones(width0, dtype=torch.bool, device=dev)
    if pad:
        X = torch.cat([X, torch.zeros(pad, p, device=dev)], 0)
        live[N:] = False
    for lv in range(levels):
        nodes = 1 << lv
        width = width0 >> lv
        Xn = X.view(nodes, width, p)
        w = live.view(nodes, width, 1).float()
        cnt = w.sum(1).clamp_min(1.0)                                    # [nodes, 1]
        mu = (Xn * w).sum(1) / cnt                                       # [nodes, p]
        Xc = (Xn - mu[:, None, :]) * w
        if p <= 4:                                                       # p (p + 1) / 2 plain reductions (a batched GEMM of these shapes takes 2.5x as long)
            C = torch.empty(nodes, p, p, device=dev)
            for i in range(p):
                for j in range(i, p):
                    C[:, i, j] = (Xc[:, :, i] * Xc[:, :, j]).sum(1)
                    if j != i:
                        C[:, j, i] = C[:, i, j]
        else:
            C = torch.einsum("nwi,nwj->nij", Xc, Xc)                     # [nodes, p, p]
        # principal direction of every node: repeated squaring of its (normalised) covariance, applied to the axis of largest variance
        C = C / C.diagonal(dim1=1, dim2=2).sum(1).clamp_min(1e-30)[:, None, None]
        for _ in range(power_steps):
            C = C @ C
            C = C / C.diagonal(dim1=1, dim2=2).sum(1).clamp_min(1e-30)[:, None, None]
        ax = C.diagonal(dim1=1, dim2=2).argmax(1)
        v = C[torch.arange(nodes, device=dev), :, ax]
        v = v / v.norm(dim=1, keepdim=True).clamp_min(1e-30)
        proj = (Xc * v[:, None, :]).sum(2)                               # [nodes, width]
        proj = torch.where(live.view(nodes, width), proj, torch.full_like(proj, float("inf")))     # padding sorts last in its node
        idx = proj.argsort(dim=1, stable=True)
        flat = (idx + torch.arange(nodes, device=dev)[:, None] * width).view(-1)
        order, live, X = order[flat], live[flat], X[flat]
    return (order[live], basis) if return_basis else order[live]


def centred_style_rows(FY32: torch.Tensor, dtype=torch.bfloat16):
    """(style rows shifted by their mean row and rounded to `dtype`, the mean row [d] f32).  The content rows must be featurised with the same
    `center` (GaussianStyleLoop(feat_center=...)): a common shift leaves every pairwise distance as it is and takes most of the 16-bit rounding
    error out of the match."""
    mu = FY32.float().mean(0).contiguous()
    return (FY32.float() - mu).to(dtype).contiguous(), mu


# Weight of the NN style term in the synthetic C3 loop.  The term is a mean over all N x M pairs (style_loop.py: scale = w_nn / (N M); NB11:1021), i.e. (k / M) x the mean selected
# squared distance — 1.8e-7 at C3 against a depth term of 6e-2: at weight 1 the style term is inert and the content drifts away from the style set.  The reference weights its
# terms so that all of them matter (1 / 2e2 / 3e1 on terms of 1e-3 / 7e-5 / 2e-4, NB11:1085 and SURVEY 8c's trace); 1e5 puts the weighted NN term within 10x of the depth term.
C3_W_NN = 1.0e5


def make_synthetic_c3(N=1_000_000, M=65536, d=256, k=20, H=800, W=800, device="cuda", seeds=(20, 21, 22, 23), timers=False, lists="knn", feat_dtype=torch.bfloat16, views=1,
                      style_density=None, center=True, w_nn=None, **kw):
    """The synthetic workload of SURVEY §8d C3 (all seeds fixed): uniform Gaussians in [-1.3,1.3]^3, nerf_synthetic
    camera (800x800, camera_angle_x 0.6911, radius 4.03), random frozen neighbour lists, style features = descriptors of a
    synthetic style patch, depth target = blurred initial depth."""
    from .camera import Camera
    style_density = style_density or "sparse"
    pc = GaussianModel.synthetic(N, seed=seeds[0], device=device)
    pc.spatial_lr_scale = 1.0
    gs = torch.Generator().manual_seed(seeds[3])
    ys = (torch.rand(M, 3, generator=gs) * 2.6 - 1.3).to(device)
    if style_density == "matched":          # style patch with the content's point density: its descriptors resemble the content's, every
        ys = ys * (M / N) ** (1.0 / 3.0)     # content row has its own neighbours (the default patch spans the content's extent with M points)
    if lists in ("knn", "exact"):
        # frozen descriptor lists = the d nearest neighbours of a frozen copy, Gaussians stored in Z-order;
        # "knn": nearest within +-4096 positions of the curve (bounded index distance), "exact": the true nearest neighbours
        perm = morton_order(pc._xyz.data)
        for p in (pc._xyz, pc._features_dc, pc._features_rest, pc._scaling, pc._rotation, pc._opacity):
            p.data = p.data[perm].contiguous()
        ys = ys[morton_order(ys)].contiguous()
        if lists == "exact":
            idx = exact_knn_lists(pc._xyz.data, d + 1)
            yidx = exact_knn_lists(ys, d + 1)
        else:
            idx = windowed_knn_lists(pc._xyz.data, d + 1)
            yidx = windowed_knn_lists(ys, d + 1, window=min(4096, M))
    else:                                                          # worst case for the gathers: uniformly random lists (seed 22)
        g = torch.Generator().manual_seed(seeds[2])
        idx = torch.randint(0, N, (N, d + 1), generator=g, dtype=torch.int32)
        idx[:, 0] = torch.arange(N, dtype=torch.int32)          # column 0 is the point itself
        idx = idx.to(device)
        yidx = torch.randint(0, M, (M, d + 1), generator=gs, dtype=torch.int32)
        yidx[:, 0] = torch.arange(M, dtype=torch.int32)
        yidx = yidx.to(device)
    mu = None
    if center:
        FY, mu = centred_style_rows(ops.raw_descriptor_fwd(ys, yidx, out_dtype=torch.float32), feat_dtype)
    else:
        FY = ops.raw_descriptor_fwd(ys, yidx, out_dtype=feat_dtype)
    cams = [Camera.look_at_origin(0.6, 0.45, 4.03, 0.6911112070083618, W, H, device=device)]
    gv = torch.Generator().manual_seed(30)                            # further cameras on the radius-4.03 sphere, seed 30
    for _ in range(views - 1):
        th, ph = float(torch.rand(1, generator=gv)) * 2 * math.pi, (float(torch.rand(1, generator=gv)) - 0.3) * 1.2
        cams.append(Camera.look_at_origin(th, ph, 4.03, 0.6911112070083618, W, H, device=device))
    bg = torch.ones(3, device=device)
    loop = GaussianStyleLoop(pc, idx, FY, cams if views > 1 else cams[0], torch.zeros(H, W, device=device), bg, k=k, sampling_seed=seeds[1], timers=timers, feat_center=mu,
                             w_nn=C3_W_NN if w_nn is None else float(w_nn), **kw)
    loop.style_xyz, loop.style_idx = ys, yidx          # kept for the fp32 parity check (bench.py, tests): the unrounded rows can be rebuilt from them
    # depth targets: the initial render's depth of every view, box-blurred (NB13 fits depth to a blurred target)
    e = torch.empty(0, device=device)
    with torch.no_grad():
        check(lib().w3d_activations_fwd(ptr(pc._scaling.data), ptr(pc._rotation.data), ptr(pc._opacity.data), N, ptr(loop.act_s), ptr(loop.act_q), ptr(loop.act_o),
                                        stream_ptr()))
        shs = torch.cat((pc._features_dc.data, pc._features_rest.data), dim=1)
        tg = []
        for cam in cams:
            _, _, depth, *_ = dgr.rasterize_gaussians_fwd(bg, pc._xyz.data, e, loop.act_o, loop.act_s, loop.act_q, 1.0, e, cam.world_view_transform,
                                                          cam.full_proj_transform, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), H, W, shs, 3, cam.camera_center,
                                                          False, False, torch.zeros(H, W, 2, device=device))
            tg.append(torch.nn.functional.avg_pool2d(depth[None, None], 9, 1, 4)[0, 0].contiguous() * 0.95)
        loop.depth_target = tg[0] if views == 1 else torch.stack(tg, 0)
    return loop
