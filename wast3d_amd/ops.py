"""Torch-facing wrappers of the C ABI (include/wast3d_hip.h).

Two layers:
  * raw_* functions: enqueue one kernel group on the current stream, no autograd, caller-owned outputs.
    The hot loops (geometry_loop.GeometryFit, style_loop.GaussianStyleLoop) call these directly so that an
    iteration is a fixed kernel sequence (hipGraph-capturable), not an autograd graph.
  * autograd Functions / modules with the reference's call signatures for drop-in use:
    get_descriptors (NB11:994), get_loss_domain_adaptation (NB11:1011), tv_loss / l2_loss
    (utils/loss_utils.py), FusedAdam (torch.optim.Adam at NB11:1067, scene/gaussian_model.py:163).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import BF16, F16, F32, FP8, W3dError, check, lib, ptr, require_cuda, stream_ptr


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float16:
        return F16
    raise _lib.W3dError(f"unsupported dtype {t.dtype} (float32, bfloat16 or float16)")


def as_i32(idx: torch.Tensor) -> torch.Tensor:
    """Neighbour lists travel as int32 across the ABI (the reference's torch.topk gives int64)."""
    return idx if idx.dtype == torch.int32 else idx.to(torch.int32)


# ------------------------------------------------------------------------------------------------
# kNN
# ------------------------------------------------------------------------------------------------
_ws_cache: dict = {}


def _workspace(nbytes: int, device, ws=None) -> torch.Tensor:
    """Scratch for ONE call.  The process-wide cache (one buffer per device, replaced when a larger request arrives) is for stateless single-call
    users only; anything that carries state from one call to the next — prepare -> match -> repair count of the fused kNN — passes its own
    `ws` (KnnWorkspace), so that two loops on one device cannot read each other's bounds or flags."""
    if ws is not None:
        return ws.tensor(nbytes, device)
    key = (device.index if device.index is not None else torch.cuda.current_device())
    t = _ws_cache.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.zeros(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)      # zero-filled: the kNN's sticky repair word starts at 0
        _ws_cache[key] = t
    return t


class KnnWorkspace:
    """A kNN workspace owned by one loop: row norms, warm-start bounds, split-tail scratch and the repair flags of ITS calls.
    poll()/check(): the repair counters are fetched with an asynchronous copy into pinned memory after every sweep and looked at one
    iteration late — no host synchronisation in the loop, and an overflow of the repair capacity (rows left with fewer than k
    neighbours) is seen for EVERY sweep, not for a sample of them."""

    def __init__(self):
        self.t = None
        self.host = None
        self.ev = None
        self.pending = False
        self.cap = 0
        self.max_redone = 0

    def xnorm_view(self, N: int, M: int, d: int, k: int, dtype_code: int, device) -> torch.Tensor:
        """The workspace's own |x|^2 slot (its first N floats) as a tensor: a featurisation pass that writes the row norms THERE hands them to the kNN calls without a
        copy (w3d_knn_prepare: xnorm == workspace).  Allocates the workspace for this problem size; a later regrow leaves the view valid but detached (the calls then copy)."""
        t = self.tensor(int(lib().w3d_knn_workspace_bytes(N, M, d, k, dtype_code)), device)
        return t[: 4 * N].view(torch.float32)

    def tensor(self, nbytes: int, device) -> torch.Tensor:
        if self.t is None or self.t.numel() < nbytes:
            if self.t is not None and self.pending:
                self.check(wait=True)                      # a poll of the old buffer is in flight: finish it (blocks once, at a regrow) before the buffer is let go
            self.t = torch.zeros(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        return self.t

    def poll(self, N: int, M: int, k: int):
        """Enqueue the copy of {last count, sticky maximum} (call right after a sweep, on its stream)."""
        if self.host is None:
            self.host = torch.zeros(2, dtype=torch.int32).pin_memory()
            self.ev = torch.cuda.Event()
        cap = C.c_int32(0)
        check(lib().w3d_knn_mfma_repair_poll(ptr(self.t), N, M, k, C.c_void_p(self.host.data_ptr()), C.byref(cap), stream_ptr()), "w3d_knn_mfma_repair_poll")
        self.cap = cap.value
        self.ev.record()
        self.pending = True

    def check(self, wait: bool = False):
        """Look at the last completed poll (never blocks unless wait): raises if any sweep overflowed the repair capacity."""
        if not self.pending:
            return
        if wait:
            self.ev.synchronize()
        elif not self.ev.query():
            return
        self.pending = False
        last, worst = int(self.host[0]), int(self.host[1])
        self.max_redone = max(self.max_redone, worst)
        if worst > self.cap:
            raise W3dError(f"kNN warm start: a sweep left {worst} rows with fewer than k neighbours (repair capacity {self.cap}); their lists hold sentinel "
                           "indices and their gradients are NaN")


def knn_set_tuning(mode: int = 0, slack_rel: float = 0.0):
    """Diagnostic knobs of the fused kNN (tests / A/B scripts): see w3d_knn_mfma_set_tuning."""
    check(lib().w3d_knn_mfma_set_tuning(int(mode), float(slack_rel)), "w3d_knn_mfma_set_tuning")


def knn_set_cu_reserve(cus: int = 0):
    """CUs the persistent launch of a sliced pruned sweep leaves free for other streams' kernels (w3d_knn_mfma_set_cu_reserve; process-wide)."""
    check(lib().w3d_knn_mfma_set_cu_reserve(int(cus)), "w3d_knn_mfma_set_cu_reserve")


def raw_knn_match(X: torch.Tensor, Y: torch.Tensor, k: int, idx: torch.Tensor = None, dist2: torch.Tensor = None, prev_idx: torch.Tensor = None, ws=None):
    require_cuda(X, Y)
    assert X.dim() == 2 and Y.dim() == 2 and X.shape[1] == Y.shape[1] and X.dtype == Y.dtype
    X = X.contiguous(); Y = Y.contiguous()
    N, d = X.shape
    M = Y.shape[0]
    if idx is None:
        idx = torch.empty(N, k, dtype=torch.int32, device=X.device)
    if dist2 is None:
        dist2 = torch.empty(N, k, dtype=torch.float32, device=X.device)
    dt = _dtype_code(X)
    nbytes = lib().w3d_knn_workspace_bytes(N, M, d, k, dt)
    ws = _workspace(nbytes, X.device, ws)
    if prev_idx is not None:
        assert prev_idx.shape == (N, k) and prev_idx.dtype == torch.int32 and prev_idx.data_ptr() != idx.data_ptr()
    check(lib().w3d_knn_match_fwd(ptr(X), ptr(Y), N, M, d, k, dt, ptr(prev_idx), ptr(idx), ptr(dist2), ptr(ws), ws.numel(), stream_ptr()), "w3d_knn_match_fwd")
    return idx, dist2


def raw_knn_match_motion(X, Y, k, idx, dist2, prev_dist2, motion2, timed=False, ws=None):
    """kNN seeded by the previous k-th distances + row motion (w3d_knn_prepare_motion).  dist2 may alias prev_dist2 (it is read by
    the prepare step, written by the main kernel).  Returns HIP events around the main kernel if timed."""
    N, d = X.shape; M = Y.shape[0]; dt = _dtype_code(X)
    ws = _workspace(lib().w3d_knn_workspace_bytes(N, M, d, k, dt), X.device, ws)
    check(lib().w3d_knn_prepare_motion(ptr(X), ptr(Y), N, M, d, k, dt, ptr(prev_dist2), ptr(motion2), None, None, ptr(ws), ws.numel(), stream_ptr()), "w3d_knn_prepare_motion")
    ev = None
    if timed:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    check(lib().w3d_knn_match_prepared(ptr(X), ptr(Y), N, M, d, k, dt, 1, ptr(idx), ptr(dist2), None, ptr(ws), ws.numel(), stream_ptr()), "w3d_knn_match_prepared")
    if timed:
        e1.record(); ev = (e0, e1)
    return ev


def raw_knn_prepare(X, Y, k: int, prev_dist2=None, motion2=None, xnorm=None, prev_idx=None, ws=None, row_order=None, style_unchanged: bool = False, motion_ratio: float = 0.0,
                    M: int = None):
    """First half of raw_knn_match / raw_knn_match_motion: row norms (+ a warm-start bound) into the workspace.  Returns warm (0/1).
    prev_dist2 + motion2: the motion bound (no style row touched); xnorm [N]: |x|^2 from the featurisation pass — the norm pass over X is skipped (either bound).
    prev_idx: the gathered bound = the previous neighbours' CURRENT distances (k gathers of style rows per content row); it may be the very tensor the
    main kernel writes afterwards (the prepare phase has finished reading it).
    xnorm written through KnnWorkspace.xnorm_view sits in the workspace already and is not even copied.
    style_unchanged (gathered / no bound): Y holds the rows of the previous prepare on this workspace — their norms are kept.
    Y = None with M (motion bound only): the style norms in the workspace are somebody else's business (w3d_knn_chunk_norms) and are left alone."""
    N, d = X.shape; M = Y.shape[0] if Y is not None else int(M); dt = _dtype_code(X)
    ws = _workspace(lib().w3d_knn_workspace_bytes(N, M, d, k, dt), X.device, ws)
    if prev_idx is not None and prev_dist2 is not None and motion2 is not None and motion_ratio > 0.0:
        # gathered bound, except for rows that moved less than motion_ratio x their previous k-th distance: those take the motion bound and fetch nothing
        check(lib().w3d_knn_prepare_hybrid(ptr(X), ptr(Y), N, M, d, k, dt, ptr(prev_idx), ptr(prev_dist2), ptr(motion2), float(motion_ratio),
                                           ptr(xnorm) if xnorm is not None else None, int(bool(style_unchanged)), ptr(row_order), ptr(ws), ws.numel(), stream_ptr()), "w3d_knn_prepare_hybrid")
        return 1
    if prev_dist2 is not None and prev_idx is None:
        check(lib().w3d_knn_prepare_motion(ptr(X), ptr(Y), N, M, d, k, dt, ptr(prev_dist2), ptr(motion2), ptr(xnorm) if xnorm is not None else None,
                                           ptr(row_order), ptr(ws), ws.numel(), stream_ptr()), "w3d_knn_prepare_motion")
        return 1
    check(lib().w3d_knn_prepare(ptr(X), ptr(Y), N, M, d, k, dt, ptr(prev_idx) if prev_idx is not None else None, ptr(xnorm) if xnorm is not None else None,
                                int(bool(style_unchanged)), ptr(row_order), ptr(ws), ws.numel(), stream_ptr()),
          "w3d_knn_prepare")
    return 0 if prev_idx is None else 1


def raw_knn_match_prepared(X, Y, k: int, idx, dist2, warm: int, timed: bool = False, ws=None, row_order=None, bounds: "KnnBounds" = None, prune: "KnnPrune" = None):
    """Second half: the fused distance-GEMM + top-k kernel on the prepared workspace; HIP events around it if timed (with bounds: four events — the launch group's
    two, then two around the emit kernel alone).
    warm: 0 cold, 1 bound prepared, 3 pruned sweep (plan in the workspace), 4 pruned sweep with EMIT selection (tight bounds only: the gathered one).
    bounds (with warm = 4 and the plan's `prune`): the sweep records the cached lower bounds of every (row block, tile) pair it sweeps."""
    N, d = X.shape; M = Y.shape[0]; dt = _dtype_code(X)
    ws = _workspace(lib().w3d_knn_workspace_bytes(N, M, d, k, dt), X.device, ws)
    ev = None
    if timed:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    if bounds is not None:
        assert warm == 4 and prune is not None and bounds.fits(N, M)
        strag = 0x100 if getattr(bounds, "stragglers", False) else 0      # the plan took straggler pairs out of its tile lists: this sweep computes them (and only this one)
        bounds.stragglers = False
        if timed:
            # the emit kernel's own duration inside the launch group (reset, emit, stragglers, select): two more events, recorded by the library around that kernel
            k0 = torch.cuda.Event(enable_timing=True); k1 = torch.cuda.Event(enable_timing=True)
            k0.record(); k1.record()                                      # (creates the hipEvent_t handles; re-recorded below)
            check(lib().w3d_knn_mfma_debug_events(k0.cuda_event, k1.cuda_event), "w3d_knn_mfma_debug_events")
        try:
            check(lib().w3d_knn_match_prepared_lb(ptr(X), ptr(Y), N, M, d, k, dt, 4 | strag, ptr(idx), ptr(dist2), ptr(row_order), ptr(bounds.lb), ptr(bounds.cum),
                                                  prune.cnorm_max, ptr(ws), ws.numel(), stream_ptr()), "w3d_knn_match_prepared_lb")
        finally:
            if timed:
                lib().w3d_knn_mfma_debug_events(None, None)
        if timed:
            e1.record()
            return (e0, e1, k0, k1)
        return None
    check(lib().w3d_knn_match_prepared(ptr(X), ptr(Y), N, M, d, k, dt, int(warm), ptr(idx), ptr(dist2), ptr(row_order), ptr(ws), ws.numel(), stream_ptr()), "w3d_knn_match_prepared")
    if timed:
        e1.record(); ev = (e0, e1)
    return ev


def cache_budget_ok(nbytes: int, budget: int, device) -> bool:
    """May a pure performance cache of `nbytes` be allocated?  budget > 0: an explicit cap in bytes; budget == 0: at most half of the device memory that is free
    right now (torch.cuda.mem_get_info + what torch's caching allocator holds unused); budget < 0: never.  The caches this guards (KnnBounds: 2 GB at C3, 32 GB at M = 1e6; NeighbourSumCache: 1 GB at C3) change
    speed only — a loop that cannot afford them runs the plain pruned sweep / the gathering backward."""
    if budget < 0:
        return False
    if budget > 0:
        return nbytes <= budget
    try:
        free, _total = torch.cuda.mem_get_info(device)
        #  memory torch's caching allocator holds but is not using is as good as free for a torch allocation: a long-lived process with a large reserved
        # pool used to be refused although the allocation would have succeeded
        free += max(0, torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device))
    except Exception:
        return True
    return nbytes <= free // 2


class KnnPrune:
    """Tile geometry of a style set for PRUNED sweeps (w3d_knn_prune_plan): per 64-row tile its mean row rounded to the feature format (the centre), |c|^2 and the
    radius max |y - c| over the tile's rows, rounded up.  Built once per style set (it is constant in the loop)."""

    def __init__(self, FY: torch.Tensor, meb_iters: int = 100, fp8: bool = False):
        # fp8: the plan is for sweeps of e4m3 rows (FY = their dequantised values): the allowance for the sweep's score error is the fp8 MFMA's (w3d_knn_prune_plan_err)
        self.score_err2 = 4e-5 if fp8 else 0.0                             # W3D_KNN_SCORE_ERR2_FP8 (include/wast3d_hip.h)
        M, d = FY.shape
        T = (M + 63) // 64
        Yf = FY.float()
        pad = T * 64 - M
        if pad:
            Yf = torch.cat([Yf, Yf[-1:].expand(pad, d)], 0)               # the last tile is completed with copies of its last row (centre and radius stay those of real rows)
        Yt = Yf.view(T, 64, d)
        # centre = (approximately) that of the tile's minimum enclosing ball — Badoiu-Clarkson steps from the mean: c += (farthest row - c) / (i + 2).  Any point is a
        # valid centre (the radius is measured from it); this one gives radii ~15 % below the mean's and 5 points more skipped tiles at C3 (scripts/prune_probe.py)
        c = Yt.mean(1)
        ar = torch.arange(T, device=FY.device)
        for i in range(meb_iters):
            far = (Yt - c[:, None]).norm(dim=2).argmax(1)
            c = c + (Yt[ar, far] - c) / (i + 2)
        self.centres = c.to(FY.dtype).contiguous()
        c = self.centres.float()
        self.cnorm = (c * c).sum(1).contiguous()
        self.cradius = ((Yt - c[:, None]).norm(dim=2).max(1).values * (1.0 + 1e-6) + 1e-30).contiguous()
        # scale of the margin for the sweep's own rounding (w3d_knn_prune_plan): the largest |y|^2 of the set is below (|c| + r)^2 <= 2 (|c|^2 + r^2)
        self.cnorm_max = float((2.0 * (self.cnorm + self.cradius * self.cradius)).max().item())
        self.M, self.d, self.of = M, d, FY
        self.key = (FY.data_ptr(), FY._version, tuple(FY.shape))      # the rows these tiles were measured on (a loop rebuilds the plan when its style rows' key changes)


class KnnBounds:
    """Cached lower bounds of a loop's pruned sweeps (include/wast3d_hip.h "CACHED LOWER BOUNDS"): lb [row blocks][tiles][512] f16 — per (block position of the sweep
    order, style tile, row of the block) the row's exact minimum distance to the tile when the block last swept it + the row's cumulative motion then; cum [N] f32.
    Keyed by POSITION: reset() whenever the row order, the style rows or N change (the loop does)."""

    def __init__(self, N: int, M: int, device, summaries: bool = True):
        self.N, self.M = N, M
        self.nblk, self.T = (N + 511) // 512, (M + 63) // 64
        self.lb = torch.full((self.nblk, self.T, 512), float("-inf"), dtype=torch.float16, device=device)
        self.cum = torch.zeros(N, dtype=torch.float32, device=device)
        # block summaries (the smallest bound any row of the block holds for the tile; +inf = none yet): what lets the STEADY plans — all but the first after a reset —
        # drop most tiles without their per-row bounds and without the ball test's pass over the centres.  None: every plan runs the ball test (A/B)
        # (-inf = unknown.)  The summaries of tiles never swept are BALL bounds aged by the rows' motion (worst case: straight at the tile).  `ball_every` > 0 runs the ball
        # test again every so many plans and raises them to the fresh ball bounds — measured at 8 (C3, profiles/r04k_summaries.log): swept fraction 0.1449 -> 0.1416, but the
        # plan averages 0.97 instead of 0.76 ms in the loop: 9.21 -> 9.46 ms per step, 92.5 -> 90.1 it/s over 1001 iterations.  Off.
        self.summ = torch.full((self.nblk, self.T, 8), float("-inf"), dtype=torch.float32, device=device) if summaries else None      # per 64 rows of the block
        self.dref = torch.zeros(N, dtype=torch.float32, device=device) if summaries else None       # the rows' cum + D at the last ball-test plan: the summaries are relative to it
        self.ball_every = 0
        self.since_ball = 0
        self.fresh = True                          # no plan since the last reset: the next one runs the ball test and writes the ball bounds' summaries
        self.rebase = False                        # set by reorder(): the next (steady) plan moves the rows' references
        self.folded = False                        # every (row, tile) holds a bound of its own (true after the first reorder since a reset)
        self.stragglers = False                    # the last plan took straggler pairs out of its tile lists (raw_knn_prune_plan(straggler_rows=...)): the next sweep runs them
        self.resets = 1

    def fits(self, N: int, M: int) -> bool:
        return (self.N, self.M) == (N, M)

    @staticmethod
    def bytes_needed(N: int, M: int, summaries: bool = True, carry: bool = False) -> int:
        """Device bytes a KnnBounds(N, M) takes: lb f16 [row blocks][tiles][512] (twice with bounds_carry), cum (+ dref) f32 [N], summaries f32 [row blocks][tiles][8]."""
        nblk, T = (N + 511) // 512, (M + 63) // 64
        return nblk * T * 512 * 2 * (2 if carry else 1) + N * 4 * (2 if summaries else 1) + (nblk * T * 8 * 4 if summaries else 0)

    def reset(self):
        self.lb.fill_(float("-inf")); self.cum.zero_(); self.resets += 1
        if self.summ is not None:
            self.summ.fill_(float("-inf"))
        self.fresh = True; self.rebase = False; self.folded = False

    def reorder(self, old_order: Optional[torch.Tensor], new_order: torch.Tensor):
        """The bounds follow their rows from the sweep order `old_order` (None = storage order) into `new_order` (int32 [N] row ids by position) instead of being reset
        (w3d_knn_bounds_reorder): no sweep of everything the ball test keeps afterwards.  Needs a second lb array (allocated at the first call)."""
        if self.fresh:                               # nothing recorded yet
            return self.reset()
        N, dev = self.N, self.lb.device
        ar = torch.arange(N, device=dev, dtype=torch.int32)
        inv_old = ar if old_order is None else torch.empty(N, dtype=torch.int32, device=dev).scatter_(0, old_order.long(), ar)
        src = inv_old[new_order.long()]
        pad = self.nblk * 512 - N
        if pad:
            src = torch.cat([src, src[-1:].expand(pad)])
        src = src.contiguous()
        if getattr(self, "_lb2", None) is None:
            self._lb2 = torch.empty_like(self.lb)
        # the summaries are folded into the per-row bounds by the FIRST reorder after a reset only: from then on every (row, tile) has a bound of its own, which is
        # never below what its sub-block's summary says (the plans form the summaries as minima of exactly these)
        fold = self.summ is not None and not self.folded
        check(lib().w3d_knn_bounds_reorder(ptr(self.lb), ptr(self._lb2), ptr(self.summ) if fold else None, ptr(self.dref) if fold else None, ptr(self.cum), ptr(src), ptr(new_order),
                                           N, self.M, stream_ptr()), "w3d_knn_bounds_reorder")
        self.folded = True
        self.lb, self._lb2 = self._lb2, self.lb
        self.cum.zero_()
        if self.summ is not None:
            self.summ.fill_(float("-inf"))
            self.rebase = True                       # the next plan: steady, and it moves the rows' references
        self.reorders = getattr(self, "reorders", 0) + 1

    def advance(self, motion2: torch.Tensor):
        """cum += |x_new - x_old| of every row (motion2 [N] = squared motion since the previous sweep)."""
        assert motion2.shape == (self.N,) and motion2.dtype == torch.float32
        check(lib().w3d_knn_bounds_advance(ptr(motion2), self.N, ptr(self.cum), stream_ptr()), "w3d_knn_bounds_advance")


def knn_plan_is_steady(bounds: "KnnBounds") -> bool:
    """Will the next raw_knn_prune_plan(bounds=bounds) be a STEADY plan (summaries + per-row bounds only: the content rows are not read)?"""
    return bounds is not None and bounds.summ is not None and not bounds.fresh and (bounds.ball_every <= 0 or bounds.since_ball + 1 < bounds.ball_every)


def raw_knn_prune_plan(X, k: int, prune: "KnnPrune", ws, row_order=None, bounds: "KnnBounds" = None, shape=None, straggler_rows: int = 0):
    """Tile plan of a pruned sweep from the bounds raw_knn_prepare(prev_idx=...) has just left in `ws`; follow with raw_knn_match_prepared(warm=3 or 4).
    bounds: tiles the ball test keeps are dropped when the cached lower bounds still rule them out for every row of the block (advance() them first).
    X = None with shape = (N, d, dtype code): only for a steady plan (knn_plan_is_steady), which does not read the rows — the fp8 loop's rows would have to be
    dequantised for nothing.
    straggler_rows (with bounds that carry summaries, 16-bit rows): a kept tile that at most this many rows of a block still need leaves the block's list; the sweep
    (raw_knn_match_prepared with the same bounds) computes those (row, tile) pairs on the side.  bounds.stragglers says whether the plan was built that way."""
    if X is None:
        assert knn_plan_is_steady(bounds) and shape is not None
        N, d, dt = shape
        dev = bounds.lb.device
    else:
        N, d = X.shape; dt = _dtype_code(X); dev = X.device
    w = _workspace(lib().w3d_knn_workspace_bytes(N, prune.M, d, k, dt), dev, ws)
    if bounds is not None:
        assert bounds.fits(N, prune.M)
        steady = knn_plan_is_steady(bounds)
        bounds.since_ball = bounds.since_ball + 1 if steady else 0
        T = min(int(straggler_rows), 255) if (straggler_rows and bounds.summ is not None and dt in (BF16, F16)) else 0
        bounds.stragglers = T > 0
        check(lib().w3d_knn_prune_plan_err(ptr(X) if X is not None else ptr(bounds.lb), N, prune.M, d, k, dt, ptr(row_order), ptr(prune.centres), ptr(prune.cnorm), ptr(prune.cradius), prune.cnorm_max,
                                           prune.score_err2, ptr(bounds.lb), ptr(bounds.cum), ptr(bounds.summ), ptr(bounds.dref), ((2 if bounds.rebase else 1) if steady else 0) | (T << 8),
                                           ptr(w), w.numel(), stream_ptr()), "w3d_knn_prune_plan_err")
        bounds.fresh = False; bounds.rebase = False
        if not steady:
            bounds.folded = False                    # a ball-test plan raises summaries above what the rows' own bounds say
        return
    check(lib().w3d_knn_prune_plan_err(ptr(X), N, prune.M, d, k, dt, ptr(row_order), ptr(prune.centres), ptr(prune.cnorm), ptr(prune.cradius), prune.cnorm_max, prune.score_err2, None, None,
                                       None, None, 0, ptr(w), w.numel(), stream_ptr()), "w3d_knn_prune_plan_err")


def raw_knn_plan_nearest_first(X, k: int, prune: "KnnPrune", ws, row_order=None):
    """Plan of a COLD sweep (no bound yet: a loop's first): every block of 512 rows (in row_order) takes all tiles, those nearest to the block's mean row first — the
    running k-th best then collapses within the first few tiles instead of falling slowly along the stored order of the style rows.  Call after raw_knn_prepare(X, Y, k)
    (norms), follow with raw_knn_match_prepared(warm=3).  Result = the cold sweep's."""
    N, d = X.shape; dt = _dtype_code(X)
    w = _workspace(lib().w3d_knn_workspace_bytes(N, prune.M, d, k, dt), X.device, ws)
    nblk = (N + 511) // 512
    Xo = X if row_order is None else X[row_order.long()]
    pad = nblk * 512 - N
    Xf = Xo.float()
    if pad:
        Xf = torch.cat([Xf, Xf[-1:].expand(pad, d)], 0)
    mean = Xf.view(nblk, 512, d).mean(1)
    score = mean @ prune.centres.float().T - 0.5 * prune.cnorm[None, :]         # largest = nearest centre
    order = torch.argsort(score, dim=1, descending=True).to(torch.int32).contiguous()
    check(lib().w3d_knn_plan_order(N, prune.M, d, k, ptr(order), 1, ptr(w), w.numel(), stream_ptr()), "w3d_knn_plan_order")


def knn_prune_counts(N: int, M: int, d: int, k: int, ws) -> torch.Tensor:
    """The plan's tiles-to-sweep per block of 512 rows (int32 view into the workspace): statistics."""
    w = ws.t if isinstance(ws, KnnWorkspace) else ws
    off = int(lib().w3d_knn_prune_counts_offset(N, M, d, k))
    nblk = (N + 511) // 512
    return w[off:off + 4 * nblk].view(torch.int32)


def knn_straggler_counts(N: int, M: int, d: int, k: int, ws) -> torch.Tensor:
    """(row, tile) pairs per style tile the last plan with straggler_rows took out of the blocks' tile lists (int32 view into the workspace; a count above
    lib().w3d_knn_straggler_capacity(N, M): that bucket overflowed, the pairs past it stayed with their blocks): statistics."""
    w = ws.t if isinstance(ws, KnnWorkspace) else ws
    off = int(lib().w3d_knn_straggler_counts_offset(N, M, d, k))
    return w[off:off + 4 * ((M + 63) // 64)].view(torch.int32)


def raw_knn_prepare_chunked(X, M: int, k: int, prev_dist2=None, motion2=None, xnorm=None, ws=None):
    """Prepare for a CHUNKED sweep (style set arriving shard by shard): content norms + bound only, no style row is touched.  Returns warm (0/1)."""
    N, d = X.shape; dt = _dtype_code(X)
    w = _workspace(lib().w3d_knn_workspace_bytes(N, M, d, k, dt), X.device, ws)
    if prev_dist2 is not None:
        check(lib().w3d_knn_prepare_motion(ptr(X), None, N, M, d, k, dt, ptr(prev_dist2), ptr(motion2), ptr(xnorm) if xnorm is not None else None, None, ptr(w), w.numel(),
                                           stream_ptr()), "w3d_knn_prepare_motion(chunked)")
        return 1
    check(lib().w3d_knn_prepare(ptr(X), None, N, M, d, k, dt, None, None, 0, None, ptr(w), w.numel(), stream_ptr()), "w3d_knn_prepare(chunked)")
    return 0


def raw_knn_chunk(X, Y, k: int, warm: int, m0: int, m_valid: int, m1: int, part: int, parts: int, part_idx, part_val, ws=None, row_order=None):
    """-0.5|y|^2 of style rows [m0, m_valid) (+ padding up to m1) and the sweep of all content rows against those rows; partial lists into slot `part`."""
    N, d = X.shape; M = Y.shape[0]; dt = _dtype_code(X)
    w = _workspace(lib().w3d_knn_workspace_bytes(N, M, d, k, dt), X.device, ws)
    check(lib().w3d_knn_chunk_norms(ptr(Y), N, M, d, k, dt, m0, m_valid, m1, ptr(w), w.numel(), stream_ptr()), "w3d_knn_chunk_norms")
    check(lib().w3d_knn_chunk_sweep(ptr(X), ptr(Y), N, M, d, k, dt, int(warm), m0, m1, part, parts, ptr(part_idx), ptr(part_val), ptr(row_order), ptr(w), w.numel(), stream_ptr()),
          "w3d_knn_chunk_sweep")


def raw_knn_chunk_norms(X, Y, k: int, m0: int, m_valid: int, m1: int, ws=None):
    """-0.5|y|^2 of style rows [m0, m_valid); rows [m_valid, m1) are padding (-inf)."""
    N, d = X.shape; M = Y.shape[0]; dt = _dtype_code(X)
    w = _workspace(lib().w3d_knn_workspace_bytes(N, M, d, k, dt), X.device, ws)
    check(lib().w3d_knn_chunk_norms(ptr(Y), N, M, d, k, dt, m0, m_valid, m1, ptr(w), w.numel(), stream_ptr()), "w3d_knn_chunk_norms")


def raw_knn_chunk_sweep(X, Y, k: int, warm: int, m0: int, m1: int, part: int, parts: int, part_idx, part_val, ws=None, row_order=None):
    """All content rows against style rows [m0, m1) (norms in place); partial lists into slot `part`."""
    N, d = X.shape; M = Y.shape[0]; dt = _dtype_code(X)
    w = _workspace(lib().w3d_knn_workspace_bytes(N, M, d, k, dt), X.device, ws)
    check(lib().w3d_knn_chunk_sweep(ptr(X), ptr(Y), N, M, d, k, dt, int(warm), m0, m1, part, parts, ptr(part_idx), ptr(part_val), ptr(row_order), ptr(w), w.numel(), stream_ptr()),
          "w3d_knn_chunk_sweep")


def raw_knn_chunk_merge(X, Y, k: int, warm: int, parts: int, part_idx, part_val, idx, dist2, ws=None, row_order=None):
    N, d = X.shape; M = Y.shape[0]; dt = _dtype_code(X)
    w = _workspace(lib().w3d_knn_workspace_bytes(N, M, d, k, dt), X.device, ws)
    check(lib().w3d_knn_chunk_merge(ptr(X), ptr(Y), N, M, d, k, dt, int(warm), parts, ptr(part_idx), ptr(part_val), ptr(idx), ptr(dist2), ptr(row_order), ptr(w), w.numel(), stream_ptr()),
          "w3d_knn_chunk_merge")


def knn_repair_count(X, Y, k: int, ws=None):
    """(rows the last warm-started kNN call on these shapes redid with a cold bound, capacity of that repair).  Blocks."""
    N, d = X.shape; M = Y.shape[0]; dt = FP8 if X.dtype == torch.uint8 else _dtype_code(X)
    ws = _workspace(lib().w3d_knn_workspace_bytes(N, M, d, k, dt), X.device, ws)
    c, cap = C.c_int32(0), C.c_int32(0)
    check(lib().w3d_knn_mfma_repair_count(ptr(ws), N, M, k, C.byref(c), C.byref(cap), stream_ptr()), "w3d_knn_mfma_repair_count")
    return c.value, cap.value


def raw_knn_fp8_phase(qX, sX, qY, sY, k: int, phase: int, warm: int, idx=None, dist2=None, prev_dist2=None, motion2=None, timed: bool = False, prev_idx=None, ws=None,
                      row_order=None, bounds: "KnnBounds" = None, prune: "KnnPrune" = None):
    """w3d_knn_fp8 one phase at a time (1 = prepare, 2 = main kernel).  phase 2 with warm = 4: the pruned sweep with the EMIT selection (w3d_knn_fp8_lb; + the cached
    lower bounds when `bounds` and the plan's `prune` are given)."""
    N, d = qX.shape; M = qY.shape[0]
    ws = _workspace(lib().w3d_knn_workspace_bytes(N, M, d, k, FP8), qX.device, ws)
    ev = None
    if timed:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    if phase == 2 and int(warm) == 4:
        assert bounds is None or (prune is not None and bounds.fits(N, M))
        check(lib().w3d_knn_fp8_lb(ptr(qX), ptr(sX), ptr(qY), ptr(sY), N, M, d, k, 4, ptr(idx), ptr(dist2), ptr(row_order), ptr(bounds.lb) if bounds is not None else None,
                                   ptr(bounds.cum) if bounds is not None else None, prune.cnorm_max if bounds is not None else 0.0, ptr(ws), ws.numel(), stream_ptr()), "w3d_knn_fp8_lb")
        if timed:
            e1.record(); ev = (e0, e1)
        return ev
    check(lib().w3d_knn_fp8(ptr(qX), ptr(sX), ptr(qY), ptr(sY), N, M, d, k, phase, int(warm), ptr(prev_idx) if prev_idx is not None else None, ptr(prev_dist2), ptr(motion2), ptr(idx), ptr(dist2), ptr(row_order), ptr(ws), ws.numel(),
                            stream_ptr()), "w3d_knn_fp8")
    if timed:
        e1.record(); ev = (e0, e1)
    return ev


def raw_descriptor_fwd_window(X4, idx, F_inout, motion2=None, xnorm=None, window: int = 0, center=None, slots: "SlotPlan" = None):
    """Featurisation with the positions of each block's window staged in LDS (16-bit rows; motion2 / xnorm optional).
    center [d] f32: subtracted from every row before the 16-bit rounding (the style rows must carry the same shift).
    slots: the backward's SlotPlan — the window of a row block is then the set of points its rows name (every neighbour fetch an LDS read)."""
    R, K = idx.shape
    if slots is not None:
        assert slots.R == R and slots.K == K and slots.n_points == X4.shape[0]
        check(lib().w3d_descriptor_fwd_slots_x4(ptr(X4), X4.shape[0], ptr(idx), R, K, _dtype_code(F_inout), ptr(F_inout), ptr(motion2) if motion2 is not None else None,
                                                ptr(xnorm) if xnorm is not None else None, ptr(center), slots.rows_per_block, max(slots.max_set, 1), ptr(slots.nbr_slot),
                                                ptr(slots.dest), ptr(slots.dest_off), ptr(slots.root) if slots.root is not None else None, stream_ptr()),
              "w3d_descriptor_fwd_slots_x4")
        return
    check(lib().w3d_descriptor_fwd_window_x4(ptr(X4), X4.shape[0], ptr(idx), R, K, _dtype_code(F_inout), ptr(F_inout), ptr(motion2) if motion2 is not None else None,
                                             ptr(xnorm) if xnorm is not None else None, window, ptr(center), stream_ptr()), "w3d_descriptor_fwd_window_x4")


def raw_descriptor_fwd_x4_motion(X4, idx, F_inout, motion2, xnorm=None, center=None):
    R, K = idx.shape
    check(lib().w3d_descriptor_fwd_x4_motion(ptr(X4), ptr(idx), R, K, _dtype_code(F_inout), ptr(F_inout), ptr(motion2), ptr(xnorm) if xnorm is not None else None,
                                             ptr(center), stream_ptr()), "w3d_descriptor_fwd_x4_motion")


def raw_knn_match_timed(X, Y, k, idx, dist2, prev_idx=None):
    """raw_knn_match as prepare + main kernel, with HIP events (on the launch stream) around the main kernel only.
    Returns (event_start, event_end)."""
    N, d = X.shape; M = Y.shape[0]; dt = _dtype_code(X)
    ws = _workspace(lib().w3d_knn_workspace_bytes(N, M, d, k, dt), X.device)
    check(lib().w3d_knn_prepare(ptr(X), ptr(Y), N, M, d, k, dt, ptr(prev_idx), None, 0, None, ptr(ws), ws.numel(), stream_ptr()), "w3d_knn_prepare")
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib().w3d_knn_match_prepared(ptr(X), ptr(Y), N, M, d, k, dt, int(prev_idx is not None), ptr(idx), ptr(dist2), None, ptr(ws), ws.numel(), stream_ptr()),
          "w3d_knn_match_prepared")
    e1.record()
    return e0, e1


# ---- fp8 distance-GEMM (BASELINE config C5): e4m3 rows + one power-of-two scale per row, scaled MFMA at twice the bf16 rate
def quantize_rows_fp8(F: torch.Tensor, q: torch.Tensor = None, scale: torch.Tensor = None, motion2: torch.Tensor = None):
    """F [R,d] (f32 / bf16 / f16) -> (q [R,d] uint8 = OCP e4m3 bytes, scale [R] int32 = E8M0 exponent byte; row = q * 2^(scale-127)).
    With motion2 [R] given, q/scale must hold the previous iteration's values: motion2 = |x^_new - x^_old|^2 of the dequantised rows."""
    require_cuda(F)
    Fc = F.contiguous()
    R, d = Fc.shape
    if q is None:
        q = torch.empty(R, d, dtype=torch.uint8, device=F.device)
    if scale is None:
        scale = torch.empty(R, dtype=torch.int32, device=F.device)
    check(lib().w3d_quantize_rows_fp8(ptr(Fc), _dtype_code(Fc), R, d, ptr(q), ptr(scale), ptr(motion2) if motion2 is not None else None, stream_ptr()),
          "w3d_quantize_rows_fp8")
    return q, scale


def dequantize_rows_fp8(q: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """The f32 rows a (q, scale) pair stands for (checker / debugging helper; torch's float8_e4m3fn is the same OCP format)."""
    return q.view(torch.float8_e4m3fn).float() * torch.exp2((scale.float() - 127.0))[:, None]


def raw_knn_match_fp8(qX, sX, qY, sY, k: int, idx=None, dist2=None, prev_idx=None, prev_dist2=None, motion2=None, timed=False):
    """Exact kNN of the dequantised rows on the scaled fp8 MFMA.  Warm start from prev_idx (any candidates) or from the previous
    k-th distances + row motion (prev_dist2 may alias dist2)."""
    require_cuda(qX, qY, sX, sY)
    N, d = qX.shape; M = qY.shape[0]
    if idx is None:
        idx = torch.empty(N, k, dtype=torch.int32, device=qX.device)
    if dist2 is None:
        dist2 = torch.empty(N, k, dtype=torch.float32, device=qX.device)
    ws = _workspace(lib().w3d_knn_workspace_bytes(N, M, d, k, FP8), qX.device)
    warm = int(prev_idx is not None or prev_dist2 is not None)
    args = (ptr(qX), ptr(sX), ptr(qY), ptr(sY), N, M, d, k)
    tail = (ptr(prev_idx), ptr(prev_dist2), ptr(motion2), ptr(idx), ptr(dist2), None, ptr(ws), ws.numel(), stream_ptr())
    check(lib().w3d_knn_fp8(*args, 1, warm, *tail), "w3d_knn_fp8(prepare)")
    ev = None
    if timed:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    check(lib().w3d_knn_fp8(*args, 2, warm, *tail), "w3d_knn_fp8(match)")
    if timed:
        e1.record(); ev = (e0, e1)
    return (idx, dist2, ev) if timed else (idx, dist2)


def raw_knn_loss(dist2: torch.Tensor, scale: float, loss_accum: torch.Tensor):
    N, k = dist2.shape
    check(lib().w3d_knn_loss(ptr(dist2), N, k, scale, ptr(loss_accum), stream_ptr()), "w3d_knn_loss")


def raw_knn_match_bwd(X, Y, idx, scale: float, grad_X: torch.Tensor, accumulate: bool, row_order=None, generic: bool = False):
    """dL/dF_X of the NN loss.  16-bit rows of 256 elements take the four-rows-per-wave kernel (optionally in `row_order`, the sweep's processing order);
    generic=True forces the any-shape kernel (tests)."""
    N, d = X.shape
    if d == 256 and X.dtype in (torch.bfloat16, torch.float16) and not generic:
        check(lib().w3d_knn_match_bwd_ordered(ptr(X), ptr(Y), ptr(idx), N, Y.shape[0], d, idx.shape[1], _dtype_code(X), scale, int(accumulate),
                                              ptr(row_order) if row_order is not None else None, ptr(grad_X), stream_ptr()), "w3d_knn_match_bwd_ordered")
        return
    check(lib().w3d_knn_match_bwd(ptr(X), ptr(Y), ptr(idx), N, Y.shape[0], d, idx.shape[1], _dtype_code(X), scale, int(accumulate), ptr(grad_X),
                                  stream_ptr()), "w3d_knn_match_bwd")


class _KnnMatchLoss(torch.autograd.Function):
    """loss = weight * mean over all N*M pairs of D^2 * mask(k nearest)  (NB11:1011-1023 on feature rows)."""

    @staticmethod
    def forward(ctx, FX, FY, k, weight):
        FXc = FX.contiguous(); FYc = FY.contiguous()
        idx, d2 = raw_knn_match(FXc, FYc, k)
        loss = torch.zeros(1, dtype=torch.float32, device=FX.device)
        scale = float(weight) / (FX.shape[0] * FY.shape[0])
        raw_knn_loss(d2, scale, loss)
        ctx.save_for_backward(FXc, FYc, idx)
        ctx.scale = scale
        ctx.mark_non_differentiable(idx)
        return loss[0], idx

    @staticmethod
    def backward(ctx, g_loss, _g_idx):
        FX, FY, idx = ctx.saved_tensors
        g = torch.empty(FX.shape, dtype=torch.float32, device=FX.device)
        raw_knn_match_bwd(FX, FY, idx, ctx.scale, g, False)
        g = g * g_loss
        return g.to(FX.dtype), None, None, None


def knn_match_loss(FX, FY, k: int = 20, weight: float = 1.0):
    return _KnnMatchLoss.apply(FX, FY, k, weight)


# ------------------------------------------------------------------------------------------------
# descriptors
# ------------------------------------------------------------------------------------------------
def raw_descriptor_fwd(X, idx, out=None, out_dtype=torch.float32):
    require_cuda(X, idx)
    R, K = idx.shape
    if out is None:
        out = torch.empty(R, K - 1, dtype=out_dtype, device=X.device)
    if R == 0:
        return out                                   # empty index tensor (torch indexing accepts it): nothing to launch
    check(lib().w3d_descriptor_fwd(ptr(X), ptr(idx), R, K, _dtype_code(out), ptr(out), stream_ptr()), "w3d_descriptor_fwd")
    return out


def raw_pack_xyz4(X, X4=None):
    n = X.shape[0]
    if X4 is None:
        X4 = torch.empty(n, 4, dtype=torch.float32, device=X.device)
    check(lib().w3d_pack_xyz4(ptr(X), n, ptr(X4), stream_ptr()), "w3d_pack_xyz4")
    return X4


def raw_descriptor_fwd_x4(X4, idx, out, center=None):
    R, K = idx.shape
    check(lib().w3d_descriptor_fwd_x4(ptr(X4), ptr(idx), R, K, _dtype_code(out), ptr(out), ptr(center), stream_ptr()), "w3d_descriptor_fwd_x4")
    return out


def raw_descriptor_bwd(X, idx, gF, gX):
    R, K = idx.shape
    if R == 0:
        return
    check(lib().w3d_descriptor_bwd(ptr(X), ptr(idx), R, K, ptr(gF), ptr(gX), stream_ptr()), "w3d_descriptor_bwd")


# ------------------------------------------------------------------------------------------------
# setup stage on point clouds: large-K neighbour lists, radius counts, distance rows
# ------------------------------------------------------------------------------------------------
def knn_lists(X: torch.Tensor, K: int, Y: torch.Tensor = None, return_dist: bool = False):
    """torch.cdist + torch.topk(largest=False) with large K (NB11:1055-1063) -> idx [N,K] int32 (and the distances).
    Y = None: neighbours within X itself, every row led by its own point (column 0 = the root)."""
    require_cuda(X)
    Xc = X.contiguous().float()
    self_first = Y is None
    Yc = Xc if self_first else Y.contiguous().float()
    N, d = Xc.shape
    M = Yc.shape[0]
    idx = torch.empty(N, K, dtype=torch.int32, device=Xc.device)
    dist = torch.empty(N, K, dtype=torch.float32, device=Xc.device) if return_dist else None
    ws = _workspace(lib().w3d_knn_lists_workspace_bytes(M, d), Xc.device)
    check(lib().w3d_knn_lists(ptr(Xc), ptr(Yc), N, M, d, K, int(self_first), ptr(idx), ptr(dist) if return_dist else None, ptr(ws), ws.numel(), stream_ptr()),
          "w3d_knn_lists")
    return (idx, dist) if return_dist else idx


def radius_count(X: torch.Tensor, Y: torch.Tensor, radius: float, y_stride: int = 1) -> torch.Tensor:
    """(cdist(X, Y[::y_stride]) <= radius).sum(1) without the matrix (NB11:119-152)."""
    require_cuda(X, Y)
    Xc, Yc = X.contiguous().float(), Y.contiguous().float()
    M = (Yc.shape[0] + y_stride - 1) // y_stride
    out = torch.empty(Xc.shape[0], dtype=torch.int32, device=Xc.device)
    check(lib().w3d_radius_count(ptr(Xc), ptr(Yc), Xc.shape[0], M, Xc.shape[1], y_stride, float(radius), ptr(out), stream_ptr()), "w3d_radius_count")
    return out


def dist_row(q: torch.Tensor, Y: torch.Tensor) -> torch.Tensor:
    """cdist(q[None], Y)[0]: one row of the distance matrix (NB11:761-785 asks for one row per greedy step)."""
    require_cuda(q, Y)
    qc, Yc = q.contiguous().float(), Y.contiguous().float()
    out = torch.empty(Yc.shape[0], dtype=torch.float32, device=Yc.device)
    check(lib().w3d_dist_row(ptr(qc), ptr(Yc), Yc.shape[0], Yc.shape[1], ptr(out), stream_ptr()), "w3d_dist_row")
    return out


def rowmax_dist(X: torch.Tensor, Y: torch.Tensor = None) -> torch.Tensor:
    """cdist(X, Y).max(1) without the matrix (NB11:731-748)."""
    require_cuda(X)
    Xc = X.contiguous().float(); Yc = Xc if Y is None else Y.contiguous().float()
    out = torch.empty(Xc.shape[0], dtype=torch.float32, device=Xc.device)
    check(lib().w3d_rowmax_dist(ptr(Xc), ptr(Yc), Xc.shape[0], Yc.shape[0], Xc.shape[1], ptr(out), stream_ptr()), "w3d_rowmax_dist")
    return out


def list_max_fanin(idx, n_points: int) -> int:
    """Largest number of list entries (columns >= 1) naming one point: sizes the fixed-point headroom of the LDS-window backward.
    Setup-time (frozen lists): one blocking read."""
    R, K = idx.shape
    counts = torch.empty(n_points, dtype=torch.int32, device=idx.device); out = torch.empty(1, dtype=torch.int32, device=idx.device)
    check(lib().w3d_list_max_fanin(ptr(idx), R, K, n_points, ptr(counts), ptr(out), stream_ptr()), "w3d_list_max_fanin")
    return max(int(out.item()), 1)


def absmax_16(t: torch.Tensor) -> float:
    """max |t| of a bf16 / f16 tensor (setup-time: one blocking read)."""
    out = torch.empty(1, dtype=torch.float32, device=t.device)
    check(lib().w3d_absmax_16(ptr(t), _dtype_code(t), t.numel(), ptr(out), stream_ptr()), "w3d_absmax_16")
    return float(out.item())


absmax_bf16 = absmax_16


def raw_descriptor_bwd_window(X4, idx, gF, gX, max_fanin: int, n_points=None, window: int = 0):
    """Row-order backward through a fixed-point LDS accumulation window (+= into gX [n,3]); any list is handled, local lists are fast."""
    R, K = idx.shape
    n = gX.shape[0] if n_points is None else n_points
    check(lib().w3d_descriptor_bwd_window_x4(ptr(X4), ptr(idx), R, K, ptr(gF), n, window, max_fanin, ptr(gX), stream_ptr()), "w3d_descriptor_bwd_window_x4")


class SlotPlan:
    """Slot plan of a FROZEN neighbour list for the fused backward (w3d_knn_descriptor_bwd_fused_slots): the rows are cut into blocks of `rows_per_block`;
    per block, the sorted distinct points its rows name (`dest`, `dest_off`) and, per entry, its point's rank among them (`nbr_slot`, int16 bits, 0xffff = the
    block's set is larger than `max_set` and this point fell outside: a global atomic in the kernel).  Built once, block chunks at a time (torch.unique on
    (block, point) keys)."""
    MAX_SET = 13312

    def __init__(self, idx: torch.Tensor, n_points: int, rows_per_block: int = 2048, max_set: int = 0, chunk_blocks: int = 32):
        R, K = idx.shape
        d = K - 1
        dev = idx.device
        cap = int(max_set) if max_set else self.MAX_SET
        rows = int(min(rows_per_block, max(R, 1)))
        nb = (R + rows - 1) // rows
        self.rows_per_block, self.n_points, self.R, self.K = rows, int(n_points), R, K
        self.nbr_slot = torch.empty(R, d, dtype=torch.int16, device=dev)
        dests, counts, outside = [], [], 0
        for b0 in range(0, nb, chunk_blocks):
            r0, r1 = b0 * rows, min(R, (b0 + chunk_blocks) * rows)
            blk = (torch.arange(r0, r1, device=dev) // rows - b0)                           # local block of each row
            key = blk[:, None] * int(n_points) + idx[r0:r1, 1:].long()
            uniq, inv = torch.unique(key, return_inverse=True)                              # sorted: by block, then by point
            ublk = uniq // int(n_points)
            nblk = int(blk[-1].item()) + 1
            cnt = torch.bincount(ublk, minlength=nblk)
            start = torch.cumsum(cnt, 0) - cnt
            rank = inv - start[blk][:, None]                                               # place of the entry's point in its block's set
            out = rank >= cap
            outside += int(out.sum().item())
            self.nbr_slot[r0:r1] = torch.where(out, torch.full_like(rank, -1), rank).to(torch.int16)
            urank = torch.arange(uniq.numel(), device=dev) - start[ublk]
            keep = urank < cap
            dests.append((uniq[keep] % int(n_points)).to(torch.int32))
            counts.append(torch.clamp(cnt, max=cap))
        cnt_all = torch.cat(counts) if counts else torch.zeros(0, dtype=torch.long, device=dev)
        self.dest = torch.cat(dests).contiguous() if dests else torch.zeros(0, dtype=torch.int32, device=dev)
        self.dest_off = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), torch.cumsum(cnt_all, 0)]).to(torch.int32).contiguous()
        self.max_set = int(cnt_all.max().item()) if nb else 1
        self.outside_fraction = outside / max(R * d, 1)
        self.blocks = nb
        # every entry has a slot: the kernels need the lists' first column only (the roots) and skip the index rows altogether
        self.root = idx[:, 0].contiguous() if outside == 0 and R > 0 else None


class NeighbourSumCache:
    """State of the fused backward's neighbour-sum cache (w3d_knn_descriptor_bwd_fused_slots_cached): per content row the sum of its k style rows (f32 [R, d]) and the
    list it was formed from (int32 [R, k]; -1 = none).  invalidate() whenever the style rows change."""

    def __init__(self, R: int, d: int, k: int, device):
        self.R, self.d, self.k = R, d, k
        self.sums = torch.zeros(R, d, dtype=torch.float32, device=device)     # (zeroed: an entry is only ever read after the pass that wrote it, but nothing uninitialised can leak)
        self.idx = torch.full((R, k), -1, dtype=torch.int32, device=device)

    @staticmethod
    def bytes_needed(R: int, d: int, k: int) -> int:
        return R * d * 4 + R * k * 4

    def invalidate(self):
        self.idx.fill_(-1)


def raw_knn_descriptor_bwd_fused(X4, desc_idx, FX, FY, fy_absmax: float, nn_idx, scale: float, gX, max_fanin: int, gF_out=None, window: int = 0, nn_dist2=None,
                                 loss_accum=None, slots: "SlotPlan" = None, sum_cache: "NeighbourSumCache" = None):
    """NN-loss backward + descriptor backward in one pass: dL/dF_X never goes through HBM (unless gF_out is given).  With nn_dist2 + loss_accum the
    NN-loss value (scale * sum of the selected squared distances) is accumulated in the same pass.  sum_cache (slot-plan form): rows whose neighbour set is the one
    their cache entry was formed from skip their k gathers."""
    R, K = desc_idx.shape
    assert FX.dtype == FY.dtype
    if slots is not None and sum_cache is not None:
        assert gF_out is None and slots.R == R and slots.K == K and slots.n_points == gX.shape[0]
        assert (sum_cache.R, sum_cache.d, sum_cache.k) == (R, K - 1, nn_idx.shape[1])
        check(lib().w3d_knn_descriptor_bwd_fused_slots_cached(ptr(X4), ptr(desc_idx), R, K, _dtype_code(FX), ptr(FX), ptr(FY), FY.shape[0], fy_absmax, ptr(nn_idx), nn_idx.shape[1],
                                                              scale, gX.shape[0], max_fanin, ptr(gX), ptr(nn_dist2) if nn_dist2 is not None else None,
                                                              ptr(loss_accum) if loss_accum is not None else None, slots.rows_per_block, max(slots.max_set, 1),
                                                              ptr(slots.nbr_slot), ptr(slots.dest), ptr(slots.dest_off), ptr(slots.root) if slots.root is not None else None,
                                                              ptr(sum_cache.sums), ptr(sum_cache.idx), stream_ptr()), "w3d_knn_descriptor_bwd_fused_slots_cached")
        return
    if slots is not None:
        assert gF_out is None and slots.R == R and slots.K == K and slots.n_points == gX.shape[0]
        check(lib().w3d_knn_descriptor_bwd_fused_slots(ptr(X4), ptr(desc_idx), R, K, _dtype_code(FX), ptr(FX), ptr(FY), FY.shape[0], fy_absmax, ptr(nn_idx), nn_idx.shape[1], scale,
                                                       gX.shape[0], max_fanin, ptr(gX), ptr(nn_dist2) if nn_dist2 is not None else None,
                                                       ptr(loss_accum) if loss_accum is not None else None, slots.rows_per_block, max(slots.max_set, 1), ptr(slots.nbr_slot),
                                                       ptr(slots.dest), ptr(slots.dest_off), ptr(slots.root) if slots.root is not None else None, stream_ptr()),
              "w3d_knn_descriptor_bwd_fused_slots")
        return
    check(lib().w3d_knn_descriptor_bwd_fused_x4(ptr(X4), ptr(desc_idx), R, K, _dtype_code(FX), ptr(FX), ptr(FY), FY.shape[0], fy_absmax, ptr(nn_idx), nn_idx.shape[1], scale,
                                                gX.shape[0], window, max_fanin, ptr(gF_out) if gF_out is not None else None, ptr(gX),
                                                ptr(nn_dist2) if nn_dist2 is not None else None, ptr(loss_accum) if loss_accum is not None else None, stream_ptr()),
          "w3d_knn_descriptor_bwd_fused_x4")


def raw_descriptor_mse(X, idx, target, weight: float, loss_accum, gX, seg_rows: int = 0):
    R, K = idx.shape
    check(lib().w3d_descriptor_mse_fwd_bwd(ptr(X), ptr(idx), ptr(target), R, K, seg_rows, weight, ptr(loss_accum), ptr(gX), stream_ptr()),
          "w3d_descriptor_mse_fwd_bwd")


def raw_descriptor_mse_seg(X, csr, target, n_per_seg: int, S: int, weight: float, loss_accum, gX):
    """Segment-batched descriptor MSE through a shared CSR (csr: DescriptorCSR built on the LOCAL lists)."""
    if csr.rootsum.shape[0] < S * csr.R:
        csr.rootsum = torch.empty(S * csr.R, 3, dtype=torch.float32, device=X.device)
    check(lib().w3d_descriptor_mse_seg_csr(ptr(X), ptr(csr.idx), ptr(target), csr.R, csr.K, n_per_seg, S, ptr(csr.row_ptr), ptr(csr.entries), weight,
                                           ptr(loss_accum), ptr(gX), ptr(csr.rootsum), stream_ptr()), "w3d_descriptor_mse_seg_csr")


class DescriptorCSR:
    """Transposed neighbour list (built once; the lists are frozen, NB11:1055-1063) for the atomic-free,
    bitwise-reproducible backward."""

    def __init__(self, idx: torch.Tensor, n_points: int):
        require_cuda(idx)
        self.idx = as_i32(idx).contiguous()
        self.R, self.K = self.idx.shape
        self.n_points = int(n_points)
        dev = idx.device
        self.row_ptr = torch.empty(n_points + 1, dtype=torch.int32, device=dev)
        self.entries = torch.empty(2 * self.R * self.K, dtype=torch.int32, device=dev)
        self.rootsum = torch.empty(self.R, 3, dtype=torch.float32, device=dev)
        nbytes = lib().w3d_csr_workspace_bytes(n_points, self.R, self.K)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        check(lib().w3d_csr_build(ptr(self.idx), self.R, self.K, n_points, ptr(self.row_ptr), ptr(self.entries), ptr(ws), nbytes, stream_ptr()),
              "w3d_csr_build")

    def backward_x4(self, X4, gF, gX, accumulate=True):
        check(lib().w3d_descriptor_bwd_csr_x4(ptr(X4), ptr(self.idx), self.R, self.K, ptr(gF), ptr(self.row_ptr), ptr(self.entries), self.n_points,
                                              ptr(self.rootsum), int(accumulate), ptr(gX), stream_ptr()), "w3d_descriptor_bwd_csr_x4")

    def backward(self, X, gF, gX, accumulate=True):
        check(lib().w3d_descriptor_bwd_csr(ptr(X), ptr(self.idx), self.R, self.K, ptr(gF), ptr(self.row_ptr), ptr(self.entries), self.n_points,
                                           ptr(self.rootsum), int(accumulate), ptr(gX), stream_ptr()), "w3d_descriptor_bwd_csr")


class _GetDescriptors(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, idx32):
        Xc = X.contiguous()
        ctx.save_for_backward(Xc, idx32)
        return raw_descriptor_fwd(Xc, idx32)

    @staticmethod
    def backward(ctx, gF):
        X, idx32 = ctx.saved_tensors
        gX = torch.zeros_like(X)
        raw_descriptor_bwd(X, idx32, gF.contiguous(), gX)
        return gX, None


def get_descriptors(X: torch.Tensor, X_nns_indices: torch.Tensor) -> torch.Tensor:
    """Drop-in for the notebook's get_descriptors(X, X_nns_indices) (NB11:994-1009)."""
    require_cuda(X, X_nns_indices)
    if X.shape[-1] != 3:
        raise _lib.W3dError("get_descriptors: X must be [n,3]")
    return _GetDescriptors.apply(X, as_i32(X_nns_indices).contiguous())


class _GetDescriptorsVec(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, V, idx32):
        Xc, Vc = X.contiguous(), V.contiguous()
        ctx.save_for_backward(Xc, Vc, idx32)
        R, K = idx32.shape
        F = torch.empty(R, 4 * (K - 1), dtype=torch.float32, device=X.device)
        if R == 0:
            return F
        check(lib().w3d_descriptor_vec_fwd(ptr(Xc), ptr(Vc), ptr(idx32), R, K, ptr(F), stream_ptr()), "w3d_descriptor_vec_fwd")
        return F

    @staticmethod
    def backward(ctx, gF):
        X, V, idx32 = ctx.saved_tensors
        R, K = idx32.shape
        gX, gV = torch.zeros_like(X), torch.zeros_like(V)
        if R == 0:
            return gX, gV, None
        check(lib().w3d_descriptor_vec_bwd(ptr(X), ptr(V), ptr(idx32), R, K, ptr(gF.contiguous()), ptr(gX), ptr(gV), stream_ptr()), "w3d_descriptor_vec_bwd")
        return gX, gV, None


def get_descriptors_vec(X: torch.Tensor, X_vec: torch.Tensor, X_nns_indices: torch.Tensor) -> torch.Tensor:
    """Drop-in for the notebook's get_descriptors_vec(X, X_vec, X_nns_indices) (visualize_and_fit_cluster.ipynb cell 26, raw lines 771-808):
    X [n,3], X_vec [R,4] (one 4-vector per list row), X_nns_indices [R,K] -> [R, 4*(K-1)]."""
    require_cuda(X, X_vec, X_nns_indices)
    if X.shape[-1] != 3 or X_vec.shape[-1] != 4 or X_vec.shape[0] != X_nns_indices.shape[0]:
        raise _lib.W3dError("get_descriptors_vec: X must be [n,3], X_vec [R,4] with one row per index row")
    if X.dtype != torch.float32 or X_vec.dtype != torch.float32:
        raise _lib.W3dError("get_descriptors_vec: float32 inputs")
    return _GetDescriptorsVec.apply(X, X_vec, as_i32(X_nns_indices).contiguous())


# ------------------------------------------------------------------------------------------------
# domain loss (reference regime)
# ------------------------------------------------------------------------------------------------
def raw_domain_loss(X, rows_per_seg: int, Y, y_offsets, k: int, weight: float, loss_accum, gX, kth_out=None, max_M: int = 0):
    S = y_offsets.numel() - 1
    d = X.shape[1]
    check(lib().w3d_domain_loss_fwd_bwd(ptr(X), rows_per_seg, d, ptr(Y), ptr(y_offsets), S, max_M, k, weight, ptr(loss_accum), ptr(gX), ptr(kth_out),
                                        stream_ptr()), "w3d_domain_loss_fwd_bwd")


class _DomainLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, domain, k):
        Xc = X.contiguous().float(); Y = domain.contiguous().float()
        loss = torch.zeros(1, dtype=torch.float32, device=X.device)
        gX = torch.zeros_like(Xc)
        off = torch.tensor([0, Y.shape[0]], dtype=torch.int32, device=X.device)
        raw_domain_loss(Xc, Xc.shape[0], Y, off, k, 1.0, loss, gX, None, Y.shape[0])
        ctx.save_for_backward(gX)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (gX,) = ctx.saved_tensors
        return gX * g, None, None


def get_loss_domain_adaptation(X, domain, domain_NNs: int = 20):
    """Drop-in for NB11:1011-1023 (ties at the k-th distance are kept in the mask, as in the reference)."""
    require_cuda(X, domain)
    return _DomainLoss.apply(X, domain, domain_NNs)


# ------------------------------------------------------------------------------------------------
# image losses
# ------------------------------------------------------------------------------------------------
def raw_tv_loss(img, weight: float, loss_accum, grad, accumulate: bool):
    if img.dim() == 2:
        Cc, H, W = 1, img.shape[0], img.shape[1]
    else:
        H, W = img.shape[-2:]
        Cc = img.numel() // (H * W)
    check(lib().w3d_tv_loss_fwd_bwd(ptr(img), Cc, H, W, weight, int(accumulate), ptr(loss_accum), ptr(grad), stream_ptr()), "w3d_tv_loss_fwd_bwd")


def raw_mse_loss(a, b, weight: float, loss_accum, grad, accumulate: bool):
    check(lib().w3d_mse_loss_fwd_bwd(ptr(a), ptr(b), a.numel(), weight, int(accumulate), ptr(loss_accum), ptr(grad), stream_ptr()), "w3d_mse_loss_fwd_bwd")


class _TvLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img):
        x = img.contiguous().float()
        loss = torch.zeros(1, dtype=torch.float32, device=x.device)
        g = torch.empty_like(x)
        raw_tv_loss(x, 1.0, loss, g, False)
        ctx.save_for_backward(g)
        return loss[0]

    @staticmethod
    def backward(ctx, go):
        (g,) = ctx.saved_tensors
        return g * go


class _L2Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        x = a.contiguous().float(); y = b.contiguous().float()
        loss = torch.zeros(1, dtype=torch.float32, device=x.device)
        g = torch.empty_like(x)
        raw_mse_loss(x, y, 1.0, loss, g, False)
        ctx.save_for_backward(g)
        return loss[0]

    @staticmethod
    def backward(ctx, go):
        (g,) = ctx.saved_tensors
        return g * go, -(g * go)


class _Ssim(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2, size_average):
        require_cuda(img1, img2)
        assert img1.shape == img2.shape and img1.dim() in (3, 4)
        a = img1.detach().float().contiguous(); b = img2.detach().float().contiguous()
        H, W = a.shape[-2:]
        planes = a.numel() // (H * W)
        sums = torch.zeros(planes, device=a.device)
        need = img1.requires_grad
        dm = torch.empty(3, a.numel(), device=a.device) if need else None
        check(lib().w3d_ssim_fwd(ptr(a), ptr(b), planes, H, W, ptr(sums), ptr(dm) if need else None, stream_ptr()), "w3d_ssim_fwd")
        ctx.save_for_backward(a, b, dm)
        ctx.meta = (planes, H, W, size_average, tuple(img1.shape))
        if size_average or a.dim() == 3:                 # (a 3-d image with size_average=False: the reference's mean(1).mean(1).mean(1) is the mean over everything too)
            return sums.sum() / a.numel()
        per = a.shape[1] * H * W                         # [B,C,H,W]: one value per batch item
        return sums.view(a.shape[0], -1).sum(1) / per

    @staticmethod
    def backward(ctx, go):
        a, b, dm = ctx.saved_tensors
        planes, H, W, size_average, shape = ctx.meta
        if size_average or len(shape) == 3:
            pw = (go.float().reshape(()) / a.numel()).expand(planes).contiguous()
        else:
            pw = (go.float() / (shape[1] * H * W)).repeat_interleave(shape[1]).contiguous()
        g = torch.empty_like(a)
        check(lib().w3d_ssim_bwd(ptr(a), ptr(b), ptr(dm), planes, H, W, ptr(pw), 0, ptr(g), stream_ptr()), "w3d_ssim_bwd")
        return g.view(shape), None, None


def ssim(img1, img2, window_size: int = 11, size_average: bool = True):
    """utils/loss_utils.py:34-64 (the reference's D-SSIM term: train.py:93 `1.0 - ssim(image, gt_image)`): [C,H,W] or [B,C,H,W] images, 11x11 Gaussian window per
    plane, zero padding.  Differentiable wrt img1 (the render); img2 is treated as a constant, as the ground truth is in every caller."""
    if window_size != 11:
        raise NotImplementedError("ssim: the reference calls it with its default window_size = 11 only; the kernel is built for that window")
    if img2.requires_grad:
        raise NotImplementedError("ssim: gradient wrt the second image is not implemented (the reference's callers pass the ground truth there)")
    return _Ssim.apply(img1, img2, bool(size_average))


def tv_loss(img):
    """utils/loss_utils.py:213-215."""
    require_cuda(img)
    return _TvLoss.apply(img)


def l2_loss(network_output, gt):
    """utils/loss_utils.py:23-24."""
    require_cuda(network_output, gt)
    return _L2Loss.apply(network_output, gt)


# ------------------------------------------------------------------------------------------------
# Adam
# ------------------------------------------------------------------------------------------------
def raw_adam_step(groups, step: int, beta1: float, beta2: float, eps: float, zero_grad: bool, lean: bool = False):
    """groups: list of (param, grad, exp_avg, exp_avg_sq, lr[, grad2]) flat fp32 tensors; grad2 = a second gradient buffer summed in the step.
    lean: the 32-VGPR form of the kernel (same arithmetic), which fits on a SIMD next to the persistent emit sweep's two waves (w3d_adam_step, zero_grad bit 1)."""
    n = len(groups)
    arr = (_lib.AdamGroup * n)()
    for i, grp in enumerate(groups):
        p, g, m, v, lr = grp[:5]
        arr[i].param = p.data_ptr(); arr[i].grad = g.data_ptr(); arr[i].exp_avg = m.data_ptr(); arr[i].exp_avg_sq = v.data_ptr()
        arr[i].numel = p.numel(); arr[i].lr = lr
        arr[i].grad2 = grp[5].data_ptr() if len(grp) > 5 and grp[5] is not None else None
    check(lib().w3d_adam_step(arr, n, step, beta1, beta2, eps, int(bool(zero_grad)) | (2 if lean else 0), stream_ptr()), "w3d_adam_step")


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam replacement (no amsgrad / weight decay — the reference uses neither): one HIP launch per
    step over all parameter groups (up to 8 tensors per launch), optional fused zero_grad.
    Keeps `param_groups` (with per-group 'lr' and 'name', scene/gaussian_model.py:154-161) and a
    torch-compatible state_dict (exp_avg / exp_avg_sq / step) so capture()/restore() round-trip."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None, zero_grad: bool = False):
        assert closure is None
        batches = {}
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                require_cuda(p)
                if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise _lib.W3dError("FusedAdam needs contiguous float32 parameters and gradients")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                key = (int(st["step"].item()), b1, b2, group["eps"])
                batches.setdefault(key, []).append((p, p.grad, st["exp_avg"], st["exp_avg_sq"], float(group["lr"])))
        for (t, b1, b2, eps), items in batches.items():
            for i in range(0, len(items), 8):
                raw_adam_step(items[i:i + 8], t, b1, b2, eps, zero_grad)


# ------------------------------------------------------------------------------------------------
# script variant (AUXW4) and the multi-cluster coverage head (NBVM)
# ------------------------------------------------------------------------------------------------
def target_kth_distance(t_xyz: torch.Tensor, k: int = 100) -> torch.Tensor:
    """k-th smallest pairwise distance of every frozen target row (aux_optimize_cluster_D_W_distance4.py:80-84), matrix-free."""
    t = t_xyz.contiguous().float()
    N = t.shape[0]
    kth = torch.empty(N, device=t.device); dummy_l = torch.zeros(1, device=t.device); dummy_g = torch.zeros_like(t)
    off = torch.tensor([0, N], dtype=torch.int32, device=t.device)
    raw_domain_loss(t, N, t, off, k, 0.0, dummy_l, dummy_g, kth, N)
    return kth


def auxw_pairwise_losses(xyz, rot, scale, t_xyz, t_rot, t_scale, kth_target, weights=(1.0, 1.0, 1.0)):
    """Returns (loss3 [3] = loss_D_xyz, loss_D_rotation, loss_D_scaling, g_xyz, g_rot, g_scale) — AUXW4:254-266 fused."""
    require_cuda(xyz, rot, scale, t_xyz, t_rot, t_scale, kth_target)
    c = lambda t: t.contiguous().float()
    xyz, rot, scale, t_xyz, t_rot, t_scale = map(c, (xyz, rot, scale, t_xyz, t_rot, t_scale))
    loss = torch.zeros(3, device=xyz.device); gx = torch.zeros_like(xyz); gr = torch.zeros_like(rot); gs = torch.zeros_like(scale)
    check(lib().w3d_auxw_pairwise_fwd_bwd(ptr(xyz), ptr(rot), ptr(scale), ptr(t_xyz), ptr(t_rot), ptr(t_scale), ptr(kth_target), xyz.shape[0], weights[0],
                                          weights[1], weights[2], ptr(loss), ptr(gx), ptr(gr), ptr(gs), stream_ptr()), "w3d_auxw_pairwise_fwd_bwd")
    return loss, gx, gr, gs


def raw_coverage_bwd(domain, points, nn_idx, nn_d2, weight: float, loss_accum, g_points):
    """loss_accum[0] += weight * mean_j sqrt(nn_d2[j]) (weight 0: the unweighted value is accumulated instead, nothing is scattered);
    g_points += weight * d/d points (NULL: value only)."""
    w = float(weight) if g_points is not None else 1.0
    check(lib().w3d_coverage_loss_bwd(ptr(domain), ptr(points), ptr(nn_idx), ptr(nn_d2), domain.shape[0], domain.shape[1], w, ptr(loss_accum), ptr(g_points), stream_ptr()),
          "w3d_coverage_loss_bwd")


def coverage_loss(domain, points, weight: float = 1.0):
    """loss_domain_coverage (NBVM cells 13-20) for the concatenation `points` of all fitted clusters: returns (loss, g_points)."""
    require_cuda(domain, points)
    dm, pt = domain.contiguous().float(), points.contiguous().float()
    idx, d2 = raw_knn_match(dm, pt, 1)
    loss = torch.zeros(1, device=dm.device); g = torch.zeros_like(pt)
    check(lib().w3d_coverage_loss_bwd(ptr(dm), ptr(pt), ptr(idx), ptr(d2), dm.shape[0], dm.shape[1], weight, ptr(loss), ptr(g), stream_ptr()), "w3d_coverage_loss_bwd")
    return loss[0], g


class _SlicedWasserstein(torch.autograd.Function):
    """Clouds beyond the LDS-resident head's reach (w3d_sw_max_points): caller-side projections (library GEMM) + rocPRIM segmented sort + elementwise head."""

    @staticmethod
    def forward(ctx, FX, FY, proj):
        px = (FX.float() @ proj.t()).t().contiguous()                 # [L,N]  plain library GEMM (rocBLAS through torch)
        py = torch.sort((FY.float() @ proj.t()).t().contiguous(), dim=1).values.contiguous()
        L, N = px.shape
        assert py.shape == px.shape, "sliced-Wasserstein head needs clouds of equal size"
        loss = torch.zeros(1, device=FX.device); g = torch.empty_like(px)
        nbytes = lib().w3d_sw_workspace_bytes(L, N)
        ws = _workspace(nbytes, FX.device)
        check(lib().w3d_sw_loss_fwd_bwd(ptr(px), ptr(py), L, N, 1.0, ptr(loss), ptr(g), ptr(ws), ws.numel(), stream_ptr()), "w3d_sw_loss_fwd_bwd")
        ctx.save_for_backward(g, proj)
        return loss[0]

    @staticmethod
    def backward(ctx, go):
        g, proj = ctx.saved_tensors
        return (g.t() @ proj) * go, None, None


def sw_sorted_projections(F: torch.Tensor, proj: torch.Tensor, with_perm: bool = False):
    """Every projection <F_i, proj_l> of the rows of F [N,d], sorted ascending per projection: [L,N] f32 (+ the permutation, a stable argsort) — the constant style
    side of the sliced-Wasserstein head, once per style set (w3d_sw_project_sort: one workgroup per projection, projections and sort in LDS)."""
    require_cuda(F, proj)
    Fc, pc = F.detach().float().contiguous(), proj.detach().float().contiguous()
    N, d = Fc.shape
    L = pc.shape[0]
    out = torch.empty(L, N, dtype=torch.float32, device=Fc.device)
    perm = torch.empty(L, N, dtype=torch.int32, device=Fc.device) if with_perm else None
    check(lib().w3d_sw_project_sort(ptr(Fc), ptr(pc), N, d, L, ptr(out), ptr(perm), stream_ptr()), "w3d_sw_project_sort")
    return (out, perm) if with_perm else out


def raw_sw_features_fwd_bwd(FX, proj, py_sorted, loss_accum, g_FX, weight: float = 1.0):
    """loss_accum[0] += weight * SW(FX, style); g_FX [N,d] = its gradient (None: loss only).  Two launches, no autograd (the C1 loop of bench.py runs on this)."""
    N, d = FX.shape
    L = proj.shape[0]
    ws = _workspace(lib().w3d_sw_features_workspace_bytes(L, N), FX.device)
    check(lib().w3d_sw_features_fwd_bwd(ptr(FX), ptr(proj), ptr(py_sorted), N, d, L, float(weight), ptr(loss_accum), ptr(g_FX), ptr(ws), ws.numel(), stream_ptr()),
          "w3d_sw_features_fwd_bwd")


class _SlicedWassersteinFused(torch.autograd.Function):
    @staticmethod
    def forward(ctx, FX, py_sorted, proj):
        Fc = FX.detach().float().contiguous()
        loss = torch.zeros(1, device=Fc.device); g = torch.empty_like(Fc)
        raw_sw_features_fwd_bwd(Fc, proj, py_sorted, loss, g)
        ctx.save_for_backward(g)
        return loss[0]

    @staticmethod
    def backward(ctx, go):
        (g,) = ctx.saved_tensors
        return g * go, None, None


def sliced_wasserstein_loss(FX, FY, proj, py_sorted=None):
    """C1 head: mean over L unit projections of the 1-D squared Wasserstein-2 distance (not in the reference; parity unpinned).  `py_sorted`: the style side's sorted
    projections from sw_sorted_projections(FY, proj) when the caller keeps them across iterations (the style set is constant); FY may then be None."""
    require_cuda(FX, proj)
    pc = proj.detach().float().contiguous()
    N = FX.shape[0]
    if N <= lib().w3d_sw_max_points():
        if py_sorted is None:
            assert FY.shape[0] == N, "sliced-Wasserstein head needs clouds of equal size"
            py_sorted = sw_sorted_projections(FY, pc)
        assert tuple(py_sorted.shape) == (pc.shape[0], N)
        return _SlicedWassersteinFused.apply(FX, py_sorted, pc)
    return _SlicedWasserstein.apply(FX, FY, proj)


# ------------------------------------------------------------------------------------------------
# a5  exact EMD between uniform, equally sized clouds (aux_optimize_cluster_D_W_distance.py:259-270)
# ------------------------------------------------------------------------------------------------
class _SqDistMatrix(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2):
        a, b = x1.detach().float().contiguous(), x2.detach().float().contiguous()
        out = torch.empty(a.shape[0], b.shape[0], dtype=torch.float32, device=a.device)
        check(lib().w3d_sqdist_matrix_fwd(ptr(a), ptr(b), a.shape[0], b.shape[0], a.shape[1], ptr(out), stream_ptr()), "w3d_sqdist_matrix_fwd")
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, go):
        a, b = ctx.saved_tensors
        g = go.float().contiguous()
        ga = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        gb = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        if ga is not None or gb is not None:
            check(lib().w3d_sqdist_matrix_bwd(ptr(g), ptr(a), ptr(b), a.shape[0], b.shape[0], a.shape[1], ptr(ga), ptr(gb), stream_ptr()), "w3d_sqdist_matrix_bwd")
        return ga, gb


def dist(x1: torch.Tensor, x2: torch.Tensor = None) -> torch.Tensor:
    """`ot.dist(x1, x2)` with its default metric (aux_optimize_cluster_D_W_distance.py:268): the squared Euclidean cost matrix [n1, n2], differentiable in both clouds
    (w3d_sqdist_matrix_fwd / _bwd: POT's published max(|a|^2 + |b|^2 - 2 a.b, 0); `emd2_points` fuses it with the assignment and never materialises it)."""
    x2 = x1 if x2 is None else x2
    require_cuda(x1, x2)
    if x1.dim() != 2 or x2.dim() != 2 or x1.shape[1] != x2.shape[1]:
        raise W3dError("dist: two [n, d] clouds of the same feature width")
    return _SqDistMatrix.apply(x1, x2)


class _Emd2Matrix(torch.autograd.Function):
    @staticmethod
    def forward(ctx, M):
        Mc = M.contiguous().float()
        B = 1 if Mc.dim() == 2 else Mc.shape[0]
        n = Mc.shape[-1]
        loss = torch.zeros(B, device=Mc.device); g = torch.zeros_like(Mc)
        match = torch.empty(B, n, dtype=torch.int32, device=Mc.device)
        check(lib().w3d_emd2_matrix_fwd_bwd(ptr(Mc), B, n, n, 1.0, ptr(loss), ptr(g), ptr(match), stream_ptr()), "w3d_emd2_matrix_fwd_bwd")
        ctx.save_for_backward(g)
        ctx.mark_non_differentiable(match)
        return (loss[0] if M.dim() == 2 else loss), (match[0] if M.dim() == 2 else match)

    @staticmethod
    def backward(ctx, go, _gm):
        (g,) = ctx.saved_tensors
        return g * (go if g.dim() == 2 else go.view(-1, 1, 1))


def emd2(a, b, M: torch.Tensor, return_matching: bool = False):
    """Drop-in for `ot.emd2(a, b, M)` as the reference calls it: a = b = ones(n)/n, M [n,n] (or a batch [B,n,n]).  Other weightings are
    not on the reference's path and raise.  Gradient wrt M = the optimal plan (POT's convention)."""
    require_cuda(M)
    n = M.shape[-1]
    if M.shape[-2] != n:
        raise W3dError("emd2: only equally sized clouds (square cost matrix) are supported")
    for w in (a, b):
        if w is not None and len(w) and (len(w) != n or not torch.allclose(torch.as_tensor(w, dtype=torch.float32).cpu(), torch.full((n,), 1.0 / n), rtol=1e-5, atol=1e-8)):
            raise W3dError("emd2: only uniform weights 1/n are supported")
    loss, match = _Emd2Matrix.apply(M)
    return (loss, match) if return_matching else loss


class _Emd2Points(torch.autograd.Function):
    @staticmethod
    def forward(ctx, P, Q):
        Pc, Qc = P.contiguous().float(), Q.contiguous().float()
        B = 1 if Pc.dim() == 2 else Pc.shape[0]
        n, d = Pc.shape[-2], Pc.shape[-1]
        loss = torch.zeros(B, device=Pc.device); gP = torch.zeros_like(Pc); gQ = torch.zeros_like(Qc)
        check(lib().w3d_emd2_points_fwd_bwd(ptr(Pc), ptr(Qc), B, n, d, 1.0, ptr(loss), ptr(gP), ptr(gQ), None, stream_ptr()), "w3d_emd2_points_fwd_bwd")
        ctx.save_for_backward(gP, gQ)
        return loss[0] if P.dim() == 2 else loss

    @staticmethod
    def backward(ctx, go):
        gP, gQ = ctx.saved_tensors
        s = go if gP.dim() == 2 else go.view(-1, 1, 1)
        return gP * s, gQ * s


def emd2_points(P: torch.Tensor, Q: torch.Tensor):
    """emd2(1/n, 1/n, ot.dist(P, Q)) fused: P, Q [n,d] or [B,n,d] (d <= 8, n <= 1024), differentiable in both clouds."""
    require_cuda(P, Q)
    if P.shape != Q.shape:
        raise W3dError("emd2_points: clouds must have the same shape")
    return _Emd2Points.apply(P, Q)


class _DualEntropic(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, v, xs, xt, reg, ws, wt):
        require_cuda(u, v, xs, xt)
        f = lambda t: None if t is None else t.detach().float().contiguous()
        uc, vc, xsc, xtc, wsc, wtc = f(u), f(v), f(xs), f(xt), f(ws), f(wt)
        ns, d = xsc.shape; nt = xtc.shape[0]
        if xtc.shape[1] != d or uc.numel() != ns or vc.numel() != nt:
            raise W3dError("loss_dual_entropic: u [ns], v [nt], xs [ns,d], xt [nt,d] expected")
        dev = uc.device
        loss = torch.zeros(1, device=dev); gu = torch.empty(ns, device=dev); gv = torch.empty(nt, device=dev)
        need_x = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        gxs = torch.empty(ns, d, device=dev) if need_x else None
        gxt = torch.empty(nt, d, device=dev) if need_x else None
        scratch = torch.zeros(8, dtype=torch.uint8, device=dev)
        check(lib().w3d_dual_entropic_fwd_bwd(ptr(uc), ptr(vc), ptr(xsc), ptr(xtc), ptr(wsc), ptr(wtc), ns, nt, d, float(reg), 1.0, ptr(loss), ptr(gu), ptr(gv),
                                              ptr(gxs), ptr(gxt), ptr(scratch), stream_ptr()), "w3d_dual_entropic_fwd_bwd")
        ctx.save_for_backward(gu, gv, gxs if need_x else gu, gxt if need_x else gv)
        ctx.need_x = need_x
        ctx.dtypes = (u.dtype, v.dtype, xs.dtype, xt.dtype)
        return loss[0].to(u.dtype)

    @staticmethod
    def backward(ctx, go):
        gu, gv, gxs, gxt = ctx.saved_tensors
        du, dv, dxs, dxt = ctx.dtypes
        return ((gu * go).to(du), (gv * go).to(dv), (gxs * go).to(dxs) if ctx.need_x and ctx.needs_input_grad[2] else None,
                (gxt * go).to(dxt) if ctx.need_x and ctx.needs_input_grad[3] else None, None, None, None)


def loss_dual_entropic(u, v, xs, xt, reg: float = 1, ws=None, wt=None):
    """Drop-in for `ot.stochastic.loss_dual_entropic(u, v, xs, xt, reg, ws, wt)` as notebooks/test_EMD_local_stats.ipynb calls it (raw lines 208, 994; squared-Euclidean
    cost, uniform weights by default): the dual objective of entropic OT, differentiable in the potentials u, v (and in the samples).  Computation in f32 storage with
    f64 sums on the GPU; the result is returned in u's dtype."""
    return _DualEntropic.apply(u, v, xs, xt, reg, ws, wt)


def plan_dual_entropic(u, v, xs, xt, reg: float = 1, ws=None, wt=None):
    """Drop-in for `ot.stochastic.plan_dual_entropic`: ws_i exp((u_i + v_j - |xs_i - xt_j|^2) / reg) wt_j as an [ns, nt] matrix (not differentiated, as the notebook uses it)."""
    require_cuda(u, v, xs, xt)
    f = lambda t: None if t is None else t.detach().float().contiguous()
    uc, vc, xsc, xtc, wsc, wtc = f(u), f(v), f(xs), f(xt), f(ws), f(wt)
    ns, d = xsc.shape; nt = xtc.shape[0]
    plan = torch.empty(ns, nt, device=uc.device)
    check(lib().w3d_dual_entropic_plan(ptr(uc), ptr(vc), ptr(xsc), ptr(xtc), ptr(wsc), ptr(wtc), ns, nt, d, float(reg), ptr(plan), stream_ptr()), "w3d_dual_entropic_plan")
    return plan


# ------------------------------------------------------------------------------------------------
# a8  VGG-feature losses on given feature maps (utils/loss_utils.py:113-131, 178-205)
# ------------------------------------------------------------------------------------------------
def raw_gram_matrix(feat: torch.Tensor) -> torch.Tensor:
    """G = F F^T for one feature map [1,C,H,W] / [C,H,W] / [C,HW]."""
    f = feat.contiguous().float()
    C = f.shape[-3] if f.dim() >= 3 else f.shape[0]
    HW = f.numel() // C
    G = torch.empty(C, C, device=f.device)
    ws = _workspace(lib().w3d_gram_workspace_bytes(C, HW), f.device)
    check(lib().w3d_gram_matrix(ptr(f), C, HW, ptr(G), ptr(ws), ws.numel(), stream_ptr()), "w3d_gram_matrix")
    return G


class _GramStyleLayer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gen, A):
        f = gen.contiguous().float()
        C = A.shape[0]; HW = f.numel() // C
        loss = torch.zeros(1, device=f.device); g = torch.zeros_like(f)
        ws = _workspace(lib().w3d_gram_workspace_bytes(C, HW), f.device)
        check(lib().w3d_gram_style_loss_fwd_bwd(ptr(f), ptr(A), C, HW, 1.0, ptr(loss), ptr(g), ptr(ws), ws.numel(), stream_ptr()), "w3d_gram_style_loss_fwd_bwd")
        ctx.save_for_backward(g)
        return loss[0]

    @staticmethod
    def backward(ctx, go):
        (g,) = ctx.saved_tensors
        return g * go, None


def style_loss(features_gt, features_pred):
    """Drop-in for utils/loss_utils.py:178-205: sum over layers of mean((Gram(pred) - Gram(gt))^2); differentiable in features_pred.
    The style Grams can be cached by passing [C,C] matrices in features_gt."""
    total = 0
    for gen, style in zip(features_pred, features_gt):
        require_cuda(gen, style)
        A = style if style.dim() == 2 else raw_gram_matrix(style.detach())
        total = total + _GramStyleLayer.apply(gen, A.contiguous().float())
    return total


def content_loss(features_gt, features_pred):
    """Drop-in for utils/loss_utils.py:113-131: sum over layers of mean((gt - pred)^2)."""
    total = 0
    for gt, pred in zip(features_gt, features_pred):
        total = total + l2_loss(pred, gt.detach())
    return total
