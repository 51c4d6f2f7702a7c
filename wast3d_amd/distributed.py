"""One scene over the GPUs of a node (BASELINE config C4): one process per GPU, torch.distributed with the "nccl" backend
(= RCCL over xGMI on ROCm).  The reference is single-GPU; its natural axis of parallelism is "independent units given the
style set" (notebooks/11.fit_cluster_to_the_full_scene.ipynb:987-993) — here the units are the Gaussians of ONE scene.

STRONG scaling is the default: the scene (P_total Gaussians stored along one space-filling curve, frozen descriptor lists with
GLOBAL indices) is block-sharded, rank r owns rows [r*P, (r+1)*P) with their parameters and Adam state.

  geometry leg (no order dependence between rows)
    * positions: the lists of rows near a shard boundary name Gaussians of the neighbouring rank -> ALL-GATHER of the packed positions
      (16 B x P_total, step start) and REDUCE-SCATTER of dL/dxyz (12 B x P_total, after the fused backward).
    * style rows F_Y [M, d]: sharded M/world per rank in a padded layout (every shard starts on a 64-row tile boundary; padding rows score
      -inf and are never selected).  Constant across iterations in the reference (NB11:1057-1063), so the default is ONE all-gather at setup
      ("cached"); `style="inloop"` (north_star's "RCCL all-gather of style-feature tiles") re-delivers them EVERY iteration shard by shard (world broadcasts on their own
      stream and communicator, started at t = 0): they land while the rows are featurised and the render leg runs; the bound kernel — the first user of a style row —
      waits for them, and bound, plan and ONE pruned emit sweep with cached lower bounds follow, exactly the single-GPU sweep (round 6; padding rows carry -inf norms).
      `style="inloop_chunked"` is rounds 1-5's form: the sweep itself is cut per arriving shard (StyleStream + w3d_knn_chunk_*: the sweep of shard c waits only for
      shard c's event) — unpruned list sweeps under the motion bound, the only bound that needs no style row; 2.9x slower than `cached` at C4 / world 1.
  render leg (alpha blending is order dependent per pixel: partial images cannot be summed)
    * each rank projects ITS Gaussians to 48-byte splat records -> ALL-GATHER of the records (48 B x P_total);
    * each rank bins / sorts / blends a contiguous band of tile rows from all records; the bands are re-cut every iteration from the previous
      iteration's per-tile-row instance counts (the scene is denser in the middle rows; every rank computes the same cut from the same records);
    * the bands are combined by an ALL-REDUCE(sum) of one [4,H,W] colour+depth image that is zero outside a rank's band (x + 0 is exact), so the
      image losses are evaluated exactly as on one GPU;
    * each rank back-propagates its band into per-splat accumulators for ALL Gaussians -> REDUCE-SCATTER(sum) to the owners (48 B x P_total).
All buffers are allocated once.  The render leg's collectives run on their own process group (their own RCCL communicator and stream), so they
are not serialised behind the geometry leg's; both overlap the distance-GEMM.  Weak scaling (every rank brings its own Gaussians, `--scaling weak`
in bench.py) is the same loop with lists that stay inside a shard.
"""
from __future__ import annotations

import ctypes as C
import math
from contextlib import contextmanager
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from . import _lib, ops
from ._lib import RasterGradIn, RasterGradOut, check, lib, ptr, stream_ptr

SPLAT = 12   # W3D_SPLAT_FLOATS


# ------------------------------------------------------------------------------------------------ host-side plan (pure python, CPU-testable)
def split_even(n: int, parts: int) -> List[Tuple[int, int]]:
    base, rem = divmod(n, parts)
    out, s = [], 0
    for i in range(parts):
        e = s + base + (1 if i < rem else 0)
        out.append((s, e))
        s = e
    return out


def balanced_bands(cost: List[float], parts: int) -> List[Tuple[int, int]]:
    """Cut len(cost) consecutive tile rows into `parts` contiguous bands of (nearly) equal total cost; every band keeps at least one row when there
    are enough rows.  Deterministic in `cost`: every rank derives the same cut from the same numbers."""
    n = len(cost)
    if n < parts:
        return split_even(n, parts)
    total = float(sum(cost))
    if not total > 0.0:
        return split_even(n, parts)
    cuts, acc, r = [0], 0.0, 1
    for i, c in enumerate(cost):
        acc += float(c)
        # close band r-1 after row i when its share is reached, leaving at least one row for each band still to come
        while r < parts and acc >= total * r / parts and i + 1 >= cuts[-1] + 1 and n - (i + 1) >= parts - r:
            cuts.append(i + 1); r += 1
    while len(cuts) < parts:                               # not enough cost left: hand the remaining rows out one by one
        cuts.append(min(cuts[-1] + 1, n - (parts - len(cuts))))
    cuts.append(n)
    return [(cuts[i], cuts[i + 1]) for i in range(parts)]


class ShardPlan:
    """Who owns which Gaussians / style rows / tile rows."""

    def __init__(self, world: int, rank: int, gaussians_per_rank: int, style_rows: int, H: int, W: int):
        self.world, self.rank, self.P = world, rank, gaussians_per_rank
        self.P_total = world * gaussians_per_rank
        self.gauss_range = (rank * gaussians_per_rank, (rank + 1) * gaussians_per_rank)
        self.M = style_rows
        self.style_ranges = split_even(style_rows, world)        # true row numbers
        # padded layout of the gathered style set: shard c occupies rows [c*Mc_pad, c*Mc_pad + size_c), Mc_pad a multiple of the 64-row tile
        self.Mc_pad = (max(e - s for s, e in self.style_ranges) + 63) // 64 * 64
        self.M_pad = world * self.Mc_pad
        self.tile_rows = (H + 15) // 16
        self.tile_cols = (W + 15) // 16
        self.bands = split_even(self.tile_rows, world)           # contiguous bands of tile rows; re-cut by cost with set_bands()
        self.H, self.W = H, W

    @property
    def band(self) -> Tuple[int, int]:
        return self.bands[self.rank]

    def set_bands(self, cost_per_tile_row) -> None:
        self.bands = balanced_bands([float(c) for c in cost_per_tile_row], self.world)

    def band_pixel_rows(self, r=None) -> Tuple[int, int]:
        b = self.bands[self.rank if r is None else r]
        return b[0] * 16, min(b[1] * 16, self.H)

    def style_padded_to_true(self, idx: torch.Tensor) -> torch.Tensor:
        """Row numbers of the padded layout -> row numbers of the style set itself (shard c's rows start at style_ranges[c][0])."""
        c = torch.div(idx, self.Mc_pad, rounding_mode="floor")
        start = torch.tensor([s for s, _ in self.style_ranges], device=idx.device, dtype=idx.dtype)
        return idx - c * self.Mc_pad + start[c.long()]

    def style_chunk_rows(self, c: int) -> Tuple[int, int, int]:
        """(first row, end of the valid rows, end incl. padding) of shard c in the padded layout."""
        s, e = self.style_ranges[c]
        return c * self.Mc_pad, c * self.Mc_pad + (e - s), (c + 1) * self.Mc_pad


# ------------------------------------------------------------------------------------------------ collectives (thin; CPU-testable with gloo)
def all_gather_rows(local: torch.Tensor, group=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Concatenate equally sized row blocks of every rank (rank order) into `out` (allocated if None)."""
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out


def all_gather_ragged_rows(local: torch.Tensor, sizes: List[int], group=None) -> torch.Tensor:
    """Row blocks whose sizes differ (setup-time helper): pad to the largest, gather, strip."""
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    full = all_gather_rows(pad, group).view((len(sizes), mx) + tuple(local.shape[1:]))
    return torch.cat([full[i, : sizes[i]] for i in range(len(sizes))], 0)


def reduce_scatter_rows(full: torch.Tensor, group=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Sum `full` over ranks and return this rank's equal row block (into `out` if given).  gloo (CPU tests) has no reduce_scatter: all_reduce + slice."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    rows = full.shape[0] // world
    if dist.get_backend(group) == "gloo":
        tmp = full.clone()
        dist.all_reduce(tmp, group=group)
        blk = tmp[rank * rows:(rank + 1) * rows]
        if out is None:
            return blk.clone()
        out.copy_(blk)
        return out
    if out is None:
        out = torch.empty((rows,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
    dist.reduce_scatter_tensor(out, full, group=group)
    return out


COMM_KEYS = ("xyz_all_gather_ms", "gx_reduce_scatter_ms", "splat_all_gather_ms", "image_all_reduce_ms", "accum_reduce_scatter_ms", "style_gather_ms")


class CommTimer:
    """Where a sharded step's communication time goes (the first 8-GPU run must be readable).  `with timer("xyz_all_gather_ms", nbytes):` brackets ONE
    collective with a pair of events on the stream it is issued on — torch's process group makes that stream wait for the collective, so the pair measures the
    collective INCLUDING the wait for the slowest peer, which is what the step pays — or, for CPU tensors (gloo tests, bench.py --config toy), with the host clock.
    summary(steps) = mean ms per step and bytes per step of every collective; the max over ranks is taken by the caller (bench.py).  Disabled: no events, no cost."""

    def __init__(self, enabled: bool = True, cuda: bool = True):
        self.enabled, self.cuda = bool(enabled), bool(cuda)
        self.records = {}                   # name -> [(e0, e1) | seconds]
        self.bytes = {}                     # name -> payload bytes of ONE call (the collective's whole buffer on this rank)
        self.calls = {}

    def __call__(self, name: str, nbytes: int = 0):
        return _CommSpan(self, name, int(nbytes))

    def reset(self):
        self.records, self.calls = {}, {}

    def summary(self, steps: int) -> dict:
        """{name: ms per step} for every COMM_KEYS entry (0.0 when the collective did not run) + "bytes": {name: bytes per step}."""
        if self.cuda and torch.cuda.is_available():
            torch.cuda.synchronize()
        steps = max(int(steps), 1)
        out = {k: 0.0 for k in COMM_KEYS}
        for name, recs in self.records.items():
            ms = sum((a.elapsed_time(b) if not isinstance(a, float) else a * 1e3) for a, b in recs)
            out[name] = ms / steps
        out["bytes"] = {k: int(self.bytes.get(k, 0) * self.calls.get(k, 0) / steps) for k in COMM_KEYS}
        return out


class _CommSpan:
    def __init__(self, t, name, nbytes):
        self.t, self.name, self.nbytes = t, name, nbytes

    def __enter__(self):
        t = self.t
        if not t.enabled:
            return self
        if t.cuda:
            self.e0 = torch.cuda.Event(enable_timing=True); self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        else:
            import time
            self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        t = self.t
        if not t.enabled:
            return False
        if t.cuda:
            self.e1.record()
            t.records.setdefault(self.name, []).append((self.e0, self.e1))
        else:
            import time
            t.records.setdefault(self.name, []).append((time.perf_counter() - self.t0, None))
        t.bytes[self.name] = self.nbytes
        t.calls[self.name] = t.calls.get(self.name, 0) + 1
        return False


def _nbytes(t: torch.Tensor) -> int:
    return t.numel() * t.element_size()


class GeometryExchange:
    """The two collectives of the geometry leg of a sharded scene (GaussianStyleLoop(exchange=...))."""

    def __init__(self, group=None, timer: Optional[CommTimer] = None):
        self.group = group
        self.timer = timer or CommTimer(False)
        self.local = False                    # DistributedStyleLoop.local_collectives(): device-local stand-ins (a measurement aid: the step with free communication)

    def gather_xyz4(self, xyz4_scene: torch.Tensor, xyz4_own: torch.Tensor) -> None:
        # every rank's packed positions into the scene-sized buffer (rank order = row order of the scene)
        if self.local:
            r, n = dist.get_rank(self.group), xyz4_own.shape[0]
            xyz4_scene[r * n:(r + 1) * n].copy_(xyz4_own)
            return
        with self.timer("xyz_all_gather_ms", _nbytes(xyz4_scene)):
            dist.all_gather_into_tensor(xyz4_scene, xyz4_own, group=self.group)

    def scatter_gx(self, gx_scene: torch.Tensor, gx_own: torch.Tensor) -> None:
        if self.local:
            r, n = dist.get_rank(self.group), gx_own.shape[0]
            gx_own.copy_(gx_scene[r * n:(r + 1) * n])
            return
        with self.timer("gx_reduce_scatter_ms", _nbytes(gx_scene)):
            reduce_scatter_rows(gx_scene, self.group, out=gx_own)


class StyleStream:
    """In-loop all-gather of the style rows, shard by shard, under the distance-GEMM: `start()` enqueues one broadcast per shard (source = its owner)
    on a side stream and records an event after each; the sweep of shard c waits for event c only.  The own shard needs no transfer and is swept
    first, the others in rank order after it (the order the broadcasts complete in)."""

    def __init__(self, plan: ShardPlan, FY_padded: torch.Tensor, group=None, timer: Optional[CommTimer] = None):
        self.plan, self.FY, self.group = plan, FY_padded, group
        self.parts = len(self.groups_of(plan))
        self.stream = torch.cuda.Stream(device=FY_padded.device)
        self.events = [torch.cuda.Event() for _ in range(plan.world)]
        self.timer = timer or CommTimer(False)
        self.local = False

    def order(self) -> List[int]:
        r, w = self.plan.rank, self.plan.world
        return [r] + [c for c in range(w) if c != r]

    @staticmethod
    def groups_of(plan) -> List[List[int]]:
        r, w = plan.rank, plan.world
        return [g for g in ([r], list(range(0, r)), list(range(r + 1, w))) if g]

    def groups(self) -> List[List[int]]:
        """Consecutive shards that are swept together: the own shard first (resident), then the shards before it and the shards after it (contiguous row
        ranges of the padded layout; 8 single-round sweeps of 1/8 of the set cost 15-25 % more than 3 sweeps — ramp-up, tail and partial lists per sweep)."""
        r, w = self.plan.rank, self.plan.world
        return [g for g in ([r], list(range(0, r)), list(range(r + 1, w))) if g]

    def chunk_rows(self, c: int) -> Tuple[int, int, int]:
        return self.plan.style_chunk_rows(c)

    def start(self) -> None:
        """Enqueue this iteration's broadcasts (all ranks call it at the same point of the step)."""
        self.stream.wait_stream(torch.cuda.current_stream())        # last iteration's backward still reads the rows
        if self.local:                                              # (measurement aid: the rows of the last real delivery stay where they are)
            with torch.cuda.stream(self.stream):
                for c in range(self.plan.world):
                    self.events[c].record()
            return
        with torch.cuda.stream(self.stream), self.timer("style_gather_ms", _nbytes(self.FY)):
            for c in range(self.plan.world):
                m0, _, m1 = self.plan.style_chunk_rows(c)
                dist.broadcast(self.FY[m0:m1], src=dist.get_global_rank(self.group, c) if self.group is not None else c, group=self.group)
                self.events[c].record()

    def wait(self, c: int) -> None:
        if c != self.plan.rank:
            torch.cuda.current_stream().wait_event(self.events[c])


def tile_row_costs(records: torch.Tensor, tile_rows: int, tile_cols: int, sample: int = 8) -> torch.Tensor:
    """Approximate number of tile instances per tile row, from (a sample of) the splat records: every visible splat touches the tile rows
    [floor((py - r)/16), floor((py + r)/16)] and about 2r/16 + 1 tiles in each.  Difference array + prefix sum, on the device.  The two histograms
    are weight-vector x one-hot products (a [n, rows+1] comparison mask and a GEMV), NOT index_add_: 10^5 float atomics onto ~50 addresses serialise
    (1.9 ms per call measured — more than the whole render leg of a rank at 8 GPUs)."""
    rec = records[::sample]
    r = rec[:, 3].contiguous().view(torch.int32).float()
    vis = r > 0
    px, py = rec[:, 0], rec[:, 1]
    r0 = ((py - r) / 16).floor().clamp_(0, tile_rows - 1)
    r1 = ((py + r) / 16).floor().clamp_(0, tile_rows - 1) + 1.0
    # integer-valued weights: their float sums are exact (< 2^24) in any order, so every rank gets the SAME costs from the same records
    w = (((px + r) / 16).floor().clamp_(0, tile_cols - 1) - ((px - r) / 16).floor().clamp_(0, tile_cols - 1) + 1.0) * vis
    bins = torch.arange(tile_rows + 1, dtype=torch.float32, device=records.device)
    diff = w @ (r0[:, None] == bins[None, :]).float() - w @ (r1[:, None] == bins[None, :]).float()
    return diff.cumsum(0)[:tile_rows]


# ------------------------------------------------------------------------------------------------ the sharded loop
class DistributedStyleLoop:
    """GaussianStyleLoop's iteration with the Gaussians of one scene sharded over the ranks of the default process group."""

    def __init__(self, base_loop, plan: ShardPlan, style_shard: torch.Tensor, style: str = "cached", balance_bands: bool = True, render_group=None,
                 balance_every: int = 8, style_group=None, cu_reserve="auto"):
        assert style in ("cached", "inloop", "inloop_chunked")
        self.b = b = base_loop                # a GaussianStyleLoop over the LOCAL rows (owns params, Adam state, buffers); lists hold scene-wide indices
        self.plan = plan
        dev = b.pc._xyz.device
        assert b.N == plan.P and b.P_total == plan.P_total and b.row0 == plan.gauss_range[0]
        self.rg = render_group                # process group of the render leg's collectives (None: the default group)
        self.comm = CommTimer(enabled=bool(b.timers), cuda=dev.type == "cuda")          # per-collective events only when the loop is timed (bench.py)
        b.exchange = GeometryExchange(None, self.comm)
        if style == "inloop" and b.gemm == "fp8":
            style = "inloop_chunked"          # (the fp8 prepare computes its own norms of the quantised rows: padding rows would become candidates)
        # ---- style rows in the padded layout (every shard tile-aligned); gathered once here, again every iteration when style == "inloop"
        s0, s1 = plan.style_ranges[plan.rank]
        assert style_shard.shape[0] == s1 - s0
        self.style_mode = style
        self.style_stream = None
        self._local = False
        if style == "cached":
            # one all-gather at setup, padding stripped: the style set is then an ordinary [M, d] matrix and the sweep is the single-GPU one (all bounds)
            b.FY = all_gather_ragged_rows(style_shard.contiguous(), [e - s for s, e in plan.style_ranges])
            b.M = b.M_true = plan.M
            b.bind_xnorm()                     # the kNN workspace is sized by the style set: the loop's |x|^2 buffer is a view of it
        else:
            # padded layout (every shard starts on a 64-row tile boundary; the padding rows score -inf in the chunked sweep), re-gathered every iteration
            pad = torch.zeros(plan.Mc_pad, style_shard.shape[1], dtype=style_shard.dtype, device=dev)
            pad[: s1 - s0] = style_shard
            FY = torch.zeros(plan.M_pad, style_shard.shape[1], dtype=style_shard.dtype, device=dev)
            FY[plan.rank * plan.Mc_pad:(plan.rank + 1) * plan.Mc_pad] = pad          # the own shard is resident; the others arrive by broadcast every iteration
            b.FY, b.M, b.M_true = FY, plan.M_pad, plan.M
            # the in-loop broadcasts run on their OWN communicator when one is given: on the default group they would be queued, in issue order, ahead of the
            # positions all-gather that gates featurisation and the sweep of the own shard — the very work they are meant to hide under
            self.style_stream = StyleStream(plan, FY, style_group, self.comm)
            if style == "inloop_chunked":
                b.style_stream = self.style_stream          # the sweep itself is cut per arriving shard (w3d_knn_chunk_*, motion bound, unpruned)
            else:
                # round 6: the shards are re-delivered while the rows are featurised / the render leg runs; bound, plan and ONE pruned emit sweep follow (see the module text)
                b.style_token = ("inloop", plan.M_pad, 0)
                b.style_wait = self._style_arrived
                b.style_external_norms = True
                # tile geometry: a padding row stands in as its shard's last real row (zeros would inflate the radius of every shard's last tile)
                gi = torch.arange(plan.M_pad, device=dev)
                for c in range(plan.world):
                    m0, mv, m1 = plan.style_chunk_rows(c)
                    if m1 > mv:
                        gi[mv:m1] = max(mv - 1, m0)
                b.style_geometry_index = gi if plan.M_pad != plan.M else None
            b.bind_xnorm()
        # CUs the rank's pruned sweep leaves to the render leg.  The persistent sweep holds every CU it takes for its whole duration; a rank's image band is ~2500 / world
        # tile workgroups, each bound by the latency of its own depth-sorted list — at 8 ranks the band's blend kernels take 0.33 + 0.3 ms on 256 CUs and hardly more
        # on 32.  "auto": 32 CUs once the rank's sweep has fewer than 768 row blocks (world >= 3 at 1M Gaussians; such sweeps are cut in slices — all of them since round 4's emit sweep,
        # before that only those of at most 255 row blocks, 256 ... 767 ran one workgroup per block), else 0 — measured per rank (compute only,
        # scripts/rank_sim.py, medians): 8 ranks 3.00 -> 2.60 ms per step, 4 ranks 4.36 -> 4.18, 2 ranks 7.45 -> 7.84 with 32 reserved (so: 0 there).
        self.cu_reserve = (32 if (b.N + 511) // 512 < 768 else 0) if cu_reserve == "auto" else int(cu_reserve)
        b.cu_reserve = self.cu_reserve
        self.loss = b.loss
        self.last_R = 0
        self.balance = balance_bands
        self.balance_every = max(1, int(balance_every))   # the bands are re-cut from fresh costs every so many iterations (the scene moves slowly)
        self._iters = 0
        # ---- persistent buffers of the render leg
        H, W, Pt = plan.H, plan.W, plan.P_total
        L = lib()
        self.records_all = torch.empty(Pt, SPLAT, dtype=torch.float32, device=dev)
        self.accum_all = torch.zeros(Pt, SPLAT, dtype=torch.float32, device=dev)
        self.img4 = torch.zeros(4, H, W, dtype=torch.float32, device=dev)         # colour (3) + depth (1): ONE all-reduce
        self.band_scratch = torch.empty(L.w3d_raster_band_scratch_bytes(Pt), dtype=torch.uint8, device=dev)
        self.img_state = torch.empty(L.w3d_raster_image_bytes(H, W), dtype=torch.uint8, device=dev)
        self.binning = None
        self.cost_host = torch.zeros(plan.tile_rows, dtype=torch.float32).pin_memory() if dev.type == "cuda" else torch.zeros(plan.tile_rows)
        self.cost_ev = torch.cuda.Event()
        self.cost_pending = False

        def grow(_u, nbytes):                                       # binning buffer: kept, regrown only when the band's tile-instance count outgrows it.
            # Twice the need: the band changes with every re-cut and a fresh device allocation costs tens of ms (hipMalloc: 38 ms measured) — one of
            # those inside 20 timed steps of 4 ms would be half of a benchmark
            if self.binning is None or self.binning.numel() < nbytes:
                self.binning = torch.empty(max(int(nbytes * 2), 256 << 20), dtype=torch.uint8, device=dev)   # 256 MB floor: more than the whole image's instances need at C3
            return self.binning.data_ptr()
        self._cb = _lib.ALLOC_FN(grow)

    @contextmanager
    def local_collectives(self):
        """Inside: every collective of the step is replaced by a device-local stand-in on the same buffers (own block copied into place, no reduction, no broadcast).
        The step then computes on stale / partial data — a MEASUREMENT aid for after a timed region (bench.py: `comm.compute_only_ms_per_step`), never for results."""
        self._local = self.b.exchange.local = True
        if self.style_stream is not None:
            self.style_stream.local = True
        try:
            yield
        finally:
            self.cost_pending = False                  # (a cost read enqueued before the stand-ins is not used after them)
            self._local = self.b.exchange.local = False
            if self.style_stream is not None:
                self.style_stream.local = False

    def _style_arrived(self):
        """GaussianStyleLoop.style_wait of the in-loop gather: the main stream waits shard by shard for this iteration's broadcasts and refreshes the shard's
        -0.5 |y|^2 in the kNN workspace (w3d_knn_chunk_norms: padding rows -inf — they can never be candidates; 2 B x d per row read: 0.1 ms at M = 1e6)."""
        b, st = self.b, self.style_stream
        for c in st.order():
            st.wait(c)
            m0, mv, m1 = st.chunk_rows(c)
            ops.raw_knn_chunk_norms(b.FX, b.FY, b.k, m0, mv, m1, ws=b.kws)

    def set_style_shard(self, style_shard: torch.Tensor) -> None:
        """Replace this rank's shard of the style set (all ranks call it in the same iteration): the per-style-set caches of the loop (plan, bounds, neighbour sums, warm
        start) are keyed on the token this bumps, so the next sweep is cold by itself."""
        plan, b = self.plan, self.b
        s0, s1 = plan.style_ranges[plan.rank]
        assert style_shard.shape[0] == s1 - s0 and self.style_mode != "cached"
        m0 = plan.rank * plan.Mc_pad
        b.FY[m0:m0 + (s1 - s0)] = style_shard
        if b.style_token is not None:
            b.style_token = b.style_token[:2] + (b.style_token[2] + 1,)

    # bench.py / tests read these through the wrapper
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

    # ---- render leg ---------------------------------------------------------------------------------------------------------------------------
    def _rin(self, xyz, cam, offs):
        """W3dRasterIn of the LOCAL Gaussians on the loop's persistent activation buffers (split SH pointers: no concatenation)."""
        b = self.b
        pc, N = b.pc, b.N
        dev = xyz.device
        mats = getattr(cam, "_w3d_mats", None)
        if mats is None:                                             # the camera's matrices are transposed VIEWS (scene/cameras.py:54-57): contiguous copies, once
            mats = cam._w3d_mats = tuple(t.detach().to(dev).float().contiguous() for t in (cam.world_view_transform, cam.full_proj_transform, cam.camera_center))
        view, proj, campos = mats
        return _lib.RasterIn(N, pc.active_sh_degree, 1 + pc._features_rest.shape[1], b.H, b.W, ptr(b.bg), ptr(xyz), ptr(pc._features_dc.data), None, ptr(b.act_o),
                             ptr(b.act_s), ptr(b.act_q), None, ptr(view), ptr(proj), ptr(campos), ptr(offs),
                             1.0, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), 0, ptr(pc._features_rest.data))

    def _buffers(self):
        b = self.b
        if getattr(self, "_rb", None) is None:
            N, dev = b.N, b.pc._xyz.device
            z = torch.empty(N * 32, dtype=torch.float32, device=dev)            # g_m2 3 | g_col 3 | g_op 1 | g_cov 6 | g_sc 3 | g_rot 4 | accum 12
            cuts = [0, 3, 6, 7, 13, 16, 20, 32]
            self._rb = dict(raw=z, parts=[z[N * a:N * c] for a, c in zip(cuts[:-1], cuts[1:])],
                            geom=torch.empty(lib().w3d_raster_geom_bytes(N), dtype=torch.uint8, device=dev), radii=torch.empty(N, dtype=torch.int32, device=dev))
            if len(b.cams) > 1:                                                   # V > 1: per-view outputs are summed (the per-Gaussian backward writes, it does not add)
                self._rb["sum"] = torch.empty_like(z)
                self._rb["view_g"] = [torch.empty_like(t) for t in (b.g_xyz_render, b.grads[1], b.grads[2])]
        return self._rb

    def _render_view_forward(self, xyz, cam, depth_target, wv, first: bool):
        """Project own Gaussians, exchange records, blend own band, combine the bands, image losses (weights scaled by wv = 1/V)."""
        b, plan = self.b, self.plan
        L = lib(); s = stream_ptr()
        rb = self._buffers()
        offs = b._offsets()                                       # same seed on every rank -> identical offsets
        rin = self._rin(xyz, cam, offs)
        check(L.w3d_raster_preprocess(C.byref(rin), ptr(rb["geom"]), ptr(rb["radii"]), s), "w3d_raster_preprocess")
        records = rb["geom"][: b.N * SPLAT * 4].view(torch.float32).view(b.N, SPLAT)
        if self._local:
            self.records_all[plan.gauss_range[0]:plan.gauss_range[1]].copy_(records)
        else:
            with self.comm("splat_all_gather_ms", _nbytes(self.records_all)):
                all_gather_rows(records, self.rg, out=self.records_all)             # RCCL all-gather over xGMI: 48 B x P_total
        if self.balance and first and not self._local:                      # (local stand-ins: every rank holds different records — no re-cut from them, the bands stay as they are)
            if self._iters % self.balance_every == 0 or self._iters == 1:      # iteration 1 picks up the costs measured in iteration 0
                self._recut_bands()
            self._iters += 1
        self.img4.zero_()
        n = C.c_int(0)
        band = plan.band
        check(L.w3d_raster_blend_forward(ptr(self.records_all), plan.P_total, b.H, b.W, band[0], band[1], ptr(b.bg), ptr(offs), ptr(self.band_scratch), ptr(self.img_state),
                                         self._cb, None, ptr(self.img4[:3]), ptr(self.img4[3]), C.byref(n), s), "w3d_raster_blend_forward")
        self.last_R += n.value
        if not self._local:
            with self.comm("image_all_reduce_ms", _nbytes(self.img4)):
                dist.all_reduce(self.img4, group=self.rg)                            # bands -> the full image on every rank (zero outside a band: exact)
        ops.raw_mse_loss(self.img4[3], depth_target, b.w_depth * wv, b.loss[1:2], b.g_depth, accumulate=False)
        ops.raw_tv_loss(self.img4[:3], b.w_tv * wv, b.loss[2:3], b.g_color, accumulate=False)
        return rin, offs, n.value, band

    def _render_view_backward(self, rin, offs, R, band, gout):
        b, plan = self.b, self.plan
        L = lib(); s = stream_ptr()
        rb = self._buffers()
        self.accum_all.zero_()
        check(L.w3d_raster_blend_backward(ptr(self.records_all), plan.P_total, b.H, b.W, band[0], band[1], ptr(b.bg), ptr(offs), ptr(self.img_state), ptr(self.binning),
                                          int(R), ptr(b.g_color), ptr(b.g_depth), ptr(self.accum_all), s), "w3d_raster_blend_backward")
        accum_own = rb["parts"][6].view(b.N, SPLAT)
        if self._local:
            accum_own.copy_(self.accum_all[plan.gauss_range[0]:plan.gauss_range[1]])
        else:
            with self.comm("accum_reduce_scatter_ms", _nbytes(self.accum_all)):
                reduce_scatter_rows(self.accum_all, self.rg, out=accum_own)          # sums over the ranks' bands, to the owners: 48 B x P_total
        check(L.w3d_raster_preprocess_backward(C.byref(rin), ptr(rb["radii"]), ptr(rb["geom"]), C.byref(gout), s), "w3d_raster_preprocess_backward")

    def _recut_bands(self):
        """Bands for THIS iteration from the per-tile-row costs of the previous one (fetched through pinned memory; the wait is on an event that
        completed an iteration ago), then enqueue this iteration's costs.  Every rank holds the same records, hence the same costs and the same cut."""
        if self.cost_pending:
            self.cost_ev.synchronize()
            self.plan.set_bands(self.cost_host.tolist())
        cost = tile_row_costs(self.records_all, self.plan.tile_rows, self.plan.tile_cols)
        self.cost_host.copy_(cost, non_blocking=True)
        self.cost_ev.record()
        self.cost_pending = True

    def _render_forward(self, xyz):
        """Forward of every view (V = 1: the state the backward needs is kept; V > 1: forward + backward per view, gradients summed)."""
        b = self.b
        pc, N = b.pc, b.N
        L = lib(); s = stream_ptr()
        rb = self._buffers()
        check(L.w3d_activations_fwd(ptr(pc._scaling.data), ptr(pc._rotation.data), ptr(pc._opacity.data), N, ptr(b.act_s), ptr(b.act_q), ptr(b.act_o), s))
        self.last_R = 0
        V = len(b.cams)
        if V == 1:
            self._fwd_state = self._render_view_forward(xyz, b.cams[0], b.depth_targets[0], 1.0, True)
            rb["raw"].zero_()
            return
        g_m2, g_col, g_op, g_cov, g_sc, g_rot, accum = rb["parts"]
        rb["sum"].zero_()
        tg = rb["view_g"]
        for v, (cam, tgt) in enumerate(zip(b.cams, b.depth_targets)):
            st = self._render_view_forward(xyz, cam, tgt, 1.0 / V, v == 0)
            rb["raw"].zero_()
            for t in tg:
                t.zero_()
            gout = RasterGradOut(ptr(g_m2), ptr(g_col), ptr(g_op), ptr(tg[0]), ptr(g_cov), ptr(tg[1]), ptr(g_sc), ptr(g_rot), ptr(accum), ptr(tg[2]))
            self._render_view_backward(*st, gout)
            rb["sum"].add_(rb["raw"])
            for dst, t in zip((b.g_xyz_render, b.grads[1], b.grads[2]), tg):
                dst.add_(t)                                           # zeroed by the previous Adam step
        rb["raw"].copy_(rb["sum"])

    def _render_backward(self, xyz):
        b = self.b
        pc, N = b.pc, b.N
        L = lib(); s = stream_ptr()
        rb = self._buffers()
        g_m2, g_col, g_op, g_cov, g_sc, g_rot, accum = rb["parts"]
        if len(b.cams) == 1:
            gout = RasterGradOut(ptr(g_m2), ptr(g_col), ptr(g_op), ptr(b.g_xyz_render), ptr(g_cov), ptr(b.grads[1]), ptr(g_sc), ptr(g_rot), ptr(accum), ptr(b.grads[2]))
            self._render_view_backward(*self._fwd_state, gout)
        check(L.w3d_activations_bwd(ptr(pc._scaling.data), ptr(pc._rotation.data), ptr(pc._opacity.data), N, ptr(g_sc), ptr(g_rot), ptr(g_op),
                                    ptr(b.grads[4]), ptr(b.grads[5]), ptr(b.grads[3]), s))

    # ---- one iteration ------------------------------------------------------------------------------------------------------------------------
    def step(self):
        """Two HIP streams per rank, as on one GPU (GaussianStyleLoop.step): the geometry leg on the main stream — positions all-gather, featurisation,
        [style broadcasts under] the distance-GEMM, fused backward, dL/dxyz reduce-scatter; the render leg on the high-priority side stream — forward with
        its records all-gather and image all-reduce, backward (accumulator reduce-scatter), then Adam of the five groups only it feeds, all from t = 0 and
        NOT gated on the GEMM (round 2 held the backward until the GEMM's end; per rank, compute only: 7.95 -> 7.45 ms at 2 ranks, 4.80 -> 4.36 at 4 without the
        gate).  On short sweeps the GEMM leaves `cu_reserve` CUs to this leg (constructor).  The legs meet in the Adam step of the positions."""
        _lib.clear_stale_hip_error()
        b = self.b
        xyz = b.pc._xyz.data
        main = torch.cuda.current_stream()
        b.loss.zero_()
        b.side.wait_stream(main)                                   # the last Adam step
        if self.style_mode != "cached":
            self.style_stream.start()                              # world broadcasts on their own stream; the bound kernel (or, chunked, the sweep) waits shard by shard
        b._geometry_leg(xyz)                                       # enqueued first (asynchronous); the host then blocks on the band's tile count below
        with torch.cuda.stream(b.side), b._stage("render_fwd"):
            self._render_forward(xyz)
        b.t += 1
        b.lrs[0] = float(b.xyz_sched(b.t))
        groups = [(p.data, g, m, v, lr) for p, g, m, v, lr in zip(b.params, b.grads, b.m, b.v, b.lrs)]
        with torch.cuda.stream(b.side):
            with b._stage("render_bwd"):
                self._render_backward(xyz)
            with b._stage("adam_render_groups"):
                ops.raw_adam_step(groups[1:], b.t, 0.9, 0.999, 1e-15, zero_grad=True)
        main.wait_stream(b.side)
        with b._stage("adam_xyz"):
            ops.raw_adam_step([groups[0] + (b.g_xyz_render,)], b.t, 0.9, 0.999, 1e-15, zero_grad=True)      # geometry-leg + render-leg dL/dxyz summed in the step


# ------------------------------------------------------------------------------------------------ staged rasteriser calls (utility; allocates per call)
class BandRaster:
    """Stages A-D of include/wast3d_hip.h on one device with per-call buffers — for tests and one-off use; DistributedStyleLoop drives the same
    entry points on persistent buffers."""

    def __init__(self, H: int, W: int, device):
        self.H, self.W, self.dev = H, W, device
        self.img = torch.empty(lib().w3d_raster_image_bytes(H, W), dtype=torch.uint8, device=device)
        self._keep = None

    def preprocess(self, rin_args):
        """rin_args: the tensors/scalars of rasterize_gaussians_fwd for the LOCAL Gaussians. Returns (records view [P,12], radii, geom)."""
        from . import diff_gaussian_rasterization as dgr
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


def make_synthetic_c4(n_total=1_000_000, M=65536, d=256, k=20, H=800, W=800, timers=False, style="cached", lists="knn", scaling="strong", balance_bands=True,
                      render_group="own", **kw):
    """The C3 scene sharded over the ranks of the default group.
    scaling = "strong" (default): ONE scene of n_total Gaussians (every rank builds the same scene from the same seeds and keeps its rows; the lists
    reach across shard boundaries);  "weak": every rank brings n_total Gaussians of its own (seeds offset by rank; lists stay inside the shard) — an
    n_total x world scene in the one view.  The M style rows are sharded M/world per rank either way."""
    from .style_loop import GaussianStyleLoop, make_synthetic_c3
    from .gaussian_model import GaussianModel
    world, rank = dist.get_world_size(), dist.get_rank()
    if scaling == "strong":
        if n_total % world:
            raise _lib.W3dError(f"strong scaling needs the Gaussian count ({n_total}) to be a multiple of the world size ({world})")
        P = n_total // world
        plan = ShardPlan(world, rank, P, M, H, W)
        whole = make_synthetic_c3(N=n_total, M=M, d=d, k=k, H=H, W=W, timers=False, lists=lists, **kw)      # same seeds on every rank: the same scene
        lo, hi = plan.gauss_range
        pc = GaussianModel.synthetic(P, seed=0, device=whole.pc._xyz.device)
        pc.spatial_lr_scale = 1.0
        for name in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"):
            getattr(pc, name).data = getattr(whole.pc, name).data[lo:hi].clone()
        idx = whole.idx[lo:hi].clone()
        s0, s1 = plan.style_ranges[rank]
        shard = whole.FY[s0:s1].clone()
        gemm = kw.get("gemm", "same")
        base = GaussianStyleLoop(pc, idx, whole.FY[: max(k, 1)].clone(), whole.cams if len(whole.cams) > 1 else whole.cams[0], torch.stack(whole.depth_targets, 0) if len(whole.cams) > 1 else whole.depth_targets[0],
                                 whole.bg, k=k, sampling_seed=21, timers=timers, gemm=gemm, feat_center=whole.center, row0=lo, n_points_total=n_total,
                                 warm=kw.get("warm", "auto"), order_style=False, w_nn=whole.w_nn)          # whole.FY is in tree order already (the shards are slices of it)
        base.style_perm = whole.style_perm
        base.style_xyz, base.style_idx = whole.style_xyz, whole.style_idx
        del whole
        torch.cuda.empty_cache()
    else:
        plan = ShardPlan(world, rank, n_total, M, H, W)
        s0, s1 = plan.style_ranges[rank]
        # every rank's own Gaussians and its own slice of the style set; the lists are local to the shard and are shifted to scene-wide row numbers
        mine = make_synthetic_c3(N=n_total, M=max(s1 - s0, k), d=d, k=k, H=H, W=W, seeds=(20 + 100 * rank, 21, 22 + 100 * rank, 23 + 100 * rank), timers=False, lists=lists,
                                 center=False, **kw)
        shard = mine.FY[: s1 - s0].clone()
        lo = rank * n_total
        base = GaussianStyleLoop(mine.pc, mine.idx + lo, mine.FY, mine.cams if len(mine.cams) > 1 else mine.cams[0], torch.stack(mine.depth_targets, 0) if len(mine.cams) > 1 else mine.depth_targets[0],
                                 mine.bg, k=k, sampling_seed=21, timers=timers, gemm=kw.get("gemm", "same"), row0=lo, n_points_total=world * n_total, warm=kw.get("warm", "auto"),
                                 order_style=False, w_nn=mine.w_nn)
        for t in base.depth_targets:                               # the depth targets must be the same on every rank: take rank 0's
            dist.broadcast(t, src=0)
        del mine
    rg = dist.new_group(backend=dist.get_backend()) if (render_group == "own" and world > 1) else None
    sg = dist.new_group(backend=dist.get_backend()) if (style != "cached" and world > 1) else None      # every rank takes this branch alike: collective-safe
    return DistributedStyleLoop(base, plan, shard, style=style, balance_bands=balance_bands, render_group=rg, style_group=sg)
