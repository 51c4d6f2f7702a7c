#!/usr/bin/env python3
"""bench.py — style-loop iterations/s on the BASELINE C3 workload (1M Gaussians, 256-d features, M = 65 536 style rows, k = 20, one 800x800
view), one JSON line on rank 0.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one full iteration of wast3d_amd.style_loop.GaussianStyleLoop (featurise -> MFMA kNN match -> NN loss fwd/bwd -> descriptor scatter ->
render fwd -> depth-MSE + TV -> render bwd -> fused Adam on 59 floats/Gaussian).  Inputs are synthetic (SURVEY §8d C3 seeds) and resident in HBM
before the timed region.
N > 1 = STRONG scaling by default: the SAME 1M-Gaussian scene, 1M/N Gaussians per rank (wast3d_amd/distributed.py), `value` = iterations/s of
that one scene.  `--scaling weak` (N x 1M Gaussians, labelled as such) and `--config C4` (M = 1e6 style rows sharded, `--style-gather inloop|cached`)
are separate workloads.  Other configs the driver can time: `--config C2` (100k x 100k x 128 kNN only), `--config nb11` (the notebook's own shape:
189 circles x 4165 points).

Extra objects on the JSON line:
  roofline       the dominant kernel (fused distance-GEMM + top-k).  Its sweeps are PRUNED (exact): every timed step's own prune plan says how many
                 (row block, style tile) pairs that sweep executes, so achieved = sum_i swept_i * 2*N*M*d / sum_i t_i over the timed steps, t_i from HIP
                 events on the launch stream around that step's launch — numerator and denominator from the SAME steps (`per_step` lists both).
                 `unpruned` = three sweeps of the same rows with the plan switched off after the timed region: the data-independent figure on the full
                 2*N*M*d (what the reference's cdist + sort computes, NB11:1016-1021).
  run_1001       a WHOLE run of the notebook's loop length (1001 iterations, NB11:1070) on a fresh loop of the same workload: iterations/s and ms per step over the
                 whole run (mean / p50 / p90 / max), the prune plan's swept fraction and the distance-GEMM's executed-flop rate at five stations of the run.  The
                 20-step `value` is the run's fastest phase; this is what a user of the loop gets.  (`late`, round 3's 20 steps after 600 iterations: `--late 600`.)
  side_lines     BASELINE configs the driver's single command would otherwise never time: C2 (100k x 100k x 128 bf16, cold sweeps), the fp8 operand
                 path of C5 (e4m3 rows + E8M0 row scales on the workload's own rows) and the loop against M = 1e6 style rows (SURVEY 8d's stress shape = C4's
                 problem on one GPU), each with its own roofline fraction.
  rooflines      every stage of the iteration with its algorithmic flop / HBM bytes (formulas: DESIGN 3) against HIP-event durations of this
                 run: `ms` = the stage alone on the chip (two serialised iterations after the timed region; fraction of 2.5 PFLOP/s or 8 TB/s),
                 `ms_in_loop` = the same stage inside the timed loop, next to whatever the other stream runs.
  parity_vs_fp32 recall@k and relative NN-loss error of the match against the reference's fp32 path (oracle on unrounded rows), sampled rows.
  cpu_baseline   the oracle (oracle/, CPU restatement of the reference maths) timed on this box's host cores on a bounded sample of the same
                 workload and extrapolated linearly in N; reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E ~8 TB/s
MFMA_PEAK_TF = {"bf16": 2500.0, "f16": 2500.0, "fp8": 5000.0}


def cpu_baseline(loop, N, M, d, k, H, W, n_knn=12288):
    """Times the oracle on a bounded sample and extrapolates linearly in the number of Gaussians / content rows (every stage of the path is linear
    in N for fixed M, image and tile load).  The kNN sample is the oracle's fp32 match of `n_knn` of the workload's OWN content rows against the full
    style set — the same run yields the fp32-parity figures of the HIP match (tests/parity_fp32.py)."""
    import math
    import numpy as np
    from oracle import geometry_ref as O
    from oracle import raster_ref as RR
    from tests.parity_fp32 import fp32_parity
    from wast3d_amd.camera import Camera
    threads = torch.get_num_threads()
    rng = np.random.default_rng(0)
    n_geo, n_ras = 200000, 400000
    par = fp32_parity(loop, n_sample=n_knn, with_fp64=False)
    t_knn, n_knn = par.pop("seconds_oracle_knn"), par["rows"]
    # descriptors fwd + scatter bwd on n_geo rows
    X = rng.uniform(-1.3, 1.3, (n_geo, 3)).astype(np.float32)
    idx = rng.integers(0, n_geo, (n_geo, d + 1)); idx[:, 0] = np.arange(n_geo)
    tgt = np.zeros((n_geo, d), np.float32)
    t0 = time.perf_counter(); O.get_descriptors(X, idx); O.descriptor_mse(X, idx, tgt); t_desc = time.perf_counter() - t0
    # rasteriser fwd + bwd (C restatement, single thread) on n_ras Gaussians, full image
    cam = Camera.look_at_origin(0.6, 0.45, 4.03, 0.6911112070083618, W, H, device="cpu")
    means = rng.uniform(-1.3, 1.3, (n_ras, 3)).astype(np.float32)
    shs = np.zeros((n_ras, 16, 3), np.float32); shs[:, 0] = rng.uniform(-1.7, 1.7, (n_ras, 3))
    f = lambda t: t.numpy().astype(np.float32)
    r = RR.Raster(np.ones(3, np.float32), means, np.full((n_ras, 1), 0.1, np.float32), f(cam.world_view_transform), f(cam.full_proj_transform),
                  f(cam.camera_center), math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), H, W, -rng.random((H, W, 2)).astype(np.float32), shs=shs, sh_degree=3,
                  scales=np.full((n_ras, 3), 0.005, np.float32), rotations=np.tile(np.array([[1, 0, 0, 0]], np.float32), (n_ras, 1)))
    t0 = time.perf_counter(); c, dd, _ = r.forward(); r.backward(np.ones_like(c), np.ones_like(dd)); t_ras = time.perf_counter() - t0
    # Adam on 59 floats per Gaussian
    p = rng.standard_normal((n_ras, 59)).astype(np.float32); g = rng.standard_normal((n_ras, 59)).astype(np.float32)
    m = np.zeros_like(p); v = np.zeros_like(p)
    t0 = time.perf_counter(); O.adam_step(p, g, m, v, 1, 1e-3, eps=1e-15); t_adam = time.perf_counter() - t0
    t_iter = t_knn * (N / n_knn) + t_desc * (N / n_geo) + (t_ras + t_adam) * (N / n_ras)
    # per-leg thread counts: only the kNN leg (numpy's BLAS matmul inside the fp32 cdist) is multi-threaded; the other three legs are single-thread numpy / C
    legs = [{"leg": "kNN: fp32 cdist (matmul form) + k smallest", "threads": threads, "sample_rows": n_knn, "seconds": t_knn, "extrapolated_seconds": t_knn * (N / n_knn)},
            {"leg": "descriptors fwd + scatter bwd (numpy)", "threads": 1, "sample_rows": n_geo, "seconds": t_desc, "extrapolated_seconds": t_desc * (N / n_geo)},
            {"leg": "rasteriser fwd + bwd (oracle/raster_ref.c)", "threads": 1, "sample_rows": n_ras, "seconds": t_ras, "extrapolated_seconds": t_ras * (N / n_ras)},
            {"leg": "Adam, 59 floats per Gaussian (numpy)", "threads": 1, "sample_rows": n_ras, "seconds": t_adam, "extrapolated_seconds": t_adam * (N / n_ras)}]
    cpu_model = None
    try:
        cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        pass
    t_tot = sum(l["extrapolated_seconds"] for l in legs)
    base = {"value": 1.0 / t_iter, "unit": "iters/s", "cores": threads, "kind": "port", "extrapolated": True, "legs": legs, "cpu_model": cpu_model, "host_cores": os.cpu_count(),
            # time-weighted thread count over the legs: 0.91 of the iteration is the kNN leg on all BLAS threads, the rest single-threaded restatements
            "threads_effective": sum(l["threads"] * l["extrapolated_seconds"] for l in legs) / t_tot,
            "sample": (f"oracle timed on a bounded sample and extrapolated linearly to N={N} ({t_iter:.1f} s/iter): fp32 kNN of {n_knn} of the workload's content rows x {M} x {d} "
                       f"on {threads} BLAS threads ({t_knn:.2f}s); descriptors fwd+bwd {n_geo} rows, 1 thread ({t_desc:.2f}s); rasteriser fwd+bwd {n_ras} Gaussians @{W}x{H}, "
                       f"1 thread C ({t_ras:.2f}s); Adam {n_ras}x59, 1 thread ({t_adam:.3f}s).  `cores` = the kNN leg's threads; the other legs are single-threaded (see `legs`)")}
    return base, par


def knn_traffic(a, pmc_dir="r06_pmc"):
    """HBM-side bytes per LAUNCH of the kNN kernel from the committed PMC passes over bench.py itself (profiles/<pmc_dir>: separate --pmc runs of
    `bench.py --steps 12 --warmup 6`, scripts/traffic_pass.sh; FETCH_SIZE x2 = the gfx950 wide-read correction of MI355X_MICROARCH.md, + WRITE_SIZE, both in
    KiB; average over the last 8 launches of the pruned sweep's kernel = 8 steady iterations).  bench.py cannot run rocprofv3 on itself: this is NOT a measurement of
    the present run and is labelled with its source; None for any other workload or when the files are absent."""
    if (a.config, a.n, a.m, a.d, a.k, a.feats, a.gemm, a.style, a.lists) != ("C3", 1_000_000, 65536, 256, 20, "bf16", "same", "matched", "exact"):
        return None, None
    try:
        import csv
        def mean(path, name):
            # the PRUNED sweep's kernel only: knn_emit_kernel<D, fmt, LB = true, ...> records the cached lower bounds, which only the loop's planned sweeps do — the three
            # unpruned sweeps bench.py runs after the timed region are the LB = false instantiation of the same kernel (they used to be averaged in)
            import re
            pruned_kernel = re.compile(r"knn_emit_kernel<\d+, \d+, true,|knn_mfma_hot_kernel|knn_mfma_persistent_kernel")
            rows = [float(r["Counter_Value"]) for r in csv.DictReader(open(os.path.join(ROOT, "profiles", pmc_dir, path)))
                    if r["Counter_Name"] == name and pruned_kernel.search(r["Kernel_Name"])]
            rows = rows[-8:]
            return sum(rows) / len(rows)
        head = None
        try:
            head = open(os.path.join(ROOT, "profiles", pmc_dir, "HEAD")).read().strip()      # the commit the passes were taken at (scripts/collect_profile.py writes it)
        except OSError:
            pass
        return (2.0 * mean("fetch_size.csv", "FETCH_SIZE") * 1024.0 + mean("write_size.csv", "WRITE_SIZE") * 1024.0,
                f"profiles/{pmc_dir} (committed rocprofv3 --pmc passes of this command at commit {head or 'unrecorded'}, not this run: stale once the kernel changes)")
    except Exception:
        return None, None


def stage_rooflines(a, loop, st, world, R, alone):
    """Per-stage algorithmic work (per rank) against the duration measured in the loop.  Formulas: DESIGN 3 / SURVEY §8d."""
    N = a.n // world if a.scaling == "strong" else a.n
    P, K, d, k, M, HW = N, a.d + 1, a.d, a.k, a.m, a.res * a.res
    fb = 2                                                          # bytes per stored feature
    out = []

    def hbm(name, key, nbytes, note, gathered=0):
        # `ms` / achieved / frac: the stage ALONE on the chip (one serialised iteration after the timed region — kernel quality);
        # `ms_in_loop`: the same stage inside the timed loop, next to whatever the other stream runs (scheduling; stages of the render leg run in the
        # distance-GEMM's shadow and are slower there without costing step time)
        t = alone.get(key, st.get(key, 0.0))
        if t > 0:
            gbs = nbytes / (t * 1e-3) / 1e9
            out.append({"stage": name, "bound": "hbm", "algorithmic_bytes": nbytes, "ms": t, "ms_in_loop": st.get(key), "bound_ms": nbytes / (HBM_PEAK_GBS * 1e9) * 1e3, "achieved": gbs,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "bytes": note})
            if gathered:
                # row gathers out of a cache-resident table (the 32-MB style set): not HBM traffic, but the bytes the stage actually moves through L2 / the Infinity
                # Cache (MI355X_MICROARCH 'Indexed rows': 8.6 TB/s chip-wide for random rows of a 38-MB table, ~17-19 TB/s when they hit L2)
                out[-1]["gathered_row_bytes"] = gathered
                out[-1]["gather_rate_GBps"] = gathered / (t * 1e-3) / 1e9
    b_fwd, b_bwd = loop.stream_bytes_per_row()
    hbm("featurise (descriptor_fwd_window_kernel)", "featurise", N * b_fwd,
        "per row: index row 4K (or 2-B slots per entry + root under a covering slot plan) + feature row written 2d (+ 2d re-read when a bound uses the row motion) + own position 16 + norm 8")
    if "bound" in st:
        hbm("warm-start bound (" + loop._warm_mode + ")", "bound", N * (fb * d + 4 * k + 8) + (M * fb * d if loop._warm_mode == "prev" else 0),
            "own row 2d + previous list 4k + norm/bound 8 per row (+ every style row once, 2dM, for the previous-neighbour bound)",
            gathered=N * fb * d * (k if loop._warm_mode == "prev" else k // 2 if loop._warm_mode == "half" else 0))
    hbm("fused NN-loss backward + descriptor scatter (descriptor_bwd_window_kernel)", "fused_bwd", N * b_bwd + M * fb * d,
        "per row: index row 4K (or 2-B slots per entry + root) + own row 2d + nn idx/d2 8k + position 16 + gradient 12, + every style row once (2dM)", gathered=N * fb * d * k)
    hbm("rasteriser forward (activations, projection, binning, blend, image losses)", "render_fwd", P * (236 + 75) + R * (12 * 2 + 28) + HW * 32,
        "SURVEY 8d: P(236+75) + R(12*2*passes + 28), passes = 1 (per-tile LDS sort) + HW*32")
    hbm("rasteriser backward (blend backward, per-Gaussian chain, activations)", "render_bwd", R * (28 + 12) + HW * 20 + P * (40 + 472),
        "SURVEY 8d: R(28+12) + HW*20 + P(40 + 236 read + 236 written)")
    if "adam_render_groups" in st:
        hbm("Adam, 56 floats/Gaussian (adam_kernel alone; in the loop adam_lean_kernel, 32 VGPRs, UNDER the sweep: hidden, its in-loop time is not on the step's path)", "adam_render_groups", 28 * 56 * P,
            "28 B per float: p, g, m, v read; p, m, v written")
        hbm("Adam, xyz (adam_kernel, two gradient buffers)", "adam_xyz", 32 * 3 * P, "32 B per float")
    return out


def run_c2(a):
    """BASELINE config C2: 100k x 100k x 128 bf16 brute-force kNN match only (+ the loss head); cold sweeps (no loop to warm-start from)."""
    from wast3d_amd import ops, _lib
    g = torch.Generator().manual_seed(10)
    N = M = 100_000
    d, k = 128, a.k
    X = torch.randn(N, d, generator=g).to(torch.bfloat16).cuda()
    Y = torch.randn(M, d, generator=torch.Generator().manual_seed(11)).to(torch.bfloat16).cuda()
    idx = torch.empty(N, k, dtype=torch.int32, device="cuda"); d2 = torch.empty(N, k, device="cuda"); loss = torch.zeros(1, device="cuda")
    ws = ops.KnnWorkspace()
    def step():
        w = ops.raw_knn_prepare(X, Y, k, ws=ws)
        ev = ops.raw_knn_match_prepared(X, Y, k, idx, d2, w, timed=True, ws=ws)
        ops.raw_knn_loss(d2, 1.0 / (N * M), loss)
        return ev
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs = [step() for _ in range(a.steps)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms = sum(e0.elapsed_time(e1) for e0, e1 in evs) / len(evs)
    flop = 2.0 * N * M * d
    ach = flop / (kms * 1e-3) / 1e12
    # the same sweep WARM: bound = the current distances of the neighbours the previous sweep found (what every iteration of a loop on this shape after its first runs)
    idx2 = torch.empty_like(idx); d2b = torch.empty_like(d2)
    wevs = []
    for _ in range(max(3, a.steps)):
        w = ops.raw_knn_prepare(X, Y, k, prev_idx=idx, ws=ws)
        wevs.append(ops.raw_knn_match_prepared(X, Y, k, idx2, d2b, w, timed=True, ws=ws))
    torch.cuda.synchronize()
    wms = sum(e0.elapsed_time(e1) for e0, e1 in wevs) / len(wevs)
    assert torch.equal(idx2, idx)
    wach = flop / (wms * 1e-3) / 1e12
    # ... and through the round-4 EMIT selection (prune plan — on structureless rows it keeps every tile — + match_prepared(warm = 4): no k-lists in the sweep, candidates
    # selected by a second kernel; the events bracket sweep + select)
    pr = ops.KnnPrune(Y, meb_iters=10)
    eevs = []
    idx3 = torch.empty_like(idx); d2c = torch.empty_like(d2)
    for _ in range(max(3, a.steps)):
        ops.raw_knn_prepare(X, Y, k, prev_idx=idx, ws=ws)
        ops.raw_knn_prune_plan(X, k, pr, ws)
        eevs.append(ops.raw_knn_match_prepared(X, Y, k, idx3, d2c, 4, timed=True, ws=ws))
    torch.cuda.synchronize()
    ems = sum(e0.elapsed_time(e1) for e0, e1 in eevs) / len(eevs)
    assert torch.equal(idx3, idx)
    each = flop / (ems * 1e-3) / 1e12
    return {"metric": "kNN dist-GEMM TFLOP/s (config C2: 100k x 100k x 128 bf16, brute-force kNN feature match only)", "value": ach, "unit": "TFLOP/s", "n_gpus": 1, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16 (f32 accumulate)", "data": "synthetic",
            "config": {"workload": "C2: 100k Gaussians' 128-d feature rows vs 100k style rows, brute-force kNN feature match, bf16, cold sweeps", "N": N, "M": M, "feat_dim": d, "k": k},
            "roofline": {"kernel": "knn_emit_kernel<128, bf16, cold> + knn_select_kernel (sweeps without a warm-start bound: the rows' bounds are tracked in registers as the sweep goes)", "bound": "mfma", "achieved": ach, "peak": 2500.0, "unit": "TFLOP/s", "frac": ach / 2500.0, "traffic": None,
                         "launch_ms": kms, "bound_ms": flop / 2.5e15 * 1e3, "algorithmic_flop_per_step": flop, "launches_per_step": 1,
                         "note": "196 row blocks on 256 CUs, one item each (the cold sweep's regions are compacted by the wave that owns them: no slices): at most 0.77 of whatever the kernel "
                                 "sustains; any exact streaming selection without a bound admits ~k (1 + ln(M / k)) candidates per row — a twelfth of the tiles is swept twice (first for the "
                                 "bound only) so that 76 per row reach the candidate buffer instead of 240; round 4's list kernel: 0.20 of peak"},
            "warm_sweep": {"ms": wms, "achieved": wach, "peak": 2500.0, "unit": "TFLOP/s", "frac": wach / 2500.0, "lists_equal_cold_sweep": True,
                           "note": "bound from the previous sweep's neighbours (w3d_knn_prepare with prev_idx): the steady state of any loop on this shape; match_prepared(warm = 1), which "
                                   "since round 5 is the emit kernel in slices + select for 16-bit rows (rounds 1-4's list kernel: 2.87 ms = 0.36)"},
            "warm_emit_sweep": {"ms": ems, "achieved": each, "peak": 2500.0, "unit": "TFLOP/s", "frac": each / 2500.0, "lists_equal_cold_sweep": True,
                                "note": "the same warm sweep with the EMIT selection (match_prepared warm = 4 after a prune plan that keeps every tile of these structureless rows): "
                                        "sliced persistent launch + select kernel"},
            "sweeps_per_s": a.steps / dt}


def run_c1(a):
    """BASELINE config C1: 2 000 points, 64-d features F_X = X A, sliced-Wasserstein loss over L = 64 unit projections against a fixed style cloud, 50 Adam
    iterations (lr 1e-2).  The reference has no sliced-Wasserstein head (parity unpinned; known answer = 1-D optimal transport, tests); BASELINE.md §4 asks for this config
    on torch-CPU at full size — timed here next to the HIP head.  One step = project + sort + head (one launch), dL/dX + loss fold (one launch), fused Adam (one launch):
    <X A, theta_l> = <X, A theta_l>, so the constant feature map is folded into the projection directions once (d = 3 inside the kernels)."""
    import numpy as np
    from wast3d_amd import ops
    N, dF, L, iters = 2000, 64, 64, 50
    X0 = torch.rand(N, 3, generator=torch.Generator().manual_seed(0)) * 2.6 - 1.3
    A = torch.randn(3, dF, generator=torch.Generator().manual_seed(1))
    FY = torch.randn(N, dF, generator=torch.Generator().manual_seed(2)) * 1.5 + 0.5
    proj = torch.nn.functional.normalize(torch.randn(L, dF, generator=torch.Generator().manual_seed(3)), dim=1)

    # ---- torch-CPU: the formulation as a torch user writes it (autograd + torch.optim.Adam), all host threads
    threads = torch.get_num_threads()
    def cpu_run(n):
        Xc = torch.nn.Parameter(X0.clone()); opt = torch.optim.Adam([Xc], lr=1e-2)
        sy = torch.sort(FY @ proj.T, dim=0).values
        losses = []
        t0 = time.perf_counter()
        for _ in range(n):
            opt.zero_grad(set_to_none=True)
            sx = torch.sort((Xc @ A) @ proj.T, dim=0).values
            loss = ((sx - sy) ** 2).mean()
            loss.backward(); opt.step()
            losses.append(float(loss))
        return time.perf_counter() - t0, Xc.detach().clone(), losses
    # (all host threads — BASELINE.md §4 — AND one thread: on 2 000 x 64 tensors torch's intra-op parallelism over 128 threads costs more than it buys; the better of
    # the two is the baseline)
    reps_cpu = 3
    t_by_threads = {}
    for nt in sorted({threads, 1}, reverse=True):
        torch.set_num_threads(nt)
        cpu_run(5)
        t_by_threads[nt] = min(cpu_run(iters)[0] for _ in range(reps_cpu))
    best_threads = min(t_by_threads, key=t_by_threads.get)
    t_cpu = t_by_threads[best_threads]
    torch.set_num_threads(threads)
    _, X_cpu, l_cpu = cpu_run(iters)

    # ---- HIP: three launches per iteration on persistent buffers
    dev = "cuda"
    projX = (proj @ A.T).contiguous().to(dev)                          # [L,3] = (A theta_l)^T
    py = ops.sw_sorted_projections(FY.to(dev), proj.to(dev))          # the style side: once
    def hip_run(n, record=False):
        X = X0.clone().to(dev); g = torch.zeros_like(X); m = torch.zeros_like(X); v = torch.zeros_like(X)
        loss = torch.zeros(n + 1, device=dev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(1, n + 1):
            ops.raw_sw_features_fwd_bwd(X, projX, py, loss[t:t + 1], g)
            ops.raw_adam_step([(X, g, m, v, 1e-2)], t, 0.9, 0.999, 1e-8, zero_grad=False)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, X, loss[1:].tolist()
    for _ in range(max(1, a.warmup)):
        hip_run(iters)
    runs = max(1, a.steps)
    ts = [hip_run(iters)[0] for _ in range(runs)]
    t_hip = sum(ts) / len(ts)
    _, X_hip, l_hip = hip_run(iters)
    # kernel time of the head alone (HIP events on the launch stream)
    X = X0.clone().to(dev); g = torch.zeros_like(X); ls = torch.zeros(1, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.raw_sw_features_fwd_bwd(X, projX, py, ls, g); torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        ops.raw_sw_features_fwd_bwd(X, projX, py, ls, g)
    e1.record(); torch.cuda.synchronize()
    head_ms = e0.elapsed_time(e1) / 20
    dX = (X_hip.cpu() - X_cpu).abs()
    err99, err = float(dX.flatten().kthvalue(int(0.99 * dX.numel())).values), float(dX.max())
    loss_rel = max(abs(a_ - b_) / abs(b_) for a_, b_ in zip(l_hip, l_cpu))
    # algorithmic HBM bytes of the head per iteration: X 12 N + sorted style projections 4 L N + dL/dp written and re-read 8 L N + gradient 12 N
    nbytes = 24 * N + 12 * L * N
    gbs = nbytes / (head_ms * 1e-3) / 1e9
    return {"metric": "sliced-Wasserstein style loop iters/s (config C1: 2k points, 64-d feats, 64 projections, 50 Adam iters)", "value": iters / t_hip, "unit": "iters/s", "n_gpus": 1,
            "steps": runs, "warmup": a.warmup, "ms_per_step": t_hip / iters * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C1: 2k-point cloud, 64-d random feats F_X = X A, sliced-Wasserstein loss (64 unit projections), 50 Adam iters (lr 1e-2); SURVEY 8d seeds 0-3; one bench "
                                   "step = one whole 50-iteration run", "N": N, "feat_dim": dF, "projections": L, "adam_iters": iters,
                       "launches_per_iter": 3, "feature_map": "folded into the projection directions (<X A, theta> = <X, A theta>): the kernels see d = 3"},
            "roofline": {"kernel": "sw_project_sort_kernel<head> (one workgroup per projection: projections, u64 (key, index) pairs and the bitonic sort in LDS) + sw_feature_grad_kernel",
                         "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None, "launch_ms": head_ms,
                         "algorithmic_bytes": nbytes, "note": "64 workgroups of an LDS-resident sort on a 256-CU chip: a latency-bound plumbing config (1.6 MB of algorithmic bytes), "
                                                              "reported because BASELINE.md lists C1; the HBM roofline is not what limits it"},
            "parity": {"loss_trace_max_rel_diff_vs_torch_cpu": loss_rel, "tolerance_loss": 1e-5, "abs_diff_X_after_50_iters_p99": err99, "tolerance_X_p99": 2e-3,
                       "abs_diff_X_after_50_iters_max": err, "ok": loss_rel < 1e-5 and err99 < 2e-3, "loss_first_last_hip": [l_hip[0], l_hip[-1]],
                       "loss_first_last_torch_cpu": [l_cpu[0], l_cpu[-1]],
                       "note": "all 50 losses within the tolerance; Adam's update is lr * sign-like where |g| ~ eps, so a coordinate whose gradient is a rounding error of "
                               "either implementation can drift by up to lr per iteration: the maximum is reported, the 99th percentile is held to the tolerance"},
            "cpu_baseline": {"value": iters / t_cpu, "unit": "iters/s", "cores": best_threads, "kind": "port", "host_cores": os.cpu_count(),
                             "ms_per_run_by_threads": {str(k_): v_ * 1e3 for k_, v_ in t_by_threads.items()},
                             "sample": f"the full config (50 iterations, best of {reps_cpu} runs) as torch-CPU autograd + torch.optim.Adam, timed on {sorted(t_by_threads)} threads; "
                                       f"value = the faster ({best_threads} threads: {t_cpu * 1e3:.1f} ms per run)"}}


def run_nb11(a):
    """The reference's own workload shape (NB11 cell 32): ONE style patch of 4165 points fitted to 189 domain circles of 400-900 points, 2000/100-neighbour
    descriptor lists, k = 20.  The notebook runs the 189 problems one after the other (480 s for 189 x 1001 iterations = 394 circle-iterations/s on its GPU,
    BASELINE.md §2); here the circles are the segment dimension of every kernel, 64 at a time."""
    from wast3d_amd.geometry_loop import GeometryFit
    g = torch.Generator().manual_seed(0)
    n, circles, batch = 4165, 189, 64
    uv = torch.rand(n, 2, generator=g) * 2 - 1
    T = torch.stack([uv[:, 0], uv[:, 1], 0.3 * torch.sin(2.5 * uv[:, 0]) * torch.cos(1.5 * uv[:, 1])], 1).mul(0.2).cuda()
    doms = [(torch.randn(int(m), 3, generator=g) * 0.3 + torch.randn(3, generator=g)).cuda() for m in torch.randint(400, 900, (circles,), generator=g)]
    fits = [GeometryFit(T, doms[i:i + batch]) for i in range(0, circles, batch)]
    for f in fits:
        for _ in range(a.warmup):
            f.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in fits:
        for _ in range(a.steps):
            f.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"metric": "geometry-loop circle-iterations/s (NB11 shape: 189 circles x 4165 points, 2000/100-neighbour lists, k = 20)", "value": circles * a.steps / dt,
            "unit": "circle-iterations/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": circles * a.steps / dt / 394.0, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "nb11: one 4165-point style patch fitted to 189 domain circles (400-900 points), the notebook's literals; one step = one iteration of ALL circles",
                       "baseline": "394 circle-iterations/s = 189 x 1001 iterations in 480 s, the notebook's own tqdm output on its (unnamed) CUDA GPU (BASELINE.md §2)"}}


def time_unpruned_sweeps(loop, n=3):
    """The distance-GEMM + top-k of the loop's CURRENT rows with the prune plan switched off (bound = the current neighbours' distances, as an unpruned loop
    has it): ms per sweep over all its launches (main + split tail round) and whether the lists equal the loop's own (pruned) ones."""
    from wast3d_amd import ops
    idx = torch.empty_like(loop.nn_idx); d2 = torch.empty_like(loop.nn_d2)
    evs = []
    for _ in range(n):
        w = ops.raw_knn_prepare(loop.FX, loop.FY, loop.k, prev_idx=loop.nn_idx, ws=loop.kws, row_order=loop.row_order)
        evs.append(ops.raw_knn_match_prepared(loop.FX, loop.FY, loop.k, idx, d2, w, timed=True, ws=loop.kws, row_order=loop.row_order))
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b in evs]
    return sum(ts) / len(ts), bool(torch.equal(idx, loop.nn_idx))


def side_line_fp8(loop, n=3):
    """The fp8 operand path (BASELINE config C5) on this workload's own rows: e4m3 bytes + one E8M0 scale per row on v_mfma_scale_f32_32x32x64_f8f6f4.  One cold
    sweep for the neighbours, then n sweeps warm-started from them (the loop's regime), unpruned; rate on the full 2*N*M*d against the 5 PFLOP/s fp8 peak."""
    from wast3d_amd import ops
    qX, sX = ops.quantize_rows_fp8(loop.FX); qY, sY = ops.quantize_rows_fp8(loop.FY)
    idx, d2, ev0 = ops.raw_knn_match_fp8(qX, sX, qY, sY, loop.k, timed=True)
    evs = [ops.raw_knn_match_fp8(qX, sX, qY, sY, loop.k, idx=torch.empty_like(idx), dist2=torch.empty_like(d2), prev_idx=idx, timed=True)[2] for _ in range(n)]
    torch.cuda.synchronize()
    ms = sum(a.elapsed_time(b) for a, b in evs) / n
    flop = 2.0 * loop.N * loop.M * loop.d
    tf = flop / (ms * 1e-3) / 1e12
    out = {"line": "fp8 sweep (C5's operand choice) on the C3 rows: e4m3 + E8M0 row scales, warm-started from its own neighbours, unpruned", "ms_per_sweep": ms,
           "cold_sweep_ms": ev0[0].elapsed_time(ev0[1]), "flop_per_sweep": flop,
           "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_TF["fp8"], "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TF["fp8"],
                        "kernel": "knn_mfma_kernel<256,fp8> (the list-keeping sweep)"}}
    # Round 5: the same unpruned warm sweep through the EMIT kernel's fp8 form (a plan that keeps every tile, the same bound) and, as its bf16 counterpart, the bf16 rows
    # through the emit kernel on the same kind of plan: what the operand format buys once the selection no longer binds the sweep
    try:
        from wast3d_amd._lib import lib, ptr, stream_ptr, check
        N, M, d, k = loop.N, loop.M, loop.d, loop.k
        nblk, T = (N + 511) // 512, (M + 63) // 64
        order = torch.arange(T, dtype=torch.int32, device="cuda").repeat(nblk, 1).contiguous()
        ws = ops.KnnWorkspace()
        oi = torch.empty_like(idx); od = torch.empty_like(d2)
        evs = []
        for _ in range(n):
            ops.raw_knn_fp8_phase(qX, sX, qY, sY, k, 1, 1, prev_idx=idx, ws=ws)
            check(lib().w3d_knn_plan_order(N, M, d, k, ptr(order), 0, ptr(ws.t), ws.t.numel(), stream_ptr()), "w3d_knn_plan_order")
            evs.append(ops.raw_knn_fp8_phase(qX, sX, qY, sY, k, 2, 4, idx=oi, dist2=od, timed=True, ws=ws))
        torch.cuda.synchronize()
        ems = sum(a.elapsed_time(b) for a, b in evs) / n
        etf = flop / (ems * 1e-3) / 1e12
        out["emit_sweep"] = {"ms_per_sweep": ems, "achieved": etf, "peak": MFMA_PEAK_TF["fp8"], "unit": "TFLOP/s", "frac": etf / MFMA_PEAK_TF["fp8"],
                             "lists_equal_list_sweep": bool(torch.equal(oi, idx)), "kernel": "knn_emit_kernel<256,fp8> + knn_select_kernel, every tile swept"}
        bi, bd = ops.raw_knn_match(loop.FX, loop.FY, k)
        wsb = ops.KnnWorkspace(); evs = []
        o2 = torch.empty_like(bi); e2 = torch.empty_like(bd)
        for _ in range(n):
            ops.raw_knn_prepare(loop.FX, loop.FY, k, prev_idx=bi, ws=wsb)
            check(lib().w3d_knn_plan_order(N, M, d, k, ptr(order), 0, ptr(wsb.t), wsb.t.numel(), stream_ptr()), "w3d_knn_plan_order")
            evs.append(ops.raw_knn_match_prepared(loop.FX, loop.FY, k, o2, e2, 4, timed=True, ws=wsb))
        torch.cuda.synchronize()
        bms = sum(a.elapsed_time(b) for a, b in evs) / n
        btf = flop / (bms * 1e-3) / 1e12
        out["emit_sweep_16bit_rows"] = {"ms_per_sweep": bms, "achieved": btf, "peak": MFMA_PEAK_TF["bf16"], "unit": "TFLOP/s", "frac": btf / MFMA_PEAK_TF["bf16"],
                                        "lists_equal_cold_sweep": bool(torch.equal(o2, bi)), "kernel": "knn_emit_kernel<256,16-bit> + knn_select_kernel, every tile swept"}
    except Exception as e:                                                # a side line never takes the headline down
        out["emit_sweep"] = {"error": repr(e)}
    return out


def side_line_fp8_loop(a, steps=20, warmup=8):
    """BASELINE config C5's operand choice in the LOOP (round 5): the C3 scene with f16 feature rows and the fp8 distance-GEMM — pruned emit sweeps with cached lower bounds
    on the quantised problem, exactly the 16-bit loop's machinery — next to the same loop with the f16 distance-GEMM.  ms per step, sweep ms, swept fraction, executed
    flop against each format's MFMA peak."""
    from wast3d_amd.style_loop import make_synthetic_c3
    res = {}
    for gemm in ("fp8", "same"):
        loop = make_synthetic_c3(N=a.n, M=a.m, d=a.d, k=a.k, H=a.res, W=a.res, timers=True, lists=a.lists, style_density=a.style, w_nn=a.w_nn, feat_dtype=torch.float16, gemm=gemm)
        for _ in range(warmup):
            loop.step()
        loop.knn_events = []; loop.swept_log = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            loop.step()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        knn = loop.knn_ms_per_step(); sw = loop.swept_pairs_per_step()
        nblk, T = (loop.N + 511) // 512, (loop.M + 63) // 64
        frac_swept = sum(sw) / (len(sw) * nblk * T) if sw else 1.0
        flop = 2.0 * loop.N * loop.M * loop.d * frac_swept
        kms = sum(knn) / len(knn)
        peak = MFMA_PEAK_TF["fp8" if gemm == "fp8" else "bf16"]
        res["fp8" if gemm == "fp8" else "f16"] = {"ms_per_step": dt / steps * 1e3, "knn_ms": kms, "swept_fraction": frac_swept, "warm": loop._warm,
                                                  "roofline": {"bound": "mfma", "achieved": flop / (kms * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s",
                                                               "frac": flop / (kms * 1e-3) / 1e12 / peak}, "rows_redone_cold_max": loop.repaired_rows_max}
        del loop
        torch.cuda.empty_cache()
    return {"line": "C3 loop with f16 feature rows: fp8 e4m3 distance-GEMM (BASELINE C5's operands; emit kernel + cached bounds on the quantised problem) next to the f16 distance-GEMM", **res}


def side_line_m1e6(a, steps=3, warmup=3):
    """SURVEY 8(d)'s stress shape (BASELINE config C4's problem on ONE GPU): the same 1M-Gaussian loop against M = 1 000 000 style rows — 15 625 style tiles per
    row block, 512 Tflop per brute-force sweep.  A few iterations after a short warm-up; every sweep priced with its own plan, as in `roofline`."""
    from wast3d_amd.style_loop import make_synthetic_c3
    M = 1_000_000
    loop = make_synthetic_c3(N=a.n, M=M, d=a.d, k=a.k, H=a.res, W=a.res, timers=True, lists=a.lists, style_density=a.style, w_nn=a.w_nn)
    for _ in range(warmup):
        loop.step()
    loop.knn_events = []; loop.swept_log = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        loop.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    knn = loop.knn_ms_per_step(); swept = loop.swept_pairs_per_step()
    pairs_all = ((a.n + 511) // 512) * ((M + 63) // 64)
    flop_full = 2.0 * a.n * M * a.d
    fr = [sp / pairs_all for sp in swept] if len(swept) == len(knn) else [1.0] * len(knn)
    tf = sum(f * flop_full for f in fr) / (sum(knn) * 1e-3) / 1e12
    out = {"line": "C3 loop against M = 1e6 style rows (SURVEY 8d stress; BASELINE config C4's problem on one GPU), bf16, pruned sweeps", "ms_per_step": dt * 1e3,
           "iters_per_s": 1.0 / dt, "knn_ms": sum(knn) / len(knn), "swept_fraction": sum(fr) / len(fr), "brute_force_flop_per_sweep": flop_full,
           "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_TF["bf16"], "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TF["bf16"],
                        "note": "flop EXECUTED (each sweep's own plan) over its own launch time"},
           "rows_redone_cold_max": loop.repaired_rows_max}
    del loop
    torch.cuda.empty_cache()
    return out


def run_whole(a, peak, n_total=1001, cfg_kw=None):
    """A WHOLE run of the notebook's loop length (NB11:1070: `for i in range(1001)`) on a FRESH loop of the same workload: iterations 1..n_total,
    nothing excluded — the cold first sweep, the one-off row-order tree, every order refresh.  What a user of the loop gets per iteration is `ms_mean`; the 20-step
    headline is this run's fastest phase.  Per-step times from HIP events recorded on the main stream after every step (no host synchronisation inside the run);
    the mean from the host clock around the whole run.  `windows`: the loop at five stations of the run — swept fraction of the prune plan, ms per step, and the
    distance-GEMM's rate on the flop it EXECUTES there (same accounting as `roofline`)."""
    from wast3d_amd.style_loop import make_synthetic_c3
    import gc
    fdt = torch.bfloat16 if a.feats == "bf16" else torch.float16
    loop = make_synthetic_c3(N=a.n, M=a.m, d=a.d, k=a.k, H=a.res, W=a.res, timers=True, lists=a.lists, feat_dtype=fdt, gemm=a.gemm, style_density=a.style, w_nn=a.w_nn, **(cfg_kw or {}))
    gc.collect(); gc.freeze()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_total + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(n_total):
        loop.step()
        evs[i + 1].record()
    torch.cuda.synchronize()
    sec = time.perf_counter() - t0
    ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(n_total)]
    knn = loop.knn_ms_per_step()                                   # one entry per iteration
    swept = loop.swept_pairs_per_step()                            # one entry per PRUNED sweep (the first iteration's sweep is cold: no plan)
    n_cold = len(knn) - len(swept)
    pairs_all = ((a.n + 511) // 512) * ((loop.M + 63) // 64)
    fr = [1.0] * n_cold + [sp / pairs_all for sp in swept]
    flop = 2.0 * a.n * a.m * a.d
    srt = sorted(ms)
    def window(c, w=25):
        lo, hi = max(0, c - w), min(n_total, c + w)
        if c == 0:
            lo, hi = 5, 5 + 2 * w                                  # first station: after the cold sweep and the tree build
        k_ms, f = knn[lo:hi], fr[lo:hi]
        tf = sum(x * flop for x in f) / (sum(k_ms) * 1e-3) / 1e12
        return {"iterations": [lo + 1, hi], "ms_per_step": sum(ms[lo:hi]) / (hi - lo), "knn_ms": sum(k_ms) / len(k_ms), "swept_fraction": sum(f) / len(f),
                "knn_tflops_executed": tf, "frac": tf / peak}
    stations = [0, n_total // 4, n_total // 2, 3 * n_total // 4, n_total]
    win = [window(c) for c in stations]
    tf_all = sum(x * flop for x in fr) / (sum(knn) * 1e-3) / 1e12
    sw = [w["swept_fraction"] for w in win]
    out = {"iterations": n_total, "seconds": sec, "iters_per_s": n_total / sec, "ms_mean": sec / n_total * 1e3, "ms_p50": srt[n_total // 2], "ms_p90": srt[int(n_total * 0.9)],
           "ms_max": srt[-1], "ms_first": ms[0], "swept_fraction_at": {str(c): round(w["swept_fraction"], 4) for c, w in zip(stations, win)},
           "swept_fraction_mean": sum(fr) / len(fr), "knn_ms_mean": sum(knn) / len(knn), "knn_tflops_executed": tf_all, "frac": tf_all / peak, "windows": win,
           "plateau": {"swept_fraction_last_quarter_change": sw[-1] - sw[-2], "reached": abs(sw[-1] - sw[-2]) < 0.02},
           "losses_weighted[nn,depth,tv]": loop.loss.tolist(), "rows_redone_cold_max": loop.repaired_rows_max,
           "note": "a fresh loop of the same workload, iterations 1.." + str(n_total) + " with nothing excluded (first sweep cold, the row-order tree at iteration 1, refreshes every 32); "
                   "per-step times = HIP events after every step, mean = host clock around the whole run"}
    del loop
    torch.cuda.empty_cache()
    return out


def side_line_wnn1(a, peak, steps=20, warmup=5, cfg_kw=None):
    """The SAME workload with the NN style term at weight 1.0 — rounds 1-2's loss balance (the headline's w_nn = 1e5 pulls the content rows towards the style
    set, which tightens the bounds; this line separates kernel work from that workload change).  warmup + steps iterations of a fresh loop, timed like the headline."""
    from wast3d_amd.style_loop import make_synthetic_c3
    fdt = torch.bfloat16 if a.feats == "bf16" else torch.float16
    loop = make_synthetic_c3(N=a.n, M=a.m, d=a.d, k=a.k, H=a.res, W=a.res, timers=True, lists=a.lists, feat_dtype=fdt, gemm=a.gemm, style_density=a.style, w_nn=1.0, **(cfg_kw or {}))
    for _ in range(warmup):
        loop.step()
    loop.knn_events = []; loop.swept_log = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        loop.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    knn = loop.knn_ms_per_step(); swept = loop.swept_pairs_per_step()
    pairs_all = ((a.n + 511) // 512) * ((loop.M + 63) // 64)
    fr = [sp / pairs_all for sp in swept] if len(swept) == len(knn) else [1.0] * len(knn)
    flop = 2.0 * a.n * a.m * a.d
    tf = sum(f * flop for f in fr) / (sum(knn) * 1e-3) / 1e12
    out = {"line": "the headline workload with the NN style term at weight 1.0 (rounds 1-2's loss balance; the headline uses 1e5)", "ms_per_step": dt * 1e3, "iters_per_s": 1.0 / dt,
           "knn_ms": sum(knn) / len(knn), "swept_fraction": sum(fr) / len(fr), "losses_weighted[nn,depth,tv]": loop.loss.tolist(),
           "roofline": {"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak}}
    del loop
    torch.cuda.empty_cache()
    return out


def side_line_c2(k, steps=5, warmup=2):
    """BASELINE config C2 at its own shape (the --config C2 line, shortened)."""
    import types
    r = run_c2(types.SimpleNamespace(k=k, steps=steps, warmup=warmup))
    return {"line": r["config"]["workload"], "value": r["value"], "unit": r["unit"], "ms_per_sweep": r["ms_per_step"], "roofline": r["roofline"], "warm_sweep": r["warm_sweep"],
            "warm_emit_sweep": r["warm_emit_sweep"]}


def side_line_raster(k):
    """The rasteriser at a trained scene's tile-instance count (the --config raster line, shortened)."""
    import types
    r = run_raster(types.SimpleNamespace(n=1_000_000, k=k, res=800, log_scale=-4.0), n_rep=5)
    return {"line": r["config"]["workload"], "tile_instances": r["config"]["tile_instances"], "instances_per_tile_mean": r["config"]["instances_per_tile_mean"],
            "contributors_per_pixel_mean": r["config"]["contributors_per_pixel_mean"], "forward_ms": r["forward_ms"], "backward_ms": r["backward_ms"], "rooflines": r["rooflines"]}


def synthetic_nb11_scene(n_content=278_573, n_patch=4165, seed=0, device="cuda"):
    """A content scene and a style patch of the notebook's sizes (NB11 cells 5-39: 278 573 content Gaussians, one 4 165-point patch): points on a unit sphere and a
    ground disc with 1 % jitter (a scene of surfaces, as a Gaussian-splat scene is), and a curved sheet of extent 0.4 with the patch's attribute arrays."""
    import numpy as np
    g = torch.Generator().manual_seed(seed)
    n_sph = int(n_content * 0.7)
    v = torch.randn(n_sph, 3, generator=g); v = v / v.norm(dim=1, keepdim=True)
    r = torch.rand(n_content - n_sph, generator=g).sqrt() * 1.3; th = torch.rand(n_content - n_sph, generator=g) * 6.283185
    disc = torch.stack([r * torch.cos(th), r * torch.sin(th), torch.full_like(r, -1.0)], 1)
    content = (torch.cat([v, disc], 0) + 0.01 * torch.randn(n_content, 3, generator=g)).to(device)
    uv = torch.rand(n_patch, 2, generator=g) * 2 - 1
    patch = torch.stack([uv[:, 0], uv[:, 1], 0.3 * torch.sin(2.5 * uv[:, 0]) * torch.cos(1.5 * uv[:, 1])], 1).mul(0.2)
    style = {"_xyz": patch.numpy(), "_features_dc": torch.rand(n_patch, 1, 3, generator=g).numpy(), "_features_rest": torch.zeros(n_patch, 15, 3).numpy(),
             "_scaling": (torch.randn(n_patch, 3, generator=g) * 0.3 - 4.0).numpy(), "_rotation": torch.randn(n_patch, 4, generator=g).numpy(),
             "_opacity": torch.randn(n_patch, 1, generator=g).numpy()}
    return content, style


def run_stylize(a):
    """stylize() end to end at the notebook's own scale (NB11 cells 5-39): prepare_scene (KMeans, outlier filter, density sampling) -> sample_circles ->
    fit_target_to_domain for every circle (batched, 1001 iterations) -> find_overlapping_gaussians -> assembly.  Seconds per stage."""
    from wast3d_amd.stylize import stylize
    content, style = synthetic_nb11_scene()
    prof = {}
    n_iters = 1001 if a.steps == 10 else a.steps                  # default: the notebook's loop length (NB11:1070); --steps overrides
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = stylize(content, style, n_iters=n_iters, assemble="nb11", profile=prof)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    circles = int(out["n_circles"])
    return {"metric": "stylize(content_scene, style_scene, n_iters) end to end at the notebook's scale (NB11 cells 5-39)", "value": total, "unit": "s", "n_gpus": 1,
            "steps": n_iters, "warmup": 0, "ms_per_step": prof.get("fit_iterations", 0.0) / max(n_iters, 1) * 1e3, "higher_is_better": False, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "stylize: 278 573-point content scene (sphere + disc surfaces), 4 165-point style patch with its attribute arrays, notebook literals otherwise",
                       "content_points": int(content.shape[0]), "domain_points": int(out["n_domain"]), "patch_points": int(style["_xyz"].shape[0]), "circles": circles,
                       "iterations_per_circle": n_iters, "output_gaussians": int(out["_xyz"].shape[0]),
                       "notebook": "278 573 -> 19 390 domain points -> 189 circles -> 378 926 output Gaussians; its fit loop alone: 480 s = 394 circle-iterations/s (BASELINE.md)"},
            "stages_s": {k: round(v, 4) for k, v in prof.items()},
            "circle_iterations_per_s": circles * n_iters / max(prof.get("fit_iterations", 0.0), 1e-9)}


def trained_like_scene(pc, seed=40, log_scale_mean=-4.0, log_scale_std=0.3):
    """Overwrites a GaussianModel IN PLACE with the splat statistics of a TRAINED scene (the C3 generator's isotropic 0.005 splats at opacity 0.1
    give 2.4 tile instances per Gaussian — the rasteriser idles): Gaussians on SURFACES (a unit sphere and a ground disc, 1 % jitter), per-axis log-scales ~
    N(log_scale_mean, log_scale_std) (the law of `synthetic_nb11_scene`'s style patch: anisotropic, e^-4 = 0.018 scene units = ~5 px at the C3 camera), random
    rotations, opacity logits ~ N(0, 2) (a spread from transparent to opaque), small higher-order SH."""
    g = torch.Generator().manual_seed(seed)
    P = pc._xyz.shape[0]
    dev = pc._xyz.device
    n_sph = int(P * 0.7)
    v = torch.randn(n_sph, 3, generator=g); v = v / v.norm(dim=1, keepdim=True)
    r = torch.rand(P - n_sph, generator=g).sqrt() * 1.3; th = torch.rand(P - n_sph, generator=g) * 6.283185
    disc = torch.stack([r * torch.cos(th), r * torch.sin(th), torch.full_like(r, -1.0)], 1)
    pc._xyz.data.copy_((torch.cat([v, disc], 0) + 0.01 * torch.randn(P, 3, generator=g)).to(dev))
    pc._scaling.data.copy_((torch.randn(P, 3, generator=g) * log_scale_std + log_scale_mean).to(dev))
    pc._rotation.data.copy_(torch.randn(P, 4, generator=g).to(dev))
    pc._opacity.data.copy_((torch.randn(P, 1, generator=g) * 2.0).to(dev))
    pc._features_rest.data.copy_((torch.randn(pc._features_rest.shape, generator=g) * 0.05).to(dev))


def run_raster(a, n_rep=10):
    """The rasteriser where it works for its living: 1M Gaussians with a trained scene's splat statistics at 800x800 — tens of tile instances
    per pixel column instead of C3's 2.4 per Gaussian.  Times the loop's own render leg (persistent buffers, the C ABI calls of GaussianStyleLoop) forward and backward
    with HIP events, alone on the chip; rooflines by SURVEY 8(d)'s formulas on the MEASURED tile-instance count R."""
    from wast3d_amd.style_loop import make_synthetic_c3
    loop = make_synthetic_c3(N=a.n, M=4096, d=64, k=a.k, H=a.res, W=a.res, lists="random", timers=True)
    trained_like_scene(loop.pc, log_scale_mean=a.log_scale)
    xyz = loop.pc._xyz.data
    loop.depth_target = torch.full((a.res, a.res), 3.5, device="cuda")
    for _ in range(2):
        loop._render_direct_forward(xyz); loop._render_direct_backward(xyz)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 * n_rep)]
    t0 = time.perf_counter()
    for i in range(n_rep):
        ev[3 * i].record(); loop._render_direct_forward(xyz); ev[3 * i + 1].record(); loop._render_direct_backward(xyz); ev[3 * i + 2].record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n_rep * 1e3
    fwd = sum(ev[3 * i].elapsed_time(ev[3 * i + 1]) for i in range(n_rep)) / n_rep
    bwd = sum(ev[3 * i + 1].elapsed_time(ev[3 * i + 2]) for i in range(n_rep)) / n_rep
    R, P, HW = int(loop.last_R), a.n, a.res * a.res
    radii = loop._rb["radii"]
    tiles = ((a.res + 15) // 16) ** 2
    b_fwd = P * (236 + 75) + R * (12 * 2 + 28) + HW * 32
    b_bwd = R * (28 + 12) + HW * 20 + P * (40 + 472)
    n_contrib = loop._rb["img"][4 * HW: 8 * HW].view(torch.int32).float()     # image buffer: final_T [HW] f32 | n_contrib [HW] u32 | ...

    def rl(nbytes, ms):
        gbs = nbytes / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "algorithmic_bytes": nbytes, "ms": ms, "bound_ms": nbytes / (HBM_PEAK_GBS * 1e9) * 1e3, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs / HBM_PEAK_GBS}
    return {"metric": "rasteriser fwd + bwd on a trained-scene splat distribution (1M Gaussians, 800x800)", "value": 1e3 / (fwd + bwd), "unit": "renders (fwd+bwd)/s", "n_gpus": 1,
            "steps": n_rep, "warmup": 2, "ms_per_step": fwd + bwd, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"raster: {P} Gaussians on surfaces (sphere + disc), per-axis log-scale ~ N({a.log_scale}, 0.3), random rotations, opacity logits ~ N(0, 2), one "
                                   f"{a.res}x{a.res} view, depth-MSE + TV gradients", "tile_instances": R, "instances_per_gaussian": R / P, "instances_per_tile_mean": R / tiles,
                       "visible_gaussians": int((radii > 0).sum()), "contributors_per_pixel_mean": float(n_contrib.mean()), "contributors_per_pixel_max": float(n_contrib.max())},
            "forward_ms": fwd, "backward_ms": bwd, "host_wall_ms": wall,
            "rooflines": [dict(rl(b_fwd, fwd), stage="rasteriser forward (activations, projection, binning, per-tile sort, blend, image losses)",
                               bytes="SURVEY 8d: P(236+75) + R(12*2*passes + 28), passes = 1 (per-tile LDS sort) + HW*32"),
                          dict(rl(b_bwd, bwd), stage="rasteriser backward (blend backward, per-Gaussian chain, activations)", bytes="SURVEY 8d: R(28+12) + HW*20 + P(40 + 236 read + 236 written)")]}


def _claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries loaded below print banners there from C (RCCL's version block) at times of their own choosing:
    everything written to file descriptor 1 from here on goes to stderr, and the returned stream — the real stdout — gets the JSON line alone."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(a, argv, out_stream):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: this process becomes the LAUNCHER.  It touches no GPU (device_count() does not
    initialise HIP on this image; nothing here may, because the ranks are CHILD processes — a process that has initialised the GPU must never be replaced or forked
    into another GPU user), starts one child per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in its environment (what
    torch.distributed.run would set), relays rank 0's single JSON line to its own stdout, and exits non-zero if any rank fails (the others are then terminated) or if
    the box has fewer than N devices."""
    import subprocess
    n = a.gpus
    if a.backend == "nccl":
        have = torch.cuda.device_count()
        if have < n:
            print(f"bench.py: --gpus {n} needs {n} visible GPUs, this box has {have} (torch.cuda.device_count()); nothing was launched", file=sys.stderr)
            return 2
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    import tempfile
    procs = []
    cap = tempfile.TemporaryFile(mode="w+")                         # rank 0's stdout (a file, not a pipe: nothing can block on it)
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), W3D_BENCH_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e, stdout=cap if r == 0 else subprocess.DEVNULL))
    rc = 0
    line = None
    pending = set(range(n))
    while pending and rc == 0:
        for r in list(pending):
            code = procs[r].poll()
            if code is not None:
                pending.discard(r)
                if code != 0:
                    rc = code if code > 0 else 1
                    print(f"bench.py: rank {r} exited with code {code}", file=sys.stderr)
        if pending and rc == 0:
            time.sleep(0.05)
    if rc != 0:
        for r in pending:                                           # a failed rank leaves the others in a collective: end exactly the children started here
            procs[r].terminate()
        for r in pending:
            try:
                procs[r].wait(timeout=10)
            except subprocess.TimeoutExpired:
                procs[r].kill()
    cap.seek(0)
    for ln in cap.read().splitlines():
        if ln.startswith("{"):
            line = ln
    if rc == 0 and line is None:
        print("bench.py: rank 0 printed no JSON line", file=sys.stderr)
        rc = 1
    if line is not None and rc == 0:
        print(line, file=out_stream, flush=True)
    return rc


def comm_record(dist, backend, world, local):
    """Proof that N ranks were in the job: every rank's device, gathered THROUGH the collective backend, + the backend's own version."""
    dev = {"rank": dist.get_rank(), "local_rank": local}
    if backend == "nccl":
        p = torch.cuda.get_device_properties(local)
        dev.update(device=torch.cuda.current_device(), name=p.name, pci_bus_id=getattr(p, "pci_bus_id", None), uuid=str(getattr(p, "uuid", "")))
    else:
        dev.update(device="cpu", pid=os.getpid())
    devs = [None] * world
    dist.all_gather_object(devs, dev)
    ver = None
    if backend == "nccl":
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            ver = None
    return {"world_size": dist.get_world_size(), "backend": backend + (" (RCCL: torch's nccl backend on ROCm)" if backend == "nccl" else ""), "version": ver,
            "ranks_device_ids": [d["device"] for d in devs], "ranks": devs}


def comm_line(timer, steps, dist, device):
    """The `comm` object of an N > 1 line: per-collective ms per step (events on the stream each collective is issued on — or the host clock for CPU tensors — MAX over
    ranks), bytes per step of this rank's buffers, and their sum."""
    from wast3d_amd.distributed import COMM_KEYS
    sm = timer.summary(steps)
    v = torch.tensor([sm[k] for k in COMM_KEYS], device=device, dtype=torch.float64)
    dist.all_reduce(v, op=dist.ReduceOp.MAX)
    out = {k: float(x) for k, x in zip(COMM_KEYS, v.tolist())}
    out["bytes"] = sm["bytes"]
    out["total_ms"] = sum(out[k] for k in COMM_KEYS)
    return out


def run_toy(a, out_stream):
    """--config toy: the distributed plumbing of bench.py (launcher or torchrun -> rendezvous -> barrier / max-over-ranks timing -> rank 0's line) on a CPU loop over the
    chosen backend (gloo here: tests/test_bench_launcher_cpu.py).  One step = an all-reduce of a small vector + a few flops; no GPU, no HIP library."""
    import torch.distributed as dist
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29512")
    if os.environ.get("W3D_BENCH_TOY_FAIL_RANK") == str(rank):       # the launcher test's failing rank
        sys.exit(7)
    dist.init_process_group(a.backend, rank=rank, world_size=world)
    assert a.gpus == world, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    # the sharded loop's own exchange code on small CPU tensors: the six collectives of a step through the wrappers of wast3d_amd.distributed and its CommTimer, so that
    # the `comm` object of an N > 1 line (keys, max over ranks, bytes) is exercised without a GPU (tests/test_bench_launcher_cpu.py)
    from wast3d_amd import distributed as D
    timer = D.CommTimer(True, cuda=False)
    ex = D.GeometryExchange(None, timer)
    P, H, W, Mc, d = 64, 8, 8, 16, 4
    xyz_own = torch.full((P, 4), float(rank + 1)); xyz_all = torch.zeros(world * P, 4)
    gx_all = torch.ones(world * P, 3); gx_own = torch.zeros(P, 3)
    rec_own = torch.full((P, 12), float(rank)); rec_all = torch.zeros(world * P, 12)
    img = torch.zeros(4, H, W); acc_all = torch.ones(world * P, 12); acc_own = torch.zeros(P, 12)
    FY = torch.zeros(world * Mc, d); FY[rank * Mc:(rank + 1) * Mc] = rank + 1.0

    def exchange_step():
        ex.gather_xyz4(xyz_all, xyz_own)
        with timer("style_gather_ms", FY.numel() * 4):
            for c in range(world):
                dist.broadcast(FY[c * Mc:(c + 1) * Mc], src=c)
        with timer("splat_all_gather_ms", rec_all.numel() * 4):
            D.all_gather_rows(rec_own, None, out=rec_all)
        img.fill_(float(rank + 1))
        with timer("image_all_reduce_ms", img.numel() * 4):
            dist.all_reduce(img)
        with timer("accum_reduce_scatter_ms", acc_all.numel() * 4):
            D.reduce_scatter_rows(acc_all, None, out=acc_own)
        ex.scatter_gx(gx_all, gx_own)
        return float(img[0, 0, 0]) + float(xyz_all[:, 0].sum()) / P + float(FY[:, 0].sum()) / Mc + float(acc_own[0, 0]) + float(gx_own[0, 0])

    for _ in range(a.warmup):
        exchange_step()
    timer.reset()
    dist.barrier(); t0 = time.perf_counter()
    acc = 0.0
    for _ in range(a.steps):
        acc += exchange_step()
    dist.barrier(); sec = time.perf_counter() - t0
    t = torch.tensor([sec], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); sec = float(t.item())
    comm = comm_line(timer, a.steps, dist, "cpu")
    rec = comm_record(dist, a.backend, world, local)
    if rank == 0:
        print(json.dumps({"metric": "toy exchange steps/s (plumbing check, CPU)", "value": a.steps / sec, "unit": "steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": sec / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "toy"}, "checksum": acc, "rccl": rec, "comm": comm}), file=out_stream, flush=True)
    dist.destroy_process_group()


def main():
    out_stream = _claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C3", choices=["C3", "C4", "C2", "C1", "nb11", "stylize", "raster", "toy"], help="C3 (default): the headline workload; C4: C3 with M = 1e6 style rows sharded over the ranks; "
                    "C1: 2k points, sliced-Wasserstein head, 50 Adam iterations (HIP next to torch-CPU); "
                    "C2: 100k x 100k x 128 kNN only; nb11: the notebook's own geometry-loop shape; stylize: stylize() end to end at the notebook's scale; "
                    "raster: the render leg alone on a trained scene's splat statistics")
    ap.add_argument("--log-scale", type=float, default=-4.0, help="--config raster: mean of the per-axis log-scales (std 0.3)")
    ap.add_argument("--n", type=int, default=1_000_000, help="Gaussians of the scene (strong scaling) / per GPU (weak scaling)")
    ap.add_argument("--m", type=int, default=None, help="style rows (total); default 65536 (C3) / 1000000 (C4)")
    ap.add_argument("--d", type=int, default=256)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--res", type=int, default=800)
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"], help="N > 1: strong = the same --n Gaussians over all ranks (default); weak = --n Gaussians per rank")
    ap.add_argument("--style-gather", default=None, choices=["cached", "inloop", "inloop_chunked"], help="sharded style rows: `cached` = gathered once at setup (the style set is "
                    "constant in the reference; default); `inloop` = re-delivered shard by shard EVERY iteration (north_star's RCCL all-gather of style tiles) while the rows are "
                    "featurised and the render leg runs, then bound + plan + one pruned emit sweep; `inloop_chunked` = rounds 1-5's form, the sweep cut per arriving shard "
                    "(unpruned list sweeps under the motion bound)")
    ap.add_argument("--both-style-gathers", action="store_true", help="sharded runs: after the timed region, time the same steps with the OTHER style-gather mode on a fresh loop and "
                    "put it in the line as `style_gather_other` (SURVEY 8e: report both cached and in-loop numbers); default on for --config C4")
    ap.add_argument("--lists", default="exact", choices=["exact", "knn", "random"],
                    help="descriptor neighbour lists of the Z-ordered scene: 'exact' (default) = the exact 256 nearest neighbours of the frozen copy, what the reference "
                         "builds (NB11:1055-1056); 'knn' = the nearest within +-4096 positions of the curve (every entry inside the LDS window: 4-5 %% faster); 'random' = "
                         "uniformly random (worst case for the gathers)")
    ap.add_argument("--feats", default="bf16", choices=["bf16", "f16"], help="feature-row dtype")
    ap.add_argument("--gemm", default="same", choices=["same", "fp8"], help="distance-GEMM operands: the feature dtype, or fp8 e4m3 + per-row scale (BASELINE config C5; NOT the headline dtype)")
    ap.add_argument("--style", default="matched", choices=["matched", "sparse"],
                    help="synthetic style patch (SURVEY 8d C3: 'style patch << content, as in NB11'): 'matched' = M points at the content's point density, a small "
                         "patch of the same kind of surface — every content row has its own neighbours, the hard case for the selection; 'sparse' = M points over "
                         "the content's whole extent — 16x sparser, all content rows end up with the same ~27 neighbours (the easy case)")
    ap.add_argument("--views", type=int, default=1, help="views rendered per iteration (BASELINE config C5: 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--w-nn", type=float, default=None, help="weight of the NN style term (default: the workload's own, chosen so that the weighted NN term is within "
                    "10x of the depth term at iteration 0, as the reference balances its terms: NB11:1085)")
    ap.add_argument("--late", type=int, default=0, help="after the timed region: advance this many untimed iterations and time 20 more (`late` object); default 0 — "
                    "since round 4 the whole-run object `run_1001` carries the late phase")
    ap.add_argument("--whole-run", type=int, default=None, help="after the timed region: a whole run of this many iterations on a fresh loop (`run_1001` object); default 1001 "
                    "(the notebook's loop length, NB11:1070) for the default single-GPU C3 line, 0 otherwise")
    ap.add_argument("--no-side-lines", action="store_true", help="skip the C2 / fp8 side lines")
    ap.add_argument("--main-priority", type=int, default=None, help="(A/B) run the loop on a stream of this priority instead of the default stream (-1 = high)")
    ap.add_argument("--loop-config", default=None, help='JSON of wast3d_amd.style_loop.LoopConfig fields for A/B runs, e.g. \'{"schedule": "early", "prune": false}\' (default: LoopConfig())')
    ap.add_argument("--force-distributed", action="store_true", help="run the sharded loop even at world size 1 (plumbing check)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="collective backend: nccl (= RCCL on ROCm; default) or gloo (--config toy on the CPU; any other config: a plumbing check with every rank on the GPUs that exist)")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torch.distributed.run: start the ranks ourselves (children; before anything here touches a GPU) and relay rank 0's line
        sys.exit(launch_ranks(a, sys.argv[1:], out_stream))
    if a.config == "toy":
        run_toy(a, out_stream)
        return
    if a.m is None:
        a.m = 1_000_000 if a.config == "C4" else 65536
    if a.style_gather is None:
        # the style set is constant (NB11:1057-1063): one all-gather at setup, then every sweep is the single-GPU one (every bound, pruned).  `inloop` re-delivers
        # the shards every iteration (the mode that exercises north_star's collective; since round 6 on the same pruned emit sweep)
        a.style_gather = "cached"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs the MI355X"
    if a.backend == "gloo":
        # dev hook (plumbing check of the N > 1 code path on a box with fewer GPUs than ranks): every rank on the devices that exist, the collectives over gloo on CUDA
        # tensors — what tests/sharded_multirank.py does for the loop itself.  Labelled in the line (`rccl.backend` = gloo); never a scaling number.
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if a.main_priority is not None:
        torch.cuda.set_stream(torch.cuda.Stream(priority=a.main_priority))
    from wast3d_amd import _lib
    _lib.lib()
    if a.config in ("C1", "C2", "nb11", "stylize", "raster"):
        assert world == 1, f"--config {a.config} is a single-GPU workload"
        print(json.dumps({"C1": run_c1, "C2": run_c2, "nb11": run_nb11, "stylize": run_stylize, "raster": run_raster}[a.config](a)), file=out_stream, flush=True)
        return
    dist = None
    if world > 1 or a.force_distributed or a.config == "C4":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))   # "nccl" is RCCL on ROCm
    if a.gpus != world:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` (it launches its own ranks) or under torch.distributed.run with --nproc-per-node N")
    comm = comm_record(dist, a.backend, world, local) if dist is not None else None      # every rank's device, gathered through RCCL itself

    fdt = torch.bfloat16 if a.feats == "bf16" else torch.float16
    cfg_kw = {}
    if a.loop_config:
        from wast3d_amd.style_loop import LoopConfig
        cfg_kw = {"config": LoopConfig(**json.loads(a.loop_config))}
    if dist is None:
        from wast3d_amd.style_loop import make_synthetic_c3
        loop = make_synthetic_c3(N=a.n, M=a.m, d=a.d, k=a.k, H=a.res, W=a.res, timers=True, lists=a.lists, feat_dtype=fdt, gemm=a.gemm, views=a.views, style_density=a.style, w_nn=a.w_nn, **cfg_kw)
    else:
        from wast3d_amd.distributed import make_synthetic_c4
        loop = make_synthetic_c4(n_total=a.n, M=a.m, d=a.d, k=a.k, H=a.res, W=a.res, timers=True, lists=a.lists, feat_dtype=fdt, gemm=a.gemm, views=a.views, style_density=a.style,
                                 style=a.style_gather, scaling=a.scaling, w_nn=a.w_nn)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    base_loop = getattr(loop, "b", loop)                          # the sharded loop wraps a GaussianStyleLoop
    if a.whole_run is None:
        a.whole_run = 1001 if (world == 1 and a.config == "C3" and a.views == 1 and not a.force_distributed) else 0
    for _ in range(a.warmup):
        loop.step()
    loop.knn_events = []
    base_loop.swept_log = []
    base_loop.stage_ms()
    # The host side of a step is ~2 ms of Python; a full pass of the cyclic garbage collector over everything the set-up left behind takes tens of ms
    # (scripts/rank_sim.py: one 37-90 ms iteration in 24).  What INTEGRATION.md tells every user of the loops to do once the loop is built: collect, then move the
    # survivors to the permanent generation.  The collector STAYS ON in the timed region (100 steps: 15.08 ms per step with it, 15.03 with gc.disable()).
    import gc
    gc.collect(); gc.freeze()

    kern_ms, strag_pairs = [], []                                 # of the LAST timed_steps call: the emit kernel's own duration per step, the plan's straggler pairs per step

    def timed_steps(n):
        """n steps between barriers; returns (seconds [max over ranks], kNN ms of every step, swept pairs of every step, stage times, weighted losses)."""
        loop.knn_events = []
        base_loop.swept_log = []
        base_loop.stage_ms()
        base_loop.knn_kernel_events = []; base_loop.strag_log = []
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            loop.step()
        barrier()
        sec = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([sec], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sec = float(t.item())
        kern_ms.clear(); kern_ms.extend(base_loop.knn_kernel_ms_per_step() if hasattr(base_loop, "knn_kernel_ms_per_step") else [])
        strag_pairs.clear(); strag_pairs.extend(base_loop.straggler_pairs_per_step() if hasattr(base_loop, "straggler_pairs_per_step") else [])
        return sec, base_loop.knn_ms_per_step(), base_loop.swept_pairs_per_step(), base_loop.stage_ms(), loop.loss.tolist()

    def comm_summary(lp, steps):
        return comm_line(lp.comm, steps, dist, "cuda")

    if dist is not None:
        loop.comm.reset()
    dt, knn_steps, swept_steps, stage, losses = timed_steps(a.steps)
    kern_steps, strag_steps = list(kern_ms), list(strag_pairs)
    comm_ms = comm_summary(loop, a.steps) if dist is not None else None
    knn_ms = sum(knn_steps) / max(len(knn_steps), 1)
    losses0 = list(losses)
    R_timed = int(loop.last_R)
    chunked = base_loop.style_stream is not None
    pruned = (not chunked) and base_loop._warm in (3, 4)              # pruned sweeps (w3d_knn_prune_plan): ONE launch that sweeps, per row block, only the style tiles it may need
    distinct_rows = int(base_loop.nn_idx.unique().numel())
    alone, unpruned, late, sides, sweep_alone = {}, None, None, [], None
    if dist is None and a.views == 1:
        # (after the timed region; nothing below feeds `value`)  stand-alone stage times: two serialised iterations
        for _ in range(2):
            loop.step(serial=True)
        torch.cuda.synchronize()
        if pruned and base_loop.knn_events and base_loop.swept_log:   # the sweep of the last serialised iteration: the chip to itself (in the loop the lean Adam stream runs under it)
            e0, e1 = base_loop.knn_events[-1]
            sweep_alone = (e0.elapsed_time(e1), int(base_loop.swept_log[-1].item()))
        loop.knn_events = []; base_loop.swept_log = []
        alone = base_loop.stage_ms()
        if a.gemm == "same" and pruned:
            unpruned = time_unpruned_sweeps(base_loop)
    if a.late > 0 and a.views == 1:
        for _ in range(a.late):
            loop.step()
        l_dt, l_knn, l_swept, l_stage, l_losses = timed_steps(20)
        late = (l_dt, l_knn, l_swept, l_stage, l_losses)
    if dist is not None:                                           # the NN loss is accumulated per shard: the shares add up
        l0 = torch.tensor([losses0[0]], device="cuda"); dist.all_reduce(l0); losses0[0] = float(l0.item())
    if dist is not None:
        # (after the timed region; nothing below feeds `value`)  the same steps with every collective replaced by a device-local stand-in of the same buffers: what the
        # step costs with free communication — `exposed_ms` = the timed step minus this (what the exchanges, and the skew between ranks they expose, add to a step)
        with loop.local_collectives():
            for _ in range(2):
                loop.step()
            c_dt = timed_steps(a.steps)[0]
        for _ in range(2):
            loop.step()
        r_dt = timed_steps(a.steps)[0]                             # ... and with the real collectives once more: the loop's own state (swept fraction) drifts with the iteration count
        comm_ms["compute_only_ms_per_step"] = c_dt / a.steps * 1e3
        comm_ms["ms_per_step_again"] = r_dt / a.steps * 1e3
        comm_ms["exposed_ms"] = 0.5 * (dt + r_dt) / a.steps * 1e3 - comm_ms["compute_only_ms_per_step"]
        comm_ms["note"] = ("ms per step of every collective between a pair of events on the stream it is issued on (includes waiting for the slowest peer), max over ranks; "
                           "style_gather_ms runs on its own stream from t = 0; compute_only = the same number of steps with the collectives replaced by local copies, run "
                           "AFTER the timed region and followed by as many real steps again (ms_per_step_again); exposed = mean(ms_per_step, ms_per_step_again) - compute_only")
    if rank == 0:
        ms = dt / a.steps * 1e3
        strong = a.scaling == "strong"
        n_rank = a.n // world if (strong and dist is not None) else a.n
        n_scene = a.n if strong else a.n * world
        m_eff = base_loop.M                                      # style rows swept (the padded layout of the in-loop gather adds < 64 rows per shard)
        # the timed region brackets every launch of the kernel in a step: per-LAUNCH averages (what rocprofv3 --stats reports) = per-step figures / launches
        # (unpruned 16-bit sweeps are one emit + select group since round 5; only the fp8 list kernel still splits its last round: w3d_knn_mfma_launches)
        launches = base_loop.style_stream.parts if chunked else (_lib.lib().w3d_knn_mfma_launches(n_rank, a.m) if (a.gemm == "fp8" and not pruned) else 1)
        flops_step = 2.0 * n_rank * a.m * a.d                    # flop of the BRUTE-FORCE sweep of this rank (padding rows excluded): what the reference's cdist costs
        pairs_all = ((n_rank + 511) // 512) * ((base_loop.M + 63) // 64)      # (row block, style tile) pairs of a full sweep
        peak = MFMA_PEAK_TF["fp8" if a.gemm == "fp8" else a.feats]

        def executed(knn_list, swept_list):
            """(TFLOP/s on the flop EXECUTED, mean swept fraction, per-step records): every step's flop comes from ITS OWN plan, every step's time from ITS OWN events."""
            if not knn_list:
                return 0.0, 1.0, []
            fr = [sp / pairs_all for sp in swept_list] if (pruned and len(swept_list) == len(knn_list)) else [1.0] * len(knn_list)
            tf = sum(f * flops_step for f in fr) / (sum(knn_list) * 1e-3) / 1e12
            return tf, sum(fr) / len(fr), [{"knn_ms": round(t, 4), "swept_fraction": round(f, 5)} for t, f in zip(knn_list, fr)]

        ach_group, swept, per_step = executed(knn_steps, swept_steps)
        # the dominant KERNEL's own duration (events the library records right around knn_emit_kernel) when the sweep is a pruned emit sweep with cached bounds; otherwise the
        # launch group is all there is
        kernel_timed = pruned and len(kern_steps) == len(knn_steps) and len(kern_steps) > 0
        if kernel_timed:
            ach, _, _ = executed(kern_steps, swept_steps)
            for r, t in zip(per_step, kern_steps):
                r["emit_kernel_ms"] = round(t, 4)
            kern_ms_mean = sum(kern_steps) / len(kern_steps)
        else:
            ach, kern_ms_mean = ach_group, knn_ms
        if len(strag_steps) == len(per_step):
            for r, sp in zip(per_step, strag_steps):
                r["straggler_pairs"] = sp
        traffic, traffic_src = knn_traffic(a)
        iters_s = a.steps / dt
        out = {
            "metric": "style-loop iters/sec @1M Gaussians, 256-d feats; kNN dist-GEMM TFLOP/s",
            "value": iters_s if strong else iters_s * world * (a.n / 1e6),
            "unit": "iters/s" if strong else "iters/s x 1M Gaussians (WEAK scaling: the scene grows with the GPU count; not the headline quantity)",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
            "dtype": ("fp8 e4m3 + per-row E8M0 scale" if a.gemm == "fp8" else a.feats) + " (dist-GEMM, f32 accumulate) / f32 (raster, Adam)",
            "data": "synthetic",
            "config": {"workload": ("C3: 1M Gaussians, diff. rasteriser fwd/bwd + NN style loss, single 800x800 view" if a.config == "C3" else
                                    "C4: C3's scene, M = 1e6 style rows sharded over the ranks (global kNN via " + {"inloop": "per-shard RCCL broadcasts every iteration, then the pruned emit sweep", "inloop_chunked": "per-shard RCCL broadcasts under the chunked distance-GEMM", "cached": "one cached all-gather"}[a.style_gather] + ")")
                                   + ("" if a.style == "matched" else " [easy style patch: --style sparse]"),
                       "gaussians_scene": n_scene, "gaussians_per_gpu": n_rank, "style_rows": a.m, "style_rows_swept_incl_padding": m_eff, "feat_dim": a.d, "k": a.k,
                       "descriptor_lists": {"knn": "256 nearest neighbours of the frozen copy within +-4096 positions of the Z-order curve the Gaussians are stored along", "exact": "the exact 256 nearest neighbours of the frozen copy, Gaussians in Z-order", "random": "uniformly random"}[a.lists],
                       "style_patch": {"sparse": "descriptors of M uniform points over the content's extent (16x sparser than the content)", "matched": "descriptors of M uniform points at the content's point density (a small patch, as in NB11)"}[a.style],
                       "feature_rows": "shifted by the style set's mean row before the 16-bit rounding (distances unchanged)" if base_loop.center is not None else "as computed (not centred)",
                       "loss_weights[nn,depth,tv]": [base_loop.w_nn, base_loop.w_depth, base_loop.w_tv],
                       # what the headline is sensitive to: the NN term's weight (the same loop at w_nn = 1.0 is ~0.6x: side_lines), the warm-start
                       # bound, the two-stream schedule and where the render leg's Adam step runs
                       "w_nn": base_loop.w_nn, "warm": base_loop._warm_mode, "adam_under_sweep": bool(base_loop.cfg.adam_under_sweep and base_loop.schedule == "front" and dist is None),
                       "loop_config": {f: getattr(base_loop.cfg, f) for f in ("warm", "prune", "emit", "bounds_cache", "bounds_summaries", "straggler_rows", "nn_sum_cache", "row_order_refresh", "row_deal", "schedule", "adam_under_sweep")},
                       "distinct_style_rows_selected": distinct_rows, "warm_start": base_loop._warm_mode, "schedule": base_loop.schedule, "rows_redone_cold_max": base_loop.repaired_rows_max,
                       "image": [a.res, a.res], "views_per_iteration": a.views, "tile_instances": R_timed,
                       "parallelism": "1 GPU" if world == 1 else f"one scene, Gaussians sharded x{world}: positions all-gather / dL/dxyz reduce-scatter, splat all-gather / image all-reduce / accumulator reduce-scatter over RCCL; style rows {a.style_gather}",
                       "losses_weighted[nn,depth,tv]": losses0},
            "roofline": {"kernel": (("knn_emit_kernel<256,bf16,16x16x32> (fused distance-GEMM + candidate emission over the tiles of the prune plan, one persistent workgroup per CU; timed alone — the launch group with knn_straggler_kernel and knn_select_kernel behind it: launch_group)"
                                     if getattr(base_loop, "_warm", 0) == 4 else "knn_mfma_persistent_kernel<256,64,128,16x16x32> (fused distance-GEMM + running top-k over the tiles of the prune plan; one persistent workgroup per CU)") if pruned else
                                    "knn_mfma_kernel<256,64,128,16x16x32> (fused distance-GEMM + running top-k)") + (", swept in groups of style shards as they arrive" if chunked else ""),
                         "bound": "mfma", "achieved": ach, "peak": peak,
                         "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_src, "launch_ms": kern_ms_mean / launches, "step_ms": knn_ms,
                         "launch_group": {"ms": knn_ms, "achieved": ach_group, "frac": ach_group / peak,
                                          "note": "the same flop over the whole launch group of a step (counter reset, knn_emit_kernel, knn_straggler_kernel, knn_select_kernel) — what rounds 4-5 "
                                                  "printed as `frac`; `frac` / `launch_ms` are the emit kernel alone (events the library records right around it: w3d_knn_mfma_debug_events), the "
                                                  "figure rocprofv3's per-kernel average reproduces"} if kernel_timed else None,
                         "bound_ms": flops_step * swept / (peak * 1e12) * 1e3, "algorithmic_flop_per_launch": flops_step * swept / launches, "launches_per_step": launches,
                         "swept_fraction": swept, "per_step": per_step, "brute_force_flop_per_step": flops_step,
                         "brute_force_equivalent_tflops": flops_step / (knn_ms * 1e-3) / 1e12,
                         "stragglers": ({"rows": base_loop.cfg.straggler_rows, "pairs_per_step": sum(strag_steps) / len(strag_steps),
                                         "flop_per_step": sum(strag_steps) / len(strag_steps) * 64 * 2.0 * a.d,
                                         "note": "(row, tile) pairs the plan took out of the blocks' tile lists because at most `rows` of the block's 512 rows still needed the tile; "
                                                 "knn_straggler_kernel computes them on the same MFMA chain inside the launch group — their flop (pairs x 64 x 2d) is NOT in `achieved`"}
                                        if strag_steps else None),
                         "note": "per rank; achieved = sum_i swept_i * 2NMd / sum_i t_i over the timed steps: swept_i = the (row block, style tile) pairs step i's own prune plan kept "
                                 "(tiles an exact ball test rules out are not swept: that flop is neither executed nor counted), t_i = HIP events on the launch stream around step i's "
                                 "emit kernel (launch_group: around the whole group); brute_force_equivalent_tflops = the full sweep's flop over the group's time, NOT a roofline figure"},
        }
        if dist is None and a.views == 1 and sweep_alone is not None:
            sa_ms, sa_pairs = sweep_alone
            sa_frac = sa_pairs / (((n_rank + 511) // 512) * ((base_loop.M + 63) // 64))
            sa_tf = flops_step * sa_frac / (sa_ms * 1e-3) / 1e12
            out["roofline"]["sweep_alone"] = {"launch_ms": sa_ms, "swept_fraction": sa_frac, "achieved": sa_tf, "peak": peak, "unit": "TFLOP/s", "frac": sa_tf / peak,
                                              "note": "the same sweep + select with the chip to itself (a serialised iteration after the timed region); in the timed steps the "
                                                      "render leg's Adam step runs UNDER the sweep as a 32-VGPR kernel (LoopConfig.adam_under_sweep): the step is ~0.3 ms shorter, the sweep ~0.1 ms longer"}
        if unpruned is not None:
            u_ms, u_same = unpruned
            u_tf = flops_step / (u_ms * 1e-3) / 1e12
            out["roofline"]["unpruned"] = {"sweep_ms": u_ms, "launches": 1, "flop": flops_step, "achieved": u_tf, "peak": peak, "unit": "TFLOP/s",
                                           "frac": u_tf / peak, "lists_equal_pruned_sweep": u_same,
                                           "note": "the sweep over ALL style tiles of the same rows without a plan (match_prepared warm = 1: since round 5 also knn_emit_kernel + knn_select_kernel for "
                                                   "16-bit rows; the list kernel for fp8), bound from the current neighbours, 3 sweeps after the timed region"}
        out["rccl"] = comm
        if comm_ms is not None:
            out["comm"] = comm_ms
            out["config"]["style_gather"] = a.style_gather
        out["knn_dist_gemm_tflops"] = ach_group * (world if dist is not None else 1)      # (on the launch group's time, as in rounds 4-5)
        out["rooflines"] = [{"stage": "distance-GEMM + top-k (" + ("launch group: knn_emit_kernel + knn_straggler_kernel + knn_select_kernel, pruned" if pruned else "knn_mfma_kernel") + ")", "bound": "mfma", "algorithmic_flop": flops_step * swept,
                             "ms": knn_ms, "bound_ms": flops_step * swept / (peak * 1e12) * 1e3, "achieved": ach_group, "peak": peak, "unit": "TFLOP/s", "frac": ach_group / peak, "swept_fraction": swept}]
        if pruned and "prune" in stage:
            pf = 2.0 * n_rank * ((base_loop.M + 63) // 64) * a.d       # rows x tile centres x d
            t = alone.get("prune", stage["prune"])
            out["rooflines"].append({"stage": "prune plan (knn_tile_prune_kernel: rows x tile centres on the matrix cores + ball test)", "bound": "mfma", "algorithmic_flop": pf, "ms": t,
                                     "ms_in_loop": stage.get("prune"), "bound_ms": pf / (peak * 1e12) * 1e3, "achieved": pf / (t * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s",
                                     "frac": pf / (t * 1e-3) / 1e12 / peak})
        out["rooflines"] += stage_rooflines(a, base_loop, stage, world if dist is not None else 1, R_timed, alone)
        if late is not None:
            l_dt, l_knn, l_swept, l_stage, l_losses = late
            l_ach, l_frac, _ = executed(l_knn, l_swept)
            out["late"] = {"iterations_before": a.warmup + a.steps + (2 if alone else 0) + a.late, "steps": 20, "ms_per_step": l_dt / 20 * 1e3, "iters_per_s": 20 / l_dt,
                           "knn_ms": sum(l_knn) / max(len(l_knn), 1), "swept_fraction": l_frac, "achieved_tflops": l_ach, "frac": l_ach / peak,
                           "losses_weighted[nn,depth,tv]": l_losses,
                           "note": "the SAME loop further into its run (the swept fraction follows the content rows' distance to the style set); same accounting as `roofline`"}
        if a.whole_run > 0 and dist is None:
            out["run_1001" if a.whole_run == 1001 else f"run_{a.whole_run}"] = run_whole(a, peak, a.whole_run, cfg_kw)
        if world == 1 and not a.no_side_lines and a.config == "C3" and a.views == 1:
            def side(fn, *args, **kw):                                 # a side line must never cost the headline line, nor the other side lines
                try:
                    sides.append(fn(*args, **kw))
                except Exception as e:
                    sides.append({"line": fn.__name__, "error": repr(e)})
            side(side_line_fp8, base_loop)
            side(side_line_c2, a.k)
            if a.n == 1_000_000 and a.feats == "bf16" and a.gemm == "same":
                side(side_line_fp8_loop, a)
                side(side_line_m1e6, a)
            if a.w_nn is None:
                side(side_line_wnn1, a, peak, cfg_kw=cfg_kw)
            side(side_line_raster, a.k)
            out["side_lines"] = sides
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"], out["parity_vs_fp32"] = cpu_baseline(base_loop, a.n, a.m, a.d, a.k, a.res, a.res)
    if dist is not None and (a.both_style_gathers or a.config == "C4"):
        # (all ranks) the same steps with the OTHER style-gather mode on a fresh loop, after everything of the headline loop has been read
        om = "inloop" if a.style_gather == "cached" else "cached"
        del loop, base_loop
        torch.cuda.empty_cache()
        from wast3d_amd.distributed import make_synthetic_c4
        loop = make_synthetic_c4(n_total=a.n, M=a.m, d=a.d, k=a.k, H=a.res, W=a.res, timers=True, lists=a.lists, feat_dtype=fdt, gemm=a.gemm, views=a.views, style_density=a.style,
                                 style=om, scaling=a.scaling, w_nn=a.w_nn)
        base_loop = loop.b
        for _ in range(a.warmup):
            loop.step()
        loop.comm.reset()
        o_dt, o_knn, o_swept, _o_stage, _ = timed_steps(a.steps)
        o_comm = comm_summary(loop, a.steps)
        if rank == 0:
            out["style_gather_other"] = {"style_gather": om, "ms_per_step": o_dt / a.steps * 1e3, "iters_per_s": a.steps / o_dt, "knn_ms": sum(o_knn) / max(len(o_knn), 1),
                                         "comm": o_comm, "warm": base_loop._warm,
                                         "note": "a fresh loop of the same workload with the other style-gather mode, same warm-up and step count (SURVEY 8e: both numbers)"}
    if rank == 0:
        print(json.dumps(out), file=out_stream, flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
