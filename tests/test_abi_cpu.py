"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/wast3d_hip.h declares; the product package never imports the oracle; host-side logic."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from wast3d_amd import build
    return build.build(verbose=False)


def header_symbols():
    src = open(os.path.join(ROOT, "include", "wast3d_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(w3d_[a-z0-9_]+)\s*\(", src)) - {"w3d_alloc_fn"})


def test_library_exports_every_declared_symbol(built_lib):
    l = ctypes.CDLL(built_lib)
    syms = header_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(l, s)]
    assert not missing, missing


def test_ctypes_signature_table_matches_header(built_lib):
    from wast3d_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_symbols()
    l = _lib.lib()
    assert l.w3d_abi_version() == 7
    assert l.w3d_strerror(0) == b"ok" and b"workspace" in l.w3d_strerror(-3)


def test_no_compute_without_gpu_is_loud(built_lib):
    import torch
    from wast3d_amd import ops, _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.W3dError):
        ops.get_descriptors(torch.zeros(4, 3), torch.zeros(2, 3, dtype=torch.int64))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "wast3d_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in re.sub(r"#.*|//.*", "", txt).replace("oracle/", ""), f  # comments may cite it
    code = "import sys; sys.path.insert(0, %r); import wast3d_amd, wast3d_amd.ops, wast3d_amd.geometry_loop; print(any(m.startswith('oracle') for m in sys.modules))" % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.stdout.strip() == "False", out.stdout + out.stderr


def test_argument_validation_needs_no_gpu(built_lib):
    """Every entry point validates its arguments before touching the device: error codes (never exceptions / crashes) for NULL
    pointers, bad sizes and unsupported dtype/size combinations.  The reference raises RuntimeError/Exception at the same places
    (rasterize_points.cu:58-60, diff_gaussian_rasterization/__init__.py:207-211); the ABI returns W3D_ERR_* which the binding raises."""
    import ctypes as C
    from wast3d_amd import _lib
    L = _lib.lib()
    ARG, WS, UNSUP = -1, -3, -4
    assert L.w3d_knn_match_fwd(None, None, 10, 10, 64, 5, 1, None, None, None, None, 0, None) == ARG
    buf = C.create_string_buffer(1024); p = C.cast(buf, C.c_void_p)
    assert L.w3d_knn_match_fwd(p, p, 10, 10, 64, 20, 1, None, p, p, p, 1024, None) == ARG          # k > M
    assert L.w3d_knn_match_fwd(p, p, 10, 100, 48, 20, 0, None, p, p, p, 1024, None) == UNSUP       # fp32 path: d <= 16 only
    assert L.w3d_knn_match_fwd(p, p, 10, 100, 256, 20, 1, None, p, p, p, 16, None) == WS           # workspace too small
    assert L.w3d_knn_match_fwd(p, p, 10, 100, 96, 20, 1, None, p, p, p, L.w3d_knn_workspace_bytes(10, 100, 96, 20, 1), None) in (UNSUP, WS)
    assert L.w3d_knn_workspace_bytes(1000, 65536, 256, 20, 1) > (1000 + 65536) * 4
    assert L.w3d_descriptor_fwd(None, None, 5, 4, 0, None, None) == ARG
    assert L.w3d_descriptor_fwd(p, p, 5, 1, 0, p, None) == ARG                                      # K >= 2
    assert L.w3d_descriptor_mse_fwd_bwd(p, p, p, 10, 4, 3, 1.0, p, p, None) == ARG                  # R not a multiple of seg_rows
    assert L.w3d_domain_loss_fwd_bwd(p, 10, 7, p, p, 1, 10, 20, 1.0, p, p, None, None) == UNSUP     # d <= 4
    assert L.w3d_domain_loss_fwd_bwd(p, 10, 3, p, p, 0, 10, 20, 1.0, p, p, None, None) == ARG
    assert L.w3d_adam_step(None, 1, 1, 0.9, 0.999, 1e-8, 0, None) == ARG
    grp = (_lib.AdamGroup * 1)(); grp[0].param = 8; grp[0].grad = 16; grp[0].exp_avg = 32; grp[0].exp_avg_sq = 48; grp[0].numel = 4; grp[0].lr = 1e-3
    assert L.w3d_adam_step(grp, 1, 0, 0.9, 0.999, 1e-8, 0, None) == ARG                             # step counts from 1
    assert L.w3d_adam_step(grp, 1, 1, 0.9, 0.999, 1e-8, 0, None) == ARG                             # mis-aligned tensors
    assert L.w3d_adam_step(grp, 9, 1, 0.9, 0.999, 1e-8, 0, None) == ARG                             # > 8 groups per launch
    grp[0].param = 16; grp[0].numel = 0
    assert L.w3d_adam_step(grp, 1, 1, 0.9, 0.999, 1e-8, 4, None) == ARG and L.w3d_adam_step(grp, 1, 1, 0.9, 0.999, 1e-8, -1, None) == ARG   # zero_grad is a bit field 0..3
    assert L.w3d_adam_step(grp, 1, 1, 0.9, 0.999, 1e-8, 3, None) == 0                               # (an empty group: nothing launched)
    assert L.w3d_tv_loss_fwd_bwd(p, 3, 1, 5, 1.0, 0, p, p, None) == ARG                             # needs H, W >= 2
    assert L.w3d_mse_loss_fwd_bwd(None, p, 5, 1.0, 0, p, p, None) == ARG
    rin = _lib.RasterIn(); rin.P = 5; rin.H = 8; rin.W = 8
    n = C.c_int(0)
    cb = _lib.ALLOC_FN(lambda u, b: 0)
    assert L.w3d_raster_forward(C.byref(rin), p, p, cb, None, p, p, p, C.byref(n), None) == ARG    # no means / matrices
    rin.means3D = rin.opacities = rin.viewmatrix = rin.projmatrix = rin.background = rin.sampling_offsets = C.addressof(buf)
    assert L.w3d_raster_forward(C.byref(rin), p, p, cb, None, p, p, p, C.byref(n), None) == ARG    # neither SHs nor colours
    rin.shs = rin.colors_precomp = C.addressof(buf)
    assert L.w3d_raster_forward(C.byref(rin), p, p, cb, None, p, p, p, C.byref(n), None) == ARG    # both SHs and colours
    rin.shs = None
    assert L.w3d_raster_forward(C.byref(rin), p, p, cb, None, p, p, p, C.byref(n), None) == ARG    # no scale/rotation and no covariance
    assert L.w3d_raster_blend_forward(p, 5, 8, 8, 1, 1, p, p, p, p, cb, None, p, p, C.byref(n), None) == ARG   # empty band
    assert L.w3d_mark_visible(None, None, None, 3, None, None) == ARG
    assert L.w3d_raster_geom_bytes(1000) > 1000 * 48 and L.w3d_raster_image_bytes(800, 800) >= 800 * 800 * 8
    assert L.w3d_sw_loss_fwd_bwd(p, p, 4, 8, 1.0, p, p, p, 16, None) == WS
    assert L.w3d_sw_features_fwd_bwd(p, p, p, 8, 4, 4, 1.0, p, p, p, 16, None) == WS and L.w3d_sw_features_fwd_bwd(p, p, p, 20000, 4, 4, 1.0, p, p, p, 1 << 30, None) == UNSUP
    assert L.w3d_sw_project_sort(p, p, 20000, 4, 4, p, None, None) == UNSUP and L.w3d_sw_project_sort(None, p, 8, 4, 4, p, None, None) == ARG and L.w3d_sw_max_points() == 16384
    # round-3 entry points / parameters
    assert L.w3d_knn_prepare(None, None, 10, 10, 64, 5, 1, None, None, 0, None, None, 0, None) == ARG
    assert L.w3d_knn_prepare(p, None, 10, 100, 256, 20, 1, p, None, 0, None, p, 1024, None) == ARG          # a gathered bound needs the style rows
    assert L.w3d_knn_prepare(p, p, 10, 100, 256, 20, 1, None, None, 1, None, p, 16, None) == WS
    assert L.w3d_knn_mfma_set_cu_reserve(-1) == ARG and L.w3d_knn_mfma_set_cu_reserve(200) == ARG
    assert L.w3d_knn_mfma_set_cu_reserve(32) == 0 and L.w3d_knn_mfma_set_cu_reserve(0) == 0
    assert L.w3d_dual_entropic_fwd_bwd(None, None, None, None, None, None, 4, 4, 3, 0.1, 1.0, None, None, None, None, None, None, None) == ARG
    # round-6 (ABI 7) entry points / parameters: straggler pairs, the emit kernel's event hook, the self-cleaning raster backward
    big = L.w3d_knn_workspace_bytes(1000, 4096, 256, 20, 1)
    assert L.w3d_knn_prune_plan_err(p, 1000, 4096, 256, 20, 1, None, p, p, p, 1.0, 0.0, None, None, None, None, 16 << 8, p, big, None) == ARG       # a straggler threshold needs the cached bounds
    assert L.w3d_knn_prune_plan_err(p, 1000, 4096, 256, 20, 1, None, p, p, p, 1.0, 0.0, p, p, p, p, 1 << 16, p, big, None) == ARG                   # bits beyond the threshold byte
    assert L.w3d_knn_match_prepared_lb(p, p, 1000, 4096, 256, 20, 1, 4 | 0x100, p, p, None, None, None, 1.0, p, big, None) == ARG                  # straggler bit without the bounds
    assert L.w3d_knn_match_prepared_lb(p, p, 1000, 4096, 256, 20, 1, 3 | 0x100, p, p, None, p, p, 1.0, p, big, None) == ARG                        # the bounds go with warm = 4 only
    assert L.w3d_knn_match_prepared_lb(p, p, 1000, 4096, 256, 20, 1, 4 | 0x200, p, p, None, p, p, 1.0, p, big, None) == ARG                        # unknown bits
    assert L.w3d_knn_straggler_capacity(1_000_000, 65536) == 8192 and L.w3d_knn_straggler_capacity(1000, 65536) == 256 and L.w3d_knn_straggler_capacity(1_000_000, 1_000_000) == 1024
    off = L.w3d_knn_straggler_counts_offset(1_000_000, 65536, 256, 20)
    assert 0 < off < L.w3d_knn_workspace_bytes(1_000_000, 65536, 256, 20, 1) - 1024 * 8192 * 4 + 4096 and off % 256 == 0
    assert L.w3d_knn_mfma_debug_events(p, None) == ARG and L.w3d_knn_mfma_debug_events(None, None) == 0
    gin = _lib.RasterGradIn(); gout = _lib.RasterGradOut()
    assert L.w3d_raster_backward_selfclean(C.byref(rin), p, C.byref(gin), C.byref(gout), p, p, p, 0, 0, None) == ARG                               # no incoming gradients / accumulators
    for code, text in ((0, b"ok"), (-1, b"invalid"), (-2, b"HIP"), (-3, b"workspace"), (-4, b"unsupported"), (-99, b"unknown")):
        assert text in L.w3d_strerror(code)


def test_knn_launch_plan_is_host_logic():
    """w3d_knn_mfma_launches (how a sweep is cut into launches) is pure host code: callable without a GPU.  1M rows = 1954 blocks of 512 rows = 7 full
    rounds of 256 + a partial one (launched apart, style range split) -> 2; an exact multiple of 256 blocks -> 1; fewer than 256 blocks -> 1 launch
    (the lone partial round; split over the style range only under a warm-start bound, which the count assumes)."""
    from wast3d_amd._lib import lib
    L = lib()
    assert L.w3d_knn_mfma_launches(1_000_000, 65536) == 2
    assert L.w3d_knn_mfma_launches(512 * 256, 65536) == 1
    assert L.w3d_knn_mfma_launches(100_000, 100_000) == 1
    assert L.w3d_knn_mfma_launches(512 * 256 * 3 + 1, 4096) == 2


def test_cache_budget_rule():
    """ops.cache_budget_ok: the rule that keeps the loop's pure performance caches (KnnBounds, NeighbourSumCache) from taking a device down."""
    from wast3d_amd import ops
    assert ops.cache_budget_ok(10, 100, "cpu") and not ops.cache_budget_ok(101, 100, "cpu") and not ops.cache_budget_ok(1, -1, "cpu")
    assert ops.cache_budget_ok(1 << 40, 0, "cpu")                      # no device to ask: allowed (the allocation itself may still fail and is caught)
    # sizes the docs quote: C3 2 GB of bounds + 64 MB of summaries, M = 1e6 about 32 GB
    assert 2.0e9 < ops.KnnBounds.bytes_needed(1_000_000, 65536) < 2.2e9
    assert 31e9 < ops.KnnBounds.bytes_needed(1_000_000, 1_000_000) < 34e9
    assert ops.NeighbourSumCache.bytes_needed(1_000_000, 256, 20) == 1_000_000 * (1024 + 80)
