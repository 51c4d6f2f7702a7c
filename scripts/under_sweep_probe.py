"""Can a lean HBM-streaming kernel (<= 32 VGPRs, no LDS: scripts/micro/stream_under_sweep.hip) run UNDER the persistent emit sweep, which leaves HBM idle and 32 VGPRs per
SIMD lane free?  C3 loop at iteration n0: the sweep alone, the stream alone (1.6 GB, the size of the 56-float Adam step), and both launched together on two streams.
usage: under_sweep_probe.py [n0=20]   (builds the micro kernel with hipcc into scripts/_ab/)"""
import os, sys, ctypes, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
from wast3d_amd import ops
from wast3d_amd.style_loop import make_synthetic_c3
so = os.path.join(root, "scripts/_ab/stream_under_sweep.so")
if not os.path.exists(so):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, os.path.join(root, "scripts/micro/stream_under_sweep.hip")])
L = ctypes.CDLL(so)
L.stream_launch.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 20
loop = make_synthetic_c3(style_density="matched", lists="exact", timers=True)
for _ in range(n0):
    loop.step()
torch.cuda.synchronize()
buf = torch.zeros(100_000_000, 2, device="cuda")                      # 0.8 GB read + 0.8 GB written
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
idx = torch.empty_like(loop.nn_idx); d2 = torch.empty_like(loop.nn_d2)
def sweep():
    ops.raw_knn_prepare(loop.FX, loop.FY, loop.k, prev_idx=loop.nn_idx, ws=loop.kws, row_order=loop.row_order)
    ops.raw_knn_prune_plan(loop.FX, loop.k, loop.prune, loop.kws, row_order=loop.row_order, bounds=loop.bounds)
    return ops.raw_knn_match_prepared(loop.FX, loop.FY, loop.k, idx, d2, 4, timed=True, ws=loop.kws, row_order=loop.row_order, bounds=loop.bounds, prune=loop.prune)
def stream(blocks):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        e0.record(); L.stream_launch(buf.data_ptr(), buf.shape[0], blocks, side.cuda_stream); e1.record()
    return e0, e1
for blocks in (1024, 4096):
    for rep in range(3):
        ev = sweep(); torch.cuda.synchronize(); a = ev[0].elapsed_time(ev[1])
        s = stream(blocks); torch.cuda.synchronize(); b = s[0].elapsed_time(s[1])
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        ops.raw_knn_prepare(loop.FX, loop.FY, loop.k, prev_idx=loop.nn_idx, ws=loop.kws, row_order=loop.row_order)
        ops.raw_knn_prune_plan(loop.FX, loop.k, loop.prune, loop.kws, row_order=loop.row_order, bounds=loop.bounds)
        pre = torch.cuda.Event(); pre.record()
        side.wait_event(pre)
        t0.record()
        ev = ops.raw_knn_match_prepared(loop.FX, loop.FY, loop.k, idx, d2, 4, timed=True, ws=loop.kws, row_order=loop.row_order, bounds=loop.bounds, prune=loop.prune)
        s = stream(blocks)
        main.wait_stream(side); t1.record(); torch.cuda.synchronize()
        print(f"{blocks} blocks: sweep alone {a:.3f} ms, stream alone {b:.3f} ms ({1.6 / b:.2f} TB/s); together: sweep {ev[0].elapsed_time(ev[1]):.3f} ms, stream {s[0].elapsed_time(s[1]):.3f} ms, "
              f"both done after {t0.elapsed_time(t1):.3f} ms (one after the other {a + b:.3f})", flush=True)
