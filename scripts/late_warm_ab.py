"""Which warm-start bound late in a run (rows moving by well under 2 % of their k-th distance per iteration)?  C3 loop after many iterations, 20 timed steps per mode (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd.style_loop import make_synthetic_c3
loop = make_synthetic_c3(style_density="matched", lists="exact", timers=True, gemm=os.environ.get("GEMM", "same"))
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 600
for i in range(n0):
    loop.step()
torch.cuda.synchronize()
print("after", n0, "iterations: auto mode =", loop._warm_mode)
for mode in (("motion", "prev", "motion", "prev") if os.environ.get("GEMM") == "fp8" else ("motion", "prev", "half", "motion", "prev")):
    loop.warm = mode
    for _ in range(3):
        loop.step()
    loop.knn_events = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        loop.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
    print(f"  warm={mode}: step {dt:.2f} ms  knn {loop.knn_ms():.2f}")
