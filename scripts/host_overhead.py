"""Host time to ENQUEUE one step of the C3 loop (no synchronisation inside the measurement except the loop's own host read) against the GPU time of the step (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gc
from wast3d_amd.style_loop import make_synthetic_c3
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
loop = make_synthetic_c3(N=n, style_density="matched", lists="exact")
for _ in range(5):
    loop.step()
torch.cuda.synchronize(); gc.collect(); gc.freeze(); gc.disable()
ts = []
t_all = time.perf_counter()
for _ in range(20):
    t0 = time.perf_counter(); loop.step(); ts.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize(); total = (time.perf_counter() - t_all) / 20 * 1e3
print(f"N = {n}: host time inside step() min {min(ts):.2f} / median {sorted(ts)[10]:.2f} ms (includes waiting for the render forward's instance count); wall per step {total:.2f} ms")
