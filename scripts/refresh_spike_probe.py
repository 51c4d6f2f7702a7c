import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, gc
from wast3d_amd.style_loop import make_synthetic_c3
loop = make_synthetic_c3(N=1_000_000, M=65536, d=256, k=20, lists="exact", style_density="matched", timers=True)
gc.collect(); gc.freeze()
n = 200
evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
host = []
evs[0].record()
for i in range(n):
    t0 = time.perf_counter(); loop.step(); host.append((time.perf_counter() - t0) * 1e3); evs[i + 1].record()
torch.cuda.synchronize()
ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
knn = loop.knn_ms_per_step()
for c in (32, 64, 96, 128):
    print("around iteration", c, " gpu ms:", [round(x, 1) for x in ms[c - 3:c + 4]], " host ms:", [round(x, 1) for x in host[c - 3:c + 4]], " knn:", [round(x, 1) for x in knn[c - 3:c + 4]])
import statistics
print("median", statistics.median(ms), "mean", sum(ms[1:]) / (n - 1))
