"""The fused backward of the C3 loop alone, with and without the neighbour-sum cache, at the loop's own state: usage sum_cache_time.py [iterations=40]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
from wast3d_amd.style_loop import make_synthetic_c3
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 40
loop = make_synthetic_c3(style_density="matched", lists="exact", timers=True)
for _ in range(n0):
    loop.step()
torch.cuda.synchronize()
N, M, k = loop.N, loop.M, loop.k
scale = loop.w_nn / (float(N) * float(M))
fya = loop._fy_absmax()
sc = loop._sum_cache
hit = float((sc.idx.sort(1).values == loop.nn_idx.sort(1).values).all(1).float().mean())
print(f"rows whose list is the cache's (as a set) after {n0} iterations: {hit:.3f}")
def t(cache, reps=10):
    gx = torch.zeros(N, 3, device="cuda"); loss = torch.zeros(1, device="cuda")
    ops.raw_knn_descriptor_bwd_fused(loop.xyz4, loop.idx, loop.FX, loop.FY, fya, loop.nn_idx, scale, gx, loop.max_fanin, nn_dist2=loop.nn_d2, loss_accum=loss, slots=loop.slots, sum_cache=cache)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ops.raw_knn_descriptor_bwd_fused(loop.xyz4, loop.idx, loop.FX, loop.FY, fya, loop.nn_idx, scale, gx, loop.max_fanin, nn_dist2=loop.nn_d2, loss_accum=loss, slots=loop.slots, sum_cache=cache)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
print(f"fused backward alone: uncached {t(None):.3f} ms, cache all hits {t(sc):.3f} ms")
sc.idx[::7] = -1
x = t(sc, reps=1)
print(f"(one pass with 1/7 of the rows invalidated: {x:.3f} ms)")

