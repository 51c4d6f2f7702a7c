"""Swept fraction / sweep time / losses of the synthetic C3 loop along a run, for several weights of the NN style term (at w_nn = 1 the NN term is
3e-6 of the loss and the content drifts away from the style set).  usage: loss_balance_probe.py [w_nn ...]   default 1 1e5 1e6"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
from wast3d_amd.style_loop import make_synthetic_c3
ws = [float(x) for x in sys.argv[1:]] or [1.0, 1e5, 1e6]
marks = (1, 5, 20, 60, 150, 300, 600)
for w in ws:
    loop = make_synthetic_c3(style_density="matched", lists="exact", timers=True, w_nn=w)
    N, M, d, k = loop.N, loop.M, loop.d, loop.k
    pairs = ((N + 511) // 512) * ((M + 63) // 64)
    it = 0
    for m in marks:
        while it < m - 1:
            loop.step(); it += 1
        loop.knn_events = []; loop.swept_log = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loop.step(); it += 1
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        kms = loop.knn_ms_per_step(); sw = loop.swept_pairs_per_step()
        l = loop.loss.tolist()
        print(f"w_nn {w:g} it {it:4d}: step {dt:6.2f} ms, knn {kms[0] if kms else 0:6.2f} ms, swept {sw[0] / pairs if sw else 1.0:.4f}, k-th distance median {float(loop.nn_d2[:, -1].clamp_min(0).sqrt().median()):.4g}, "
              f"losses weighted nn {l[0]:.3e} depth {l[1]:.3e} tv {l[2]:.3e}", flush=True)
    del loop
    torch.cuda.empty_cache()
