"""A/B of the data-layout hints on the C3 workload (same device, same process): style rows in tree order on/off x content-row processing order on/off,
kNN time per step from HIP events, for the matched and sparse style patches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd.style_loop import make_synthetic_c3
from wast3d_amd import ops
combos = [("1", "1", "prev", "0"), ("1", "1", "half", "0"), ("1", "1", "prev", "44"), ("1", "1", "prev", "56"), ("1", "1", "half", "48")] if len(sys.argv) < 2 else [tuple(a.split(",")) for a in sys.argv[1:]]
for style in ("matched",):
    for so, ro, warm, vote in combos:
        os.environ["W3D_STYLE_ORDER"] = so; os.environ["W3D_ROW_ORDER"] = ro
        ops.knn_set_tuning(mode=int(vote) << 16)
        loop = make_synthetic_c3(timers=True, style_density=style, warm=warm)
        for _ in range(3):
            loop.step()
        loop.knn_events = []; loop.stage_ms()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8):
            loop.step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8 * 1e3
        st = loop.stage_ms()
        print(f"{style} style_order={so} row_order={ro} warm={warm} vote_at={vote}: step {dt:.2f} ms  knn {loop.knn_ms():.2f} ms  bound {st.get('bound', 0):.2f}  fused_bwd {st.get('fused_bwd', 0):.2f}  warm={loop._warm_mode}", flush=True)
        del loop; torch.cuda.empty_cache()
