"""Per-rank COMPUTE cost of the strong-scaling loop at world size W, measured on ONE GPU: the collectives of torch.distributed are replaced by local stand-ins
("this rank's block goes where it belongs, the other ranks' blocks keep the real data of the initial scene; a reduce-scatter returns the own block; an
all-reduce is the identity"), so every kernel runs on the data volumes it would see as rank r of W — all P_total splat records are binned for the rank's
band, the halo positions are real, the style set is whole.  What it does NOT contain: transfer time, and the other ranks' share of the image.
    python scripts/rank_sim.py [W=8] [rank=3] [style=cached|inloop] [M=65536]
Prints the step time and the stage times; DESIGN 4 quotes them as the compute floor of the 8-GPU step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
style = sys.argv[3] if len(sys.argv) > 3 else "cached"
M = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
N = 1_000_000

def _ag(out, inp, group=None):
    n = inp.shape[0]
    out[R * n:(R + 1) * n].copy_(inp)
def _rs(out, full, group=None):
    n = out.shape[0]
    out.copy_(full[R * n:(R + 1) * n])
dist.get_world_size = lambda group=None: W
dist.get_rank = lambda group=None: R
dist.get_backend = lambda group=None: "nccl"
dist.all_gather_into_tensor = _ag
dist.reduce_scatter_tensor = _rs
dist.all_reduce = lambda t, group=None, op=None: None
dist.broadcast = lambda t, src=0, group=None: None
dist.new_group = lambda backend=None: None
dist.barrier = lambda: None

from wast3d_amd import ops
from wast3d_amd.distributed import make_synthetic_c4
from wast3d_amd.style_loop import make_synthetic_c3

DENS = os.environ.get("STYLE", "matched")                            # the bench's default style patch (the hard case); STYLE=sparse: the easy one
loop = make_synthetic_c4(n_total=N, M=M, timers=True, style=style, lists="exact", style_density=DENS)
b = loop.b
# the other ranks' blocks: real data of the initial scene (positions, splat records of the first view)
whole = make_synthetic_c3(N=N, M=M, lists="exact", style_density=DENS)
ops.raw_pack_xyz4(whole.pc._xyz.data, b.xyz4)
if style == "cached":
    b.FY = whole.FY; b.M = b.M_true = whole.FY.shape[0]
else:
    for c in range(W):
        s0, s1 = loop.plan.style_ranges[c]; m0, _, _ = loop.plan.style_chunk_rows(c)
        b.FY[m0:m0 + (s1 - s0)] = whole.FY[s0:s1]
whole.step(); torch.cuda.synchronize()
import ctypes as C
geom = whole._rb["geom"]
loop.records_all.copy_(geom[: N * 48].view(torch.float32).view(N, 12))
del whole
if os.environ.get("KNN_MODE"):
    ops.knn_set_tuning(mode=int(os.environ["KNN_MODE"], 0))
if os.environ.get("CU_RESERVE"):                                    # A/B: CUs the sweep leaves to the render leg (default: the loop's own choice)
    loop.cu_reserve = loop.b.cu_reserve = int(os.environ["CU_RESERVE"])
print(f"  CUs reserved for the render leg during the sweep: {loop.cu_reserve}")
if os.environ.get("RANKSIM_TRACE"):                                  # where does a slow iteration spend its time?  (synchronising wrappers: A/B only)
    def _wrap(obj, name):
        f = getattr(obj, name)
        def g(*a, **k):
            torch.cuda.synchronize(); t = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize()
            dt = (time.perf_counter() - t) * 1e3
            if dt > 5.0:
                print(f"    {name}: {dt:.1f} ms (iteration {loop._iters})")
            return r
        setattr(obj, name, g)
    for nm in ("_recut_bands", "_render_forward", "_render_backward"):
        _wrap(loop, nm)
    _wrap(b, "_geometry_leg")
    _wrap(loop.plan, "set_bands")
for _ in range(4):
    loop.step()
loop.knn_events = []; b.stage_ms()
steps = 24
per = []
mem = []
if os.environ.get("RANKSIM_GC", "1") == "0":                         # the hiccup hunt: is the one slow iteration in 24 the cyclic garbage collector?
    import gc
    gc.collect(); gc.disable()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    t1 = time.perf_counter(); loop.step(); torch.cuda.synchronize(); per.append((time.perf_counter() - t1) * 1e3)
    mem.append((torch.cuda.memory_reserved() >> 20, loop.last_R, tuple(loop.plan.band)))
dt = (time.perf_counter() - t0) / steps * 1e3
print("  per-iteration wall ms (synchronised after each):", [round(x, 2) for x in per])
slow = [i for i, x in enumerate(per) if x > 2 * sorted(per)[len(per) // 2]]
for i in slow:                                                      # a hiccup: did the caching allocator go to the device (hipMalloc: tens of ms)?
    print(f"  slow iteration {i}: {per[i]:.1f} ms; (reserved MB, R of the band, band) before {mem[i - 1] if i else None} after {mem[i]}")
st = b.stage_ms()
print(f"rank {R} of {W} (compute only, style {style}, M {M}): step median {sorted(per)[len(per) // 2]:.2f} ms, mean {dt:.2f} ms  knn {loop.knn_ms():.2f}  bands {loop.plan.bands}  R_band {loop.last_R}")
print("  stages in loop:", {k: round(v, 2) for k, v in st.items()})

# ---- the same stages ALONE on the chip (no overlap between the two streams)
def timed(fn, n=5):
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)
xyz = b.pc._xyz.data
b.stage_ms()
t_geo = timed(lambda: b._geometry_leg(xyz))
geo = b.stage_ms()
t_rf = timed(lambda: loop._render_forward(xyz))
t_rb = timed(lambda: loop._render_backward(xyz))
print(f"  alone: geometry leg {t_geo:.2f} ms {dict((k, round(v, 2)) for k, v in geo.items())}; render forward {t_rf:.2f}; render backward {t_rb:.2f}")
