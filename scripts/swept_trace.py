"""Per-iteration trace of the prune plan's swept fraction around a station of the C3 loop: usage swept_trace.py [upto=330] [from=250] [loop-config JSON]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd.style_loop import make_synthetic_c3, LoopConfig
upto = int(sys.argv[1]) if len(sys.argv) > 1 else 330
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 250
kw = json.loads(sys.argv[3]) if len(sys.argv) > 3 else {}
loop = make_synthetic_c3(style_density="matched", lists="exact", timers=True, config=LoopConfig(**kw))
ev = []
for i in range(upto):
    loop.step()
    e = torch.cuda.Event(enable_timing=True); e.record(); ev.append(e)
torch.cuda.synchronize()
sw = loop.swept_pairs_per_step()
pairs = ((loop.N + 511) // 512) * ((loop.M + 63) // 64)
fr = [1.0] * (upto - len(sw)) + [s / pairs for s in sw]
for i in range(max(lo, 1), upto):
    print(f"it {i + 1:4d}  swept {fr[i]:.4f}  step {ev[i - 1].elapsed_time(ev[i]):6.2f} ms")
