import os, sys, time, json
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from wast3d_amd.distributed import make_synthetic_c4
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
loop = make_synthetic_c4(n_total=1_000_000, M=M, d=256, k=20, timers=True, lists="exact", style_density="matched", style="cached")
b = loop.b
def run(n, tag):
    b.stage_ms(); b.knn_events = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): loop.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
    st = b.stage_ms(); knn = b.knn_ms_per_step()
    print(tag, "ms/step %.2f" % dt, "knn %.2f" % (sum(knn) / len(knn)), {k: round(v, 2) for k, v in st.items()}, flush=True)
for _ in range(3): loop.step()
run(4, "real ")
with loop.local_collectives():
    run(4, "local")
run(4, "real ")
with loop.local_collectives():
    run(4, "local")
run(4, "real ")
dist.destroy_process_group()
