"""Cold sweeps (no bound) at BASELINE config C2's shape: the round-5 cold EMIT sweep (bound tracked in registers, candidates + select) against the list kernel
(tuning bit 0x8000000), same device, same rows; lists must be equal.  usage: [W3D_LIB_PATH=...] python scripts/cold_emit_ab.py [N=100000] [M=100000] [d=128] [k=20]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
d = int(sys.argv[3]) if len(sys.argv) > 3 else 128
k = int(sys.argv[4]) if len(sys.argv) > 4 else 20
if os.environ.get("CLUSTERS"):                                      # style rows stored cluster by cluster, content rows sorted by cluster: candidates come in storms
    g = torch.Generator().manual_seed(47); nc = int(os.environ["CLUSTERS"])
    cen = torch.randn(nc, d, generator=g)
    Y = (cen[torch.arange(M) * nc // M] + 0.05 * torch.randn(M, d, generator=g)).to(torch.bfloat16).cuda()
    X = (cen[torch.randint(0, nc, (N,), generator=g).sort().values] + 0.05 * torch.randn(N, d, generator=g)).to(torch.bfloat16).cuda()
else:
    X = torch.randn(N, d, generator=torch.Generator().manual_seed(10)).to(torch.bfloat16).cuda()
    Y = torch.randn(M, d, generator=torch.Generator().manual_seed(11)).to(torch.bfloat16).cuda()
ws = ops.KnnWorkspace()
flop = 2.0 * N * M * d
out = {}
for name, mode in (("list kernel (0x8000000)", 0x8000000), ("cold emit", 0)):
    ops.knn_set_tuning(mode=mode)
    idx = torch.empty(N, k, dtype=torch.int32, device="cuda"); d2 = torch.empty(N, k, device="cuda"); evs = []
    for _ in range(5):
        w = ops.raw_knn_prepare(X, Y, k, ws=ws)
        evs.append(ops.raw_knn_match_prepared(X, Y, k, idx, d2, w, timed=True, ws=ws))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)[2]
    rep = ops.knn_repair_count(X, Y, k, ws=ws)
    out[name] = (idx, d2)
    print(f"{N} x {M} x {d}, k = {k}  {name:26s} {ms:7.3f} ms  {flop / ms / 1e9:7.0f} TFLOP/s = {flop / ms / 1e9 / 2500:.3f} of peak   rows redone by the repair: {rep}", flush=True)
ops.knn_set_tuning(mode=0)
a, b = out["list kernel (0x8000000)"], out["cold emit"]
print("lists equal:", bool(torch.equal(a[0], b[0])), "distances equal:", bool(torch.equal(a[1], b[1])), " differing rows:", int((a[0] != b[0]).any(1).sum()))
