"""Driver for scripts/cold_pmc.sh: a few COLD emit sweeps and a few WARM ones (one item per row block: tuning bits 28-30 = 1) at BASELINE config C2's shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
N = M = 100_000; d = 128; k = 20
X = torch.randn(N, d, generator=torch.Generator().manual_seed(10)).to(torch.bfloat16).cuda()
Y = torch.randn(M, d, generator=torch.Generator().manual_seed(11)).to(torch.bfloat16).cuda()
ws = ops.KnnWorkspace()
idx = torch.empty(N, k, dtype=torch.int32, device="cuda"); d2 = torch.empty(N, k, device="cuda")
for _ in range(4):
    w = ops.raw_knn_prepare(X, Y, k, ws=ws)
    ops.raw_knn_match_prepared(X, Y, k, idx, d2, w, ws=ws)
torch.cuda.synchronize()
idx0 = idx.clone()
pr = ops.KnnPrune(Y, meb_iters=10)
ops.knn_set_tuning(mode=1 << 28)
for _ in range(4):
    ops.raw_knn_prepare(X, Y, k, prev_idx=idx0, ws=ws)
    ops.raw_knn_prune_plan(X, k, pr, ws)
    ops.raw_knn_match_prepared(X, Y, k, idx, d2, 4, ws=ws)
torch.cuda.synchronize()
ops.knn_set_tuning(mode=0)
assert torch.equal(idx, idx0)
