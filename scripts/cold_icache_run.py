"""Driver: 4 cold emit sweeps in slices (all CUs busy), then 4 unsliced (tuning bit 0x1000: 196 of 256 CUs busy), C2's shape — for a --pmc pass over the instruction cache."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
N = M = 100_000; d = 128; k = 20
X = torch.randn(N, d, generator=torch.Generator().manual_seed(10)).to(torch.bfloat16).cuda()
Y = torch.randn(M, d, generator=torch.Generator().manual_seed(11)).to(torch.bfloat16).cuda()
ws = ops.KnnWorkspace()
idx = torch.empty(N, k, dtype=torch.int32, device="cuda"); d2 = torch.empty(N, k, device="cuda")
for mode in (0, 0x1000):
    ops.knn_set_tuning(mode=mode)
    for _ in range(4):
        w = ops.raw_knn_prepare(X, Y, k, ws=ws)
        ops.raw_knn_match_prepared(X, Y, k, idx, d2, w, ws=ws)
    torch.cuda.synchronize()
ops.knn_set_tuning(mode=0)
