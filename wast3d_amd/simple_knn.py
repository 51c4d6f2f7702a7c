"""Drop-in for the reference's `simple_knn._C.distCUDA2` (submodules/simple-knn, used once at initialisation by
scene/gaussian_model.py:134): the mean SQUARED distance of every point to its 3 nearest neighbours.

    from wast3d_amd.simple_knn import distCUDA2

Computed with the exact large-K list kernel of csrc/pointcloud.hip (K = 4: the point itself + 3 neighbours); the reference's
extension finds the same 3 neighbours with a Morton-ordered box search."""
import torch

from . import ops


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """points [P,3] f32 on the GPU -> [P] f32.  Fewer than 4 points: the mean runs over the neighbours that exist (0 for a single point)."""
    P = points.shape[0]
    if P == 0:
        return torch.zeros(0, dtype=torch.float32, device=points.device)
    K = min(4, P)
    if K == 1:
        return torch.zeros(P, dtype=torch.float32, device=points.device)
    pts = points.contiguous().float()
    idx = ops.knn_lists(pts, K)
    # squared distances from the coordinate differences (as simple-knn does), not from the matmul form the list kernel ranks with:
    # at initialisation the neighbours are very close and |x|^2 + |y|^2 - 2 x.y would lose their leading digits
    diff = pts[idx[:, 1:].long()] - pts[:, None, :]
    return (diff * diff).sum(-1).sum(1) / float(K - 1 if K < 4 else 3)
