"""simple_knn._C.distCUDA2 (call site /root/reference/scene/gaussian_model.py:170)."""
import torch

from . import _lib as L


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """(P,3) float32 device tensor -> (P,) mean squared distance to the 3 nearest neighbours."""
    pts = L.require_cuda_f32("points", points, (points.shape[0], 3))
    P = pts.shape[0]
    out = torch.empty(P, dtype=torch.float32, device=pts.device)
    ws = torch.empty(L.lib().riggs_knn_workspace_bytes(P), dtype=torch.uint8, device=pts.device)
    L.check(L.lib().riggs_dist2_knn3(P, pts.data_ptr(), out.data_ptr(), ws.data_ptr(), L.stream_ptr()), "riggs_dist2_knn3")
    return out
