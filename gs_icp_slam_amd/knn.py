"""Host-side mirror of ``simple_knn._C`` [REF scene/gaussian_model.py:20]: ``distCUDA2(points) -> (P,) float32``."""
import ctypes

import torch

from . import _lib


def distCUDA2(points):
    lib = _lib.load()
    if not points.is_cuda:
        raise RuntimeError("simple_knn.distCUDA2 (gfx950): points must live on the HIP device; there is no CPU path")
    pts = points.detach().to(torch.float32).contiguous().view(-1, 3)
    out = torch.empty((pts.shape[0],), dtype=torch.float32, device=pts.device)
    with torch.cuda.device(pts.device):
        stream = ctypes.c_void_p(torch.cuda.current_stream(pts.device).cuda_stream)
        _lib.check(lib.gsicp_knn_dist2(pts.shape[0], ctypes.c_void_p(pts.data_ptr()), ctypes.c_void_p(out.data_ptr()), stream),
                   "gsicp_knn_dist2")
    return out
