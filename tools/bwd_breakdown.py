import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gs_icp_slam_amd import rasterizer as R, _lib
orig = R._RasterizeGaussians.backward
T = {}
def timed_backward(ctx, *grads):
    t0 = time.perf_counter()
    out = orig(ctx, *grads)
    T["my_backward"] = T.get("my_backward", 0) + time.perf_counter() - t0
    return out
R._RasterizeGaussians.backward = staticmethod(timed_backward)
lib = _lib.load()
real = lib.gsicp_raster_backward
class Wrap:
    def __call__(self, *a):
        t0 = time.perf_counter(); r = real(*a); T["c_call"] = T.get("c_call", 0) + time.perf_counter() - t0; return r
lib.gsicp_raster_backward = Wrap()
_e = torch.empty
def te(*a, **k):
    t0 = time.perf_counter(); r = _e(*a, **k); T["torch.empty"] = T.get("torch.empty", 0) + time.perf_counter() - t0; return r
R.torch.empty = te
exec(open(os.path.join(os.path.dirname(__file__), "host_breakdown.py")).read().split("T = {}")[0])
T.clear()
N = 40
for it in range(N + 5):
    if it == 5: T.clear()
    m2 = torch.zeros_like(params["means3D"], requires_grad=True)
    d, c, r, u = rast(means3D=params["means3D"], means2D=m2, shs=params["shs"], opacities=params["opacities"], scales=params["scales"], rotations=params["rotations"])
    loss = (c - gt_c).abs().mean() + 0.1 * ((d - gt_d) / 10).abs().mean()
    torch.cuda.synchronize(); t0 = time.perf_counter(); loss.backward(); T["bwd.host"] = T.get("bwd.host", 0) + time.perf_counter() - t0
    for p in params.values(): p.grad = None
for k, v in T.items(): print(f"{k:16s} {1e6 * v / N:9.1f} us/step")
