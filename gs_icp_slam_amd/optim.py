"""Multi-tensor Adam in one HIP launch: a drop-in for the `torch.optim.Adam(l, lr=0.0, eps=1e-15)`
that GaussianModel builds [REF scene/gaussian_model.py:222-231] and the mapper steps [REF mp_Mapper.py:247].

State layout is torch's (`state[p] = {"step", "exp_avg", "exp_avg_sq"}`), because GaussianModel edits the optimiser state
in place when it appends or prunes Gaussians [REF scene/gaussian_model.py:409-492].  amsgrad / weight decay / maximize are
not supported (the reference does not use them).
"""
import ctypes

import torch

from . import _lib

_MAX = 8


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        buckets = {}   # (beta1, beta2, eps, step, device) -> list of (p, g, m, v, lr)
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam (gfx950): parameters must live on the HIP device; there is no CPU path")
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdam: parameters must be contiguous float32")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] = int(st["step"]) + 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                buckets.setdefault((float(b1), float(b2), float(group["eps"]), st["step"], p.device), []).append(
                    (p, g.to(torch.float32), st["exp_avg"], st["exp_avg_sq"], float(group["lr"])))
        for (b1, b2, eps, step, dev), items in buckets.items():
            with torch.cuda.device(dev):
                stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                for i in range(0, len(items), _MAX):
                    chunk = items[i:i + _MAX]
                    n = len(chunk)
                    P = (ctypes.c_void_p * n)(*[t[0].data_ptr() for t in chunk])
                    G = (ctypes.c_void_p * n)(*[t[1].data_ptr() for t in chunk])
                    M = (ctypes.c_void_p * n)(*[t[2].data_ptr() for t in chunk])
                    V = (ctypes.c_void_p * n)(*[t[3].data_ptr() for t in chunk])
                    N = (ctypes.c_longlong * n)(*[t[0].numel() for t in chunk])
                    LR = (ctypes.c_float * n)(*[t[4] for t in chunk])
                    _lib.check(lib.gsicp_adam_step(n, P, G, M, V, N, LR, b1, b2, eps, step, stream), "gsicp_adam_step")
        return loss
