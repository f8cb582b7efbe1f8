"""GaussianModel's activation getters as one HIP launch each way (with the loss and Adam):

    opacity, scaling, rotation = activate(gaussians._opacity, gaussians._scaling, gaussians._rotation)

replaces `pc.get_opacity`, `pc.get_scaling`, `pc.get_rotation` in render_3 [REF gaussian_renderer/__init__.py:263, 273-274;
scene/gaussian_model.py:44-56, 105-125] — sigmoid, exp, torch.nn.functional.normalize and their autograd, ~20 torch
launches per iteration otherwise.
"""
import ctypes

import torch

from . import _lib


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class _Activate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, opacity_raw, scaling_raw, rotation_raw, live_rows=None):
        lib = _lib.load()
        if not opacity_raw.is_cuda:
            raise RuntimeError("activate (gfx950): tensors must live on the HIP device; there is no CPU path")
        dev = opacity_raw.device
        f = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        o_r, s_r, q_r = f(opacity_raw), f(scaling_raw), f(rotation_raw)
        P = q_r.shape[0]
        if o_r.numel() != P or s_r.numel() != 3 * P or q_r.numel() != 4 * P:
            raise RuntimeError("activate: expected _opacity (P,1), _scaling (P,3), _rotation (P,4)")
        o, s, q = torch.empty_like(o_r), torch.empty_like(s_r), torch.empty_like(q_r)
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.gsicp_mapper_activations_forward(P, _p(o_r), _p(s_r), _p(q_r), _p(o), _p(s), _p(q), _p(live_rows), stream),
                       "gsicp_mapper_activations_forward")
        ctx.save_for_backward(o, s, q_r, live_rows)
        return o, s, q

    @staticmethod
    def backward(ctx, g_o, g_s, g_q):
        lib = _lib.load()
        o, s, q_r, live_rows = ctx.saved_tensors
        dev = o.device
        P = q_r.shape[0]
        need = ctx.needs_input_grad
        c = lambda g: None if g is None else g.to(dtype=torch.float32).contiguous()
        g_o, g_s, g_q = c(g_o), c(g_s), c(g_q)
        d_o = torch.empty_like(o) if need[0] else None
        d_s = torch.empty_like(s) if need[1] else None
        d_q = torch.empty_like(q_r) if need[2] else None
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.gsicp_mapper_activations_backward(P, _p(o), _p(s), _p(q_r), _p(g_o), _p(g_s), _p(g_q), _p(d_o), _p(d_s), _p(d_q),
                                                             _p(live_rows), stream), "gsicp_mapper_activations_backward")
        return d_o, d_s, d_q, None


def activate(opacity_raw, scaling_raw, rotation_raw, live_rows=None):
    """-> (sigmoid(opacity_raw), exp(scaling_raw), normalize(rotation_raw)), differentiable.
    live_rows: optional int32[1] DEVICE tensor — only the first live_rows[0] rows are Gaussians (capacity-backed map: the inputs are
    the full-capacity buffers, their live count changes on the device without changing any pointer; outputs of the other rows are
    unspecified, their gradients zero)."""
    return _Activate.apply(opacity_raw, scaling_raw, rotation_raw, live_rows)
