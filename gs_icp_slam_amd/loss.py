"""Fused mapping loss: what mp_Mapper.py computes with `l1_loss`, `ssim` and a depth `l1_loss`
[REF mp_Mapper.py:225-240; utils/loss_utils.py:17-69], forward and gradient, in three HIP launches.

    loss = mapper_loss(image, depth_image, gt_image, gt_depth_image, lambda_dssim=0.2)   # replaces REF mp_Mapper.py:225-240
    loss.backward()

`gt_image` is the UNMASKED ground-truth image; the `gt_image * (gt_depth > 0)` masking of [REF mp_Mapper.py:225-228] is
done inside.  Returns a 0-dim tensor; `mapper_loss.parts(...)` also returns (L1, SSIM mean, depth L1).
"""
import ctypes

import torch

from . import _lib


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class _MapperLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, depth, gt_image, gt_depth, lambda_dssim, depth_weight, d_max):
        lib = _lib.load()
        if not image.is_cuda:
            raise RuntimeError("mapper_loss (gfx950): tensors must live on the HIP device; there is no CPU path")
        dev = image.device
        f = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        image_c, depth_c, gt_c, gtd_c = f(image), f(depth), f(gt_image), f(gt_depth)
        H, W = image_c.shape[-2], image_c.shape[-1]
        if image_c.numel() != 3 * H * W or depth_c.numel() != H * W or gt_c.numel() != 3 * H * W or gtd_c.numel() != H * W:
            raise RuntimeError("mapper_loss: expected image/gt_image (3,H,W) and depth/gt_depth (1,H,W)")
        need_grad = image.requires_grad or depth.requires_grad
        with torch.cuda.device(dev):
            out = torch.empty(4, dtype=torch.float32, device=dev)
            g_img = torch.empty_like(image_c) if need_grad else None
            g_dep = torch.empty_like(depth_c) if need_grad else None
            scratch = torch.empty(int(lib.gsicp_mapper_loss_scratch_bytes(W, H)), dtype=torch.uint8, device=dev)
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.gsicp_mapper_loss(_p(image_c), _p(depth_c), _p(gt_c), _p(gtd_c), W, H, float(lambda_dssim), float(depth_weight),
                                             float(d_max), _p(out), _p(g_img), _p(g_dep), _p(scratch), stream), "gsicp_mapper_loss")
        ctx.save_for_backward(g_img, g_dep)
        ctx.shapes = (image.shape, depth.shape)
        ctx.mark_non_differentiable(out)
        return out[0], out

    @staticmethod
    def backward(ctx, g_loss, _g_parts):
        g_img, g_dep = ctx.saved_tensors
        if g_img is None:
            return (None,) * 7
        return (g_img * g_loss).view(ctx.shapes[0]), (g_dep * g_loss).view(ctx.shapes[1]), None, None, None, None, None


def mapper_loss(image, depth, gt_image, gt_depth, lambda_dssim=0.2, depth_weight=0.1, d_max=10.0):
    return _MapperLoss.apply(image, depth, gt_image, gt_depth, lambda_dssim, depth_weight, d_max)[0]


def mapper_loss_parts(image, depth, gt_image, gt_depth, lambda_dssim=0.2, depth_weight=0.1, d_max=10.0):
    """-> (loss, tensor([loss, L1, SSIM mean, depth L1]))"""
    return _MapperLoss.apply(image, depth, gt_image, gt_depth, lambda_dssim, depth_weight, d_max)


def mapper_loss_and_grads(image, depth, gt_image, gt_depth, lambda_dssim=0.2, depth_weight=0.1, d_max=10.0, tile_mod=1, tile_rem=0, gt_slots=None,
                          step_bump=None):
    """No-autograd form for callers that drive the backward themselves (gs_icp_slam_amd/graph.py):
    -> (tensor([loss, L1, SSIM mean, depth L1]), dL/dimage (3,H,W), dL/ddepth (1,H,W)), the gradients being those of `loss`
    itself, so no ones_like / multiply launches are needed before `torch.autograd.backward((image, depth), (g_image, g_depth))`.
    tile_mod > 1 (multi-GPU mapper): this rank's 32x32 blocks only (gsicp_mapper_loss_sharded) — the four values are this rank's SHARE
    (their sum over the ranks is the loss) and the gradients are defined on the rank's own blocks (zero elsewhere).
    gt_slots (int64[2] DEVICE tensor holding the addresses of the ground-truth image and depth, written by gsicp_mapper_select_view): the kernels
    read the two pointers on the device when they start (gsicp_mapper_loss_indirect) — a captured iteration then follows the keyframe selection
    without any image being copied; gt_image / gt_depth are ignored (may be None).
    step_bump (with gt_slots; round 6): (step int32 device tensor, guard count tensor or None, guard limit, skipped-steps tensor or None) — the thread that
    finishes the loss value also advances the optimiser's device step counter (or, under a tripped guard, the skipped counter): the Adam launch that
    follows in the same stream order is then issued with `FusedAdam.step(step_already_bumped=True)` and no one-thread bump launch (gsicp_mapper_loss_indirect_bump)."""
    lib = _lib.load()
    if not image.is_cuda:
        raise RuntimeError("mapper_loss (gfx950): tensors must live on the HIP device; there is no CPU path")
    dev = image.device
    f = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
    image_c, depth_c = f(image), f(depth)
    H, W = image_c.shape[-2], image_c.shape[-1]
    if image_c.numel() != 3 * H * W or depth_c.numel() != H * W:
        raise RuntimeError("mapper_loss: expected image (3,H,W) and depth (1,H,W)")
    if gt_slots is None:
        gt_c, gtd_c = f(gt_image), f(gt_depth)
        if gt_c.numel() != 3 * H * W or gtd_c.numel() != H * W:
            raise RuntimeError("mapper_loss: expected image/gt_image (3,H,W) and depth/gt_depth (1,H,W)")
    elif not (gt_slots.is_cuda and gt_slots.dtype == torch.int64 and gt_slots.numel() == 2 and gt_slots.is_contiguous()):
        raise RuntimeError("mapper_loss: gt_slots must be a contiguous int64[2] device tensor")
    with torch.cuda.device(dev):
        out = torch.empty(4, dtype=torch.float32, device=dev)
        sharded = int(tile_mod) > 1
        g_img, g_dep = (torch.zeros_like(image_c), torch.zeros_like(depth_c)) if sharded else (torch.empty_like(image_c), torch.empty_like(depth_c))
        scratch = torch.empty(int(lib.gsicp_mapper_loss_scratch_bytes(W, H)), dtype=torch.uint8, device=dev)
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if step_bump is not None:
            if gt_slots is None:
                raise RuntimeError("mapper_loss: step_bump needs gt_slots (the captured iteration's loss)")
            step_t, guard_t, guard_lim, skipped_t = step_bump
            _lib.check(lib.gsicp_mapper_loss_indirect_bump(_p(image_c), _p(depth_c), _p(gt_slots), W, H, float(lambda_dssim), float(depth_weight), float(d_max),
                                                           int(tile_mod), int(tile_rem), _p(out), _p(g_img), _p(g_dep), _p(scratch), _p(step_t), _p(guard_t),
                                                           int(guard_lim), _p(skipped_t), stream), "gsicp_mapper_loss_indirect_bump")
        elif gt_slots is not None:
            _lib.check(lib.gsicp_mapper_loss_indirect(_p(image_c), _p(depth_c), _p(gt_slots), W, H, float(lambda_dssim), float(depth_weight), float(d_max),
                                                      int(tile_mod), int(tile_rem), _p(out), _p(g_img), _p(g_dep), _p(scratch), stream),
                       "gsicp_mapper_loss_indirect")
        elif sharded:
            _lib.check(lib.gsicp_mapper_loss_sharded(_p(image_c), _p(depth_c), _p(gt_c), _p(gtd_c), W, H, float(lambda_dssim), float(depth_weight),
                                                     float(d_max), int(tile_mod), int(tile_rem), _p(out), _p(g_img), _p(g_dep), _p(scratch), stream),
                       "gsicp_mapper_loss_sharded")
        else:
            _lib.check(lib.gsicp_mapper_loss(_p(image_c), _p(depth_c), _p(gt_c), _p(gtd_c), W, H, float(lambda_dssim), float(depth_weight),
                                             float(d_max), _p(out), _p(g_img), _p(g_dep), _p(scratch), stream), "gsicp_mapper_loss")
    return out, g_img.view(image.shape), g_dep.view(depth.shape)
