"""Host-side mirror of the reference's ``diff_gaussian_rasterization`` Python package, backed by libgsicp_hip.so.

Same names, call keywords, error behaviour and return order as the reference's call sites expect:
  * ``GaussianRasterizationSettings`` — 12 keyword fields      [REF gaussian_renderer/__init__.py:244-257]
  * ``GaussianRasterizer(raster_settings=...)`` and its call     [REF gaussian_renderer/__init__.py:259, 294-302]
    returning ``(depth (1,H,W), colour (3,H,W), radii (P,) int32, is_used (P,) int32)``
  * gradients reach ``means3D, means2D, shs | colors_precomp, opacities, scales, rotations | cov3D_precomp`` when
    ``loss.backward()`` runs                                     [REF mp_Mapper.py:242]
Tensors live on the current HIP device ("cuda" under ROCm); all kernels run on torch's current stream.
The three scratch buffers that survive between forward and backward are torch-owned byte tensors handed to the
library through resize callbacks, exactly like the reference's geom/binning/img buffers.
"""
import ctypes
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    # Extension (not in the reference): blend only tiles with tile_id % tile_mod == tile_rem (multi-GPU sharding).
    tile_mod: int = 1
    tile_rem: int = 0
    # Extension: duplicate-list capacity.  0 = size the lists by reading the count back (one host sync, the reference's
    # behaviour); > 0 = sync-free forward (HIP-graph capturable); GaussianRasterizer.num_rendered then holds the true
    # count on the device, and a count above the capacity renders nothing (see include/gsicp_hip.h).
    capacity: int = 0
    # Extension: depth compositing rule (the fork's rule cannot be read off the reference tree).  0 = sum z alpha T
    # (default), 1 = alpha-normalised (divided by the accumulated alpha 1 - T_final).  Both have exact backward passes.
    depth_mode: int = 0
    # Extension: int32[1] DEVICE tensor holding the number of live Gaussians when the input tensors are the full-capacity buffers of a
    # preallocated map (gs_icp_slam_amd/gaussian_store.py): rows behind it are ignored.  Needs capacity > 0 (the sync-free forward).
    live_count: object = None
    # Extension: opacities / scales / rotations are GaussianModel's RAW parameters (_opacity, _scaling, _rotation); sigmoid / exp / normalize
    # [REF scene/gaussian_model.py:44-56] and their chain rule run inside the preprocess kernels.  Needs capacity > 0.
    raw_params: bool = False
    # Extension (round 6): SPARSE gradients — the backward does not write the gradient rows of culled Gaussians (radii == 0, or behind live_count);
    # they are zero by definition and the consumer takes that from `radii` (FusedAdam.set_grad_row_mask; gs_icp_slam_amd/graph.py sets both).  Only for
    # callers that own every reader of the gradients: with the default (False) every row of every gradient is written, as the reference's backward does.
    sparse_grads: bool = False
    # Extension (round 6): the forward's per-call counter region has been cleared ahead of this call, in stream order (include/gsicp_hip.h "PRE-ZEROED forward"): no
    # zero-fill launch.  Set by gs_icp_slam_amd/graph.py on the CAPTURED forward only; needs capacity > 0.
    prezeroed: bool = False


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _f32c(t, device):
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_contiguous() or t.device != device:
        t = t.to(device=device, dtype=torch.float32).contiguous()
    return t


class _Scratch:
    """Resize callback target: allocates a torch byte tensor and keeps it alive."""

    def __init__(self, device):
        self.device = device
        self.tensor = None
        self.cb = _lib.RESIZE_FN(self._resize)

    def _resize(self, _user, nbytes):
        self.tensor = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=self.device)
        return self.tensor.data_ptr()


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                        count_out=None):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings, count_out)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    @_lib.traced("raster.forward")
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs, count_out=None):
        lib = _lib.load()
        if not means3D.is_cuda:
            raise RuntimeError("diff_gaussian_rasterization (gfx950): tensors must live on the HIP device; there is no CPU path")
        dev = means3D.device
        with torch.cuda.device(dev):
            means3D = _f32c(means3D, dev)
            P = means3D.shape[0]
            sh_c = _f32c(sh, dev) if sh is not None and sh.numel() > 0 else None
            col_c = _f32c(colors_precomp, dev) if colors_precomp is not None and colors_precomp.numel() > 0 else None
            op_c = _f32c(opacities, dev)
            sc_c = _f32c(scales, dev) if scales is not None and scales.numel() > 0 else None
            rot_c = _f32c(rotations, dev) if rotations is not None and rotations.numel() > 0 else None
            cov_c = _f32c(cov3Ds_precomp, dev) if cov3Ds_precomp is not None and cov3Ds_precomp.numel() > 0 else None
            M = 0 if sh_c is None else (sh_c.shape[1] if sh_c.dim() == 3 else sh_c.numel() // (3 * max(P, 1)))
            bg = _f32c(rs.bg, dev)
            view = _f32c(rs.viewmatrix, dev)
            proj = _f32c(rs.projmatrix, dev)
            campos = _f32c(rs.campos, dev)
            H, W = int(rs.image_height), int(rs.image_width)
            sharded = rs.tile_mod > 1
            alloc = torch.zeros if sharded else torch.empty
            color = alloc((3, H, W), dtype=torch.float32, device=dev)
            depth = alloc((1, H, W), dtype=torch.float32, device=dev)
            radii = torch.empty((P,), dtype=torch.int32, device=dev)
            is_used = torch.empty((P,), dtype=torch.int32, device=dev)
            geom, binning, img = _Scratch(dev), _Scratch(dev), _Scratch(dev)
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            args = (geom.cb, None, binning.cb, None, img.cb, None, P, int(rs.sh_degree), int(M), _ptr(bg), W, H, _ptr(means3D),
                    _ptr(sh_c), _ptr(col_c), _ptr(op_c), _ptr(sc_c), float(rs.scale_modifier), _ptr(rot_c), _ptr(cov_c), _ptr(view),
                    _ptr(proj), _ptr(campos), float(rs.tanfovx), float(rs.tanfovy), int(bool(rs.prefiltered)), _ptr(color),
                    _ptr(depth), _ptr(radii), _ptr(is_used), int(rs.tile_mod), int(rs.tile_rem),
                    int(bool(rs.debug)) | (2 if getattr(rs, "prezeroed", False) else 0), int(getattr(rs, "depth_mode", 0) or 0))
            capacity = int(getattr(rs, "capacity", 0) or 0)
            live = getattr(rs, "live_count", None)
            if live is not None and not (capacity > 0 and P > 0):
                raise RuntimeError("GaussianRasterizationSettings.live_count needs capacity > 0 (the sync-free forward)")
            if live is not None and (not live.is_cuda or live.dtype != torch.int32 or live.numel() != 1):
                raise RuntimeError("GaussianRasterizationSettings.live_count must be an int32[1] device tensor")
            raw = bool(getattr(rs, "raw_params", False))
            if raw and not (capacity > 0 and P > 0 and cov_c is None):
                raise RuntimeError("GaussianRasterizationSettings.raw_params needs capacity > 0 and (scales, rotations) inputs")
            if capacity > 0 and P > 0:
                n = lib.gsicp_raster_forward_async(*args, capacity, _ptr(count_out), _ptr(live), int(raw), stream)
            else:
                n = lib.gsicp_raster_forward(*args, stream)
                if count_out is not None:
                    count_out.fill_(max(int(n), 0))
            _lib.check(n, "gsicp_raster_forward")
        ctx.set_materialize_grads(False)   # no zero tensors for the non-differentiable int outputs (two 5 us fills per backward otherwise)
        ctx.rs = rs
        ctx.num_rendered = n
        ctx.M = int(M)
        ctx.have = (sh_c is not None, col_c is not None, sc_c is not None, rot_c is not None, cov_c is not None)
        ctx.depth_mode = int(getattr(rs, "depth_mode", 0) or 0)
        ctx.save_for_backward(means3D, sh_c, col_c, sc_c, rot_c, cov_c, radii, geom.tensor, binning.tensor, img.tensor, bg, view,
                              proj, campos, depth if ctx.depth_mode == 1 else None)
        ctx.mark_non_differentiable(radii, is_used)
        return depth, color, radii, is_used

    @staticmethod
    @_lib.traced("raster.backward")
    def backward(ctx, grad_depth, grad_color, _grad_radii, _grad_used):
        lib = _lib.load()
        (means3D, sh_c, col_c, sc_c, rot_c, cov_c, radii, geom, binning, img, bg, view, proj, campos, depth_out) = ctx.saved_tensors
        rs = ctx.rs
        dev = means3D.device
        P = means3D.shape[0]
        M = ctx.M
        H, W = int(rs.image_height), int(rs.image_width)
        with torch.cuda.device(dev):
            g_color = _f32c(grad_color, dev) if grad_color is not None else torch.zeros((3, H, W), device=dev)
            g_depth = _f32c(grad_depth, dev) if grad_depth is not None else None
            f32 = dict(dtype=torch.float32, device=dev)
            dL_dmeans2D = torch.empty((P, 3), **f32)
            dL_dconic = None                                       # intermediate results nobody reads: NULL skips their writes (56 B per Gaussian)
            dL_dopacity = torch.empty((P, 1), **f32)
            dL_dcolors = torch.empty((P, 3), **f32) if col_c is not None else None
            dL_ddepths = None
            dL_dmeans3D = torch.empty((P, 3), **f32)
            dL_dcov3D = torch.empty((P, 6), **f32) if cov_c is not None else None
            dL_dsh = torch.empty((P, M, 3), **f32) if sh_c is not None else None
            dL_dscales = torch.empty((P, 3), **f32) if sc_c is not None else None
            dL_drots = torch.empty((P, 4), **f32) if rot_c is not None else None
            scratch = torch.empty(int(lib.gsicp_raster_backward_scratch_bytes(int(ctx.num_rendered), W, H)), dtype=torch.uint8, device=dev)
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            rc = lib.gsicp_raster_backward(
                P, int(rs.sh_degree), M, int(ctx.num_rendered), _ptr(bg), W, H, _ptr(means3D), _ptr(sh_c), _ptr(col_c), _ptr(sc_c),
                float(rs.scale_modifier), _ptr(rot_c), _ptr(cov_c), _ptr(view), _ptr(proj), _ptr(campos), float(rs.tanfovx),
                float(rs.tanfovy), _ptr(radii), _ptr(geom), _ptr(binning), _ptr(img), _ptr(scratch), _ptr(g_color), _ptr(g_depth),
                _ptr(dL_dmeans2D), _ptr(dL_dconic), _ptr(dL_dopacity), _ptr(dL_dcolors), _ptr(dL_ddepths), _ptr(dL_dmeans3D),
                _ptr(dL_dcov3D), _ptr(dL_dsh), _ptr(dL_dscales), _ptr(dL_drots), int(rs.tile_mod), int(rs.tile_rem),
                int(bool(rs.debug)), ctx.depth_mode, _ptr(depth_out.detach() if depth_out is not None else None),
                _ptr(getattr(rs, "live_count", None)), int(bool(getattr(rs, "raw_params", False))) | (2 if getattr(rs, "sparse_grads", False) else 0), stream)
            _lib.check(rc, "gsicp_raster_backward")
        return (dL_dmeans3D, dL_dmeans2D, dL_dsh, dL_dcolors if col_c is not None else None, dL_dopacity, dL_dscales, dL_drots,
                dL_dcov3D if cov_c is not None else None, None, None)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings
        self.num_rendered = None   # int32[1] device tensor: (Gaussian, tile) duplicates of the last forward (extension)

    def markVisible(self, positions):
        lib = _lib.load()
        rs = self.raster_settings
        with torch.no_grad():
            pos = _f32c(positions, positions.device)
            P = pos.shape[0]
            present = torch.empty((P,), dtype=torch.uint8, device=pos.device)
            with torch.cuda.device(pos.device):
                stream = ctypes.c_void_p(torch.cuda.current_stream(pos.device).cuda_stream)
                _lib.check(lib.gsicp_raster_mark_visible(P, _ptr(pos), _ptr(_f32c(rs.viewmatrix, pos.device)),
                                                         _ptr(_f32c(rs.projmatrix, pos.device)), _ptr(present), stream),
                           "gsicp_raster_mark_visible")
        return present.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        if getattr(rs, "capacity", 0) and (self.num_rendered is None or self.num_rendered.device != means3D.device):
            self.num_rendered = torch.zeros(1, dtype=torch.int32, device=means3D.device)
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, rs,
                                   self.num_rendered if getattr(rs, "capacity", 0) else None)
