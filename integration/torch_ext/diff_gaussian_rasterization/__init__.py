"""`diff_gaussian_rasterization` as a Python package around the COMPILED `_C` extension module (integration/torch_ext_pybind.cpp).

This is the package shape the upstream submodule has — a settings tuple, an autograd function and an `nn.Module` over three `_C` entry
points — bound to libgsicp_hip.so instead of CUDA.  Put `integration/torch_ext/` in front of the repository root on `sys.path` to use it and `simple_knn._C` (and `integration/` for the
compiled `pygicp`) in place of the default ctypes mirrors of `gs_icp_slam_amd`; tools/run_reference_slam.py does that with
`--compiled-ext`.  Surface and conventions: SURVEY 8(b) — 12 settings fields constructed by keyword [REF gaussian_renderer/__init__.py:244-257],
`GaussianRasterizer(raster_settings=...)` called with `means3D, means2D, shs | colors_precomp, opacities, scales, rotations | cov3D_precomp`,
returning `(depth, colour, radii, is_used)` [REF gaussian_renderer/__init__.py:259, 294-302]; `means2D.grad` receives the screen-space gradient
[REF mp_Mapper.py:242, 252].  The extensions of the default mirror (sync-free capacity, tile sharding, raw parameters) are not part of this
surface: the fused / captured mapper iteration uses `gs_icp_slam_amd` directly.
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C


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


def _or_empty(t):
    return torch.Tensor([]) if t is None else t


def _or_none(t):
    return t if t.numel() > 0 else None


class _Rasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, rs):
        n, color, depth, radii, is_used, geom, binning, img = _C.rasterize_gaussians(
            rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3D_precomp, rs.viewmatrix, rs.projmatrix,
            rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh, rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)
        ctx.rs, ctx.num_rendered = rs, n
        ctx.save_for_backward(means3D, colors_precomp, scales, rotations, cov3D_precomp, sh, radii, geom, binning, img)
        ctx.mark_non_differentiable(radii, is_used)
        return depth, color, radii, is_used

    @staticmethod
    def backward(ctx, grad_depth, grad_color, _grad_radii, _grad_used):
        means3D, colors_precomp, scales, rotations, cov3D_precomp, sh, radii, geom, binning, img = ctx.saved_tensors
        rs = ctx.rs
        if grad_color is None:
            grad_color = torch.zeros((3, rs.image_height, rs.image_width), dtype=torch.float32, device=means3D.device)
        grad_depth = _or_empty(grad_depth)
        d_means2D, d_colors, d_opacity, d_means3D, d_cov3D, d_sh, d_scales, d_rots = _C.rasterize_gaussians_backward(
            rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3D_precomp, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
            rs.tanfovy, grad_color, grad_depth, sh, rs.sh_degree, rs.campos, geom, ctx.num_rendered, binning, img, rs.debug)
        return (d_means3D, d_means2D, _or_none(d_sh), _or_none(d_colors), d_opacity, _or_none(d_scales), _or_none(d_rots), _or_none(d_cov3D), None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
    return _Rasterize.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        return rasterize_gaussians(means3D, means2D, _or_empty(shs), _or_empty(colors_precomp), opacities, _or_empty(scales), _or_empty(rotations),
                                   _or_empty(cov3D_precomp), self.raster_settings)
