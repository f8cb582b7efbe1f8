"""Shared helpers for the parity tests (oracle = checker, never the thing under test)."""
import numpy as np


def torch_inputs(g, device="cuda", requires_grad=False):
    import torch
    out = {}
    for k, v in g.items():
        if isinstance(v, np.ndarray) and v.dtype == np.float32:
            t = torch.from_numpy(v).to(device)
            if requires_grad:
                t.requires_grad_(True)
            out[k] = t
    return out


def make_settings(cam, bg, sh_degree=0, device="cuda", tile_mod=1, tile_rem=0, depth_mode=0):
    import torch
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    return GaussianRasterizationSettings(
        image_height=cam["H"], image_width=cam["W"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
        bg=torch.tensor(bg, dtype=torch.float32, device=device), scale_modifier=1.0,
        viewmatrix=torch.from_numpy(cam["viewmatrix"]).to(device), projmatrix=torch.from_numpy(cam["projmatrix"]).to(device),
        sh_degree=sh_degree, campos=torch.from_numpy(cam["campos"]).to(device), prefiltered=False, debug=False,
        tile_mod=tile_mod, tile_rem=tile_rem, depth_mode=depth_mode)


def oracle_forward(g, cam, bg, sh_degree=0, dtype=np.float32, **kw):
    import oracle
    return oracle.raster_forward(g["means3D"], g["opacities"], cam["viewmatrix"], cam["projmatrix"], cam["campos"], cam["tanfovx"],
                                 cam["tanfovy"], cam["W"], cam["H"], bg, shs=g.get("shs"), scales=g.get("scales"),
                                 rotations=g.get("rotations"), colors_precomp=g.get("colors_precomp"),
                                 cov3D_precomp=g.get("cov3D_precomp"), sh_degree=sh_degree, dtype=dtype, **kw)


def oracle_backward(g, cam, bg, dL_dcolor, dL_ddepth, sh_degree=0, dtype=np.float32):
    import oracle
    return oracle.raster_backward(g["means3D"], g["opacities"], cam["viewmatrix"], cam["projmatrix"], cam["campos"], cam["tanfovx"],
                                  cam["tanfovy"], cam["W"], cam["H"], bg, dL_dcolor, dL_ddepth, shs=g.get("shs"),
                                  scales=g.get("scales"), rotations=g.get("rotations"), colors_precomp=g.get("colors_precomp"),
                                  cov3D_precomp=g.get("cov3D_precomp"), sh_degree=sh_degree, dtype=dtype)


def read_scratch(lib, ctx_tensors, P, num_rendered, W, H):
    """Decode the product's scratch buffers (sorted lists, ranges, per-pixel state) for bit-exact comparison."""
    import ctypes
    import torch
    geom, binning, img = ctx_tensors
    out = (ctypes.c_size_t * 12)()
    lib.gsicp_raster_layout(P, num_rendered, W, H, out)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    g8, b8, i8 = geom.cpu().numpy(), binning.cpu().numpy(), img.cpu().numpy()
    rec = g8[out[3]: out[3] + P * 48].view(np.float32).reshape(P, 12)
    pl_raw = b8[out[4]: out[4] + num_rendered * 4].view(np.uint32)
    entry_gauss = b8[out[10]: out[10] + num_rendered * 4].view(np.uint32)
    pl = entry_gauss[pl_raw & np.uint32(0x0FFFFFFF)] if num_rendered else pl_raw   # low 28 bits = emission slot -> Gaussian id
    tk = b8[out[5]: out[5] + num_rendered * 4].view(np.uint32)
    ranges = i8[out[6]: out[6] + T * 8].view(np.uint32).reshape(T, 2)
    fT = i8[out[7]: out[7] + W * H * 4].view(np.float32).reshape(H, W)
    nc = i8[out[8]: out[8] + W * H * 4].view(np.uint32).reshape(H, W)
    return dict(rec=rec, point_list=pl, strip_bits=pl_raw >> np.uint32(28), tile_keys=tk, ranges=ranges, final_T=fT, n_contrib=nc)
