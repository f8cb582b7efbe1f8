"""Generates the golden fixtures in this directory FROM THE CPU ORACLE (oracle/), with fixed seeds.

The reference cannot produce golden vectors here: its native submodules are empty directories and it ships no
tests — PARITY IS UNPINNED against the reference.  These fixtures pin the oracle itself
(regression), and give the GPU tests a committed, machine-independent target.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle  # noqa: E402
from gs_icp_slam_amd import synth  # noqa: E402


def raster_case(seed, P, W, H, deg):
    cam = synth.make_camera(W, H, 0.8 * W, 0.8 * W)
    g = synth.random_gaussians(P, seed=seed, sh_degree=deg)
    bg = np.array([0.05, 0.1, 0.15], np.float32)
    rng = np.random.default_rng(seed + 100)
    gc = rng.normal(size=(3, H, W)).astype(np.float32)
    gd = rng.normal(size=(H, W)).astype(np.float32)
    kw = dict(shs=g["shs"], scales=g["scales"], rotations=g["rotations"], sh_degree=deg)
    args = (g["means3D"], g["opacities"], cam["viewmatrix"], cam["projmatrix"], cam["campos"], cam["tanfovx"], cam["tanfovy"], W, H, bg)
    f = oracle.raster_forward(*args, **kw)
    b = oracle.raster_backward(*args, gc, gd, **kw)
    out = dict(W=W, H=H, deg=deg, bg=bg, grad_color=gc, grad_depth=gd, viewmatrix=cam["viewmatrix"], projmatrix=cam["projmatrix"],
               campos=cam["campos"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], **{"in_" + k: v for k, v in g.items()})
    for k in ("color", "depth", "radii", "is_used", "point_list", "ranges", "n_contrib", "final_T", "margin"):
        out["fwd_" + k] = f[k]
    out["fwd_tile_keys"] = (f["keys"] >> np.uint64(32)).astype(np.uint32)
    for k, v in b.items():
        if v is not None:
            out["bwd_" + k] = v
    return out


def gicp_case(cfg, stride):
    sp = synth.s_pair(cfg, noise=(cfg is synth.TUM))
    pw = (sp["points_a"].astype(np.float64) @ sp["pose_a"][:3, :3].T + sp["pose_a"][:3, 3])[::stride]
    src = sp["points_b"][::stride]
    reg = oracle.OracleGICP()
    reg.set_max_correspondence_distance(cfg["max_corr"] * 2)
    reg.set_max_knn_distance(99999.0)
    reg.set_input_target(pw)
    reg.calculate_target_covariance_with_filter()
    rots, scales = reg.get_target_rotationsq().copy(), reg.get_target_scales().copy()
    cov_t = reg.get_target_covariances().copy()
    reg.set_input_source(src)
    T = reg.align(sp["pose_a"])
    idx, d2 = reg.get_source_correspondence()
    return dict(target=pw, source=src, init=sp["pose_a"], gt=sp["pose_b"], max_corr=cfg["max_corr"] * 2, T=T, corr_idx=idx, corr_d2=d2,
                target_rots=rots, target_scales=scales, target_cov=cov_t, iterations=reg.iterations)


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "raster_deg0.npz"), **raster_case(1, 160, 96, 64, 0))
    np.savez_compressed(os.path.join(HERE, "raster_deg3.npz"), **raster_case(2, 120, 80, 48, 3))
    np.savez_compressed(os.path.join(HERE, "gicp_replica.npz"), **gicp_case(synth.REPLICA, 6))
    np.savez_compressed(os.path.join(HERE, "gicp_tum.npz"), **gicp_case(synth.TUM, 8))
    pts = np.random.default_rng(7).normal(size=(400, 3)).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "knn.npz"), points=pts, dist2=oracle.knn_dist2(pts))
    print("golden fixtures written to", HERE)
