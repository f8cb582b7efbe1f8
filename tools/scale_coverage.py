#!/usr/bin/env python
"""SURVEY 8(a)'s one-keyframe render-coverage experiment: which meaning of `get_*_scales()` gives the mapper a usable first keyframe?

The tracker exports, per point, a quaternion and three "scales" of the kNN covariance [REF mp_Tracker.py:166-169]; the mapper turns them into
3D Gaussians with `log(scales / clamp_min(2 z^1.5, 1))`, opacity 0.1 [REF scene/gaussian_model.py:134-163] and renders them.  The fork's
native source is absent, so whether a scale is sqrt(eigenvalue) (a standard deviation — what a 3DGS scale IS) or the eigenvalue itself
(a variance) cannot be read off; both are implemented (`FastGICP.set_scale_semantics`).  This script builds the first keyframe of the
synthetic Replica-shaped sequence under each, renders it at the keyframe's own pose with the drop-in rasteriser, and measures how much
of the image the splats reach:

  * `touched`      — fraction of valid pixels that ANY Gaussian contributes to (accumulated alpha >= 1/255)
  * `alpha_mean`   — mean accumulated alpha (at the reference's initial opacity 0.1 this cannot exceed ~0.1 per layer of splats)
  * `solid@0.99`   — fraction of valid pixels with accumulated alpha > 0.5 when the same splats are rendered with opacity 0.99: the
                     footprint coverage, independent of the initial opacity
  * `sigma_px`     — median of the largest screen-space standard deviation of a splat, in pixels (the sampling stride is 10 px)

Prints one JSON object (and writes it to --json).  GPU only."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", choices=["replica", "tum"], default="replica")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    import torch
    import pygicp
    from gs_icp_slam_amd import synth
    from gs_icp_slam_amd.frontend import DepthFrontEnd
    from gs_icp_slam_amd.gaussian_store import rows_from_gicp
    from gs_icp_slam_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer

    cfg = synth.REPLICA if a.shape == "replica" else synth.TUM
    H, W = cfg["H"], cfg["W"]
    dev = torch.device("cuda", 0)
    pose = synth.trajectory(1)[0]
    rgb, d16 = synth.render_frame(cfg, pose)
    d16_dev, rgb_dev = torch.from_numpy(d16.view(np.int16)).to(dev), torch.from_numpy(rgb).to(dev)
    fe = DepthFrontEnd(H, W, cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], cfg["stride"], cfg["depth_scale"], cfg["depth_trunc"])
    pc = fe.make_pointcloud(d16_dev, rgb_dev)
    pw = DepthFrontEnd.to_world(pc.points, pose).contiguous()
    valid = torch.from_numpy(d16 > 0).to(dev)[None]
    cam = synth.make_camera(W, H, cfg["fx"], cfg["fy"], pose)
    rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(3, device=dev),
                                       scale_modifier=1.0, viewmatrix=torch.from_numpy(cam["viewmatrix"]).to(dev),
                                       projmatrix=torch.from_numpy(cam["projmatrix"]).to(dev), sh_degree=0,
                                       campos=torch.from_numpy(cam["campos"]).to(dev), prefiltered=False, debug=False)
    out = dict(shape=a.shape, points=int(pw.shape[0]), stride_px=cfg["stride"], data="synthetic analytic room, first frame of synth.trajectory")
    for sem in ("stddev", "variance"):
        reg = pygicp.FastGICP()
        reg.set_max_correspondence_distance(cfg["max_corr"])
        reg.set_max_knn_distance(99999.0)
        reg.set_scale_semantics(sem)
        reg.set_input_target(pw.cpu().numpy())
        trk = pc.trackable_idx.cpu().numpy()
        filt = np.zeros(pw.shape[0], np.int32)
        filt[trk] = np.arange(1, len(trk) + 1)
        reg.set_target_filter(len(trk), filt)
        reg.calculate_target_covariance_with_filter()
        rots = torch.from_numpy(np.reshape(np.asarray(reg.get_target_rotationsq(), np.float32), (-1, 4)).copy()).to(dev)
        scales = torch.from_numpy(np.reshape(np.asarray(reg.get_target_scales(), np.float32), (-1, 3)).copy()).to(dev)
        rows, _ = rows_from_gicp(pw, pc.colors, rots, scales, pc.z_values, pc.trackable_idx.long())   # create_from_pcd2_tensor's arithmetic
        act_scales = torch.exp(rows["scaling"])
        res = dict(exported_scale_median=[float(v) for v in scales.median(dim=0).values],
                   gaussian_scale_median_m=[float(v) for v in act_scales.median(dim=0).values])
        white = torch.ones((pw.shape[0], 3), device=dev)
        for tag, op in (("init_opacity_0.1", torch.sigmoid(rows["opacity"])), ("opacity_0.99", torch.full_like(rows["opacity"], 0.99))):
            with torch.no_grad():
                _d, acc, radii, _u = GaussianRasterizer(rs)(means3D=rows["xyz"], means2D=torch.zeros_like(rows["xyz"]), colors_precomp=white,
                                                           opacities=op, scales=act_scales, rotations=torch.nn.functional.normalize(rows["rotation"]))
            alpha = acc[0:1]     # white splats on a black background: the colour IS the accumulated alpha 1 - T_final
            v = valid
            res[tag] = dict(touched=float(((alpha >= 1.0 / 255.0) & v).sum() / v.sum()), alpha_gt_0p05=float(((alpha > 0.05) & v).sum() / v.sum()),
                            alpha_gt_0p5=float(((alpha > 0.5) & v).sum() / v.sum()), alpha_mean=float(alpha[v].mean()),
                            radius_px_median=float(radii[radii > 0].float().median()) if bool((radii > 0).any()) else 0.0,
                            visible=int((radii > 0).sum()))
        z = pc.z_values
        res["sigma_px_median"] = float((act_scales.max(dim=1).values * cfg["fx"] / z).median())
        out[sem] = res
    s, v = out["stddev"], out["variance"]
    out["verdict"] = ("std-dev semantics renders a usable first keyframe; variance semantics renders specks"
                      if s["opacity_0.99"]["alpha_gt_0p5"] > 4 * max(v["opacity_0.99"]["alpha_gt_0p5"], 1e-6) else "inconclusive")
    print(json.dumps(out, indent=1))
    if a.json:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        with open(a.json, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
