#!/usr/bin/env python
"""Golden vectors from the reference's OWN Python helpers (run in the build container, where /root/reference exists):

  * utils/general_utils.py  build_rotation / build_scaling_rotation / strip_symmetric  -> xyzw quaternion convention and the
    3-D covariance R S S^T R^T in the 6-vector order the rasteriser consumes                       [REF utils/general_utils.py:60-114]
  * utils/sh_utils.py       eval_sh (degrees 0..3) + the clamp_min(x + 0.5, 0) of render_3       [REF utils/sh_utils.py:57-112;
                                                                                                       gaussian_renderer/__init__.py:282-286]
  * utils/graphics_utils.py getWorld2View2 / getProjectionMatrix / focal2fov composed as SharedCam composes them
                                                                                                  [REF scene/shared_objs.py:157-166]

They pin the oracle (and synth.make_camera) to the reference where the reference's source is present.  The reference functions
allocate with device="cuda"; this script strips that keyword (no GPU here) — nothing else is altered.

    python tests/golden/make_golden_utils.py      -> tests/golden/ref_utils.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

_zeros = torch.zeros


def _zeros_cpu(*a, **k):
    k.pop("device", None)
    return _zeros(*a, **k)


torch.zeros = _zeros_cpu
import utils.general_utils as gu   # noqa: E402
import utils.graphics_utils as gr  # noqa: E402
import utils.sh_utils as su        # noqa: E402
from gs_icp_slam_amd import synth  # noqa: E402

out = {}
rng = np.random.default_rng(42)

# ---- quaternion convention + covariance assembly, on the smoke scene's Gaussians and on random un-normalised quaternions
g = synth.random_gaussians(300, seed=1)
for tag, q, s in (("scene", g["rotations"], g["scales"]),
                  ("rand", rng.normal(size=(64, 4)).astype(np.float32), np.abs(rng.normal(0.05, 0.03, (64, 3))).astype(np.float32) + 1e-3)):
    tq, ts = torch.from_numpy(q), torch.from_numpy(s)
    R = gu.build_rotation(tq)
    L = gu.build_scaling_rotation(1.0 * ts, tq)
    cov6 = gu.strip_symmetric(L @ L.transpose(1, 2))
    out[f"{tag}_q"], out[f"{tag}_s"] = q, s
    out[f"{tag}_R"], out[f"{tag}_cov6"] = R.numpy(), cov6.numpy()

# ---- SH evaluation, degrees 0..3: features (N, 16, 3) as GaussianModel stores them, view directions from a camera centre
N = 300
feats = rng.normal(0, 0.4, (N, 16, 3)).astype(np.float32)
campos = np.array([0.3, -0.2, -1.5], np.float32)
out["sh_feats"], out["sh_campos"] = feats, campos
for deg in range(4):
    shs_view = torch.from_numpy(feats).transpose(1, 2).reshape(-1, 3, 16)
    dir_pp = torch.from_numpy(g["means3D"]) - torch.from_numpy(campos)[None]
    dir_pp = dir_pp / dir_pp.norm(dim=1, keepdim=True)
    rgb = torch.clamp_min(su.eval_sh(deg, shs_view, dir_pp) + 0.5, 0.0)
    out[f"sh_rgb_deg{deg}"] = rgb.numpy()

# ---- camera matrices as SharedCam builds them
poses = [synth.DEFAULT_POSE_A, synth.se3((3.0, -20.0, 5.0), (0.4, 0.1, -0.7)), np.eye(4)]
for i, pose in enumerate(poses):
    for name, cfg in (("replica", synth.REPLICA), ("tum", synth.TUM)):
        W, H = cfg["W"], cfg["H"]
        fovx, fovy = gr.focal2fov(cfg["fx"], W), gr.focal2fov(cfg["fy"], H)
        w2c = np.linalg.inv(pose)
        Rr, Tt = w2c[:3, :3].transpose(), w2c[:3, 3]          # mp_Tracker.py:224-226: R = inv(pose)[:3,:3]^T, T = inv(pose)[:3,3]
        wvt = torch.tensor(gr.getWorld2View2(Rr, Tt)).transpose(0, 1)
        proj = gr.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
        full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
        center = wvt.inverse()[3, :3]
        k = f"cam{i}_{name}"
        out[k + "_pose"] = np.asarray(pose, np.float64)
        out[k + "_view"], out[k + "_full"], out[k + "_center"] = wvt.numpy(), full.numpy(), center.numpy()
        out[k + "_tanfov"] = np.array([np.tan(fovx * 0.5), np.tan(fovy * 0.5)])

np.savez_compressed(os.path.join(HERE, "ref_utils.npz"), **out)
print("wrote", os.path.join(HERE, "ref_utils.npz"), {k: v.shape for k, v in list(out.items())[:6]})
