#!/usr/bin/env python
"""A miniature SLAM loop on synthetic frames that strings the whole device-resident path together the way the reference's two
processes use it [REF mp_Tracker.py:135-330; mp_Mapper.py:140-250]:

  frame 0   : front-end kernel -> world points -> tracker target (kNN covariances) -> first Gaussians into the GaussianStore
              -> mapper iterations as one hipGraph launch each
  frame k   : front-end kernel -> set_input_source / set_source_trackable (device tensors) -> align -> correspondences ->
              overlap statistics
  keyframes : not-yet-mapped points become new Gaussians (quaternions composed with the camera rotation on the device, store.append),
              the mapper graph is re-captured and run, and the map is handed back to the tracker on the device
              (set_target_from_gaussians)

It prints the pose error of every frame against the known trajectory and the mapper loss before / after each keyframe's iterations.
Not a benchmark (the ray-caster that makes the frames runs on the CPU); a functional end-to-end check."""
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import pygicp  # noqa: E402
from gs_icp_slam_amd import synth  # noqa: E402
from gs_icp_slam_amd.frontend import DepthFrontEnd, overlap_statistics, quaternion_multiply, rotation_to_quaternion_xyzw  # noqa: E402
from gs_icp_slam_amd.gaussian_store import GaussianStore  # noqa: E402
from gs_icp_slam_amd.graph import MapperIterationGraph  # noqa: E402
from gs_icp_slam_amd.optim import FusedAdam  # noqa: E402

cfg = synth.REPLICA
H, W = cfg["H"], cfg["W"]
FRAMES = int(sys.argv[1]) if len(sys.argv) > 1 else 7
KEY_EVERY = 2   # the synthetic pair leaves GICP's convergence basin (2 cm gate) after ~2 cm / 1 deg from its target: refresh the target that often
C0 = 0.28209479177387814
LRS = {"xyz": 1.6e-6 * 2.5, "f_dc": 2.5e-3, "f_rest": 1.25e-4, "opacity": 0.05, "scaling": 5e-3, "rotation": 1e-3}
dev = torch.device("cuda", 0)

motion = synth.se3((0.2, 0.3, 0.0), (0.008, 0.0, 0.003))
poses = [synth.DEFAULT_POSE_A.copy()]
for _ in range(FRAMES - 1):
    poses.append(poses[-1] @ motion)


def make_frame(pose):
    depth = synth.raycast_depth(cfg, pose)
    d16 = np.clip(np.round(depth * cfg["depth_scale"]), 0, 65535).astype(np.uint16)
    z = d16.astype(np.float64) / cfg["depth_scale"]
    u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    pc = np.stack([(u - cfg["cx"]) / cfg["fx"] * z, (v - cfg["cy"]) / cfg["fy"] * z, z], -1).reshape(-1, 3)
    pw = pc @ pose[:3, :3].T + pose[:3, 3]
    rgb = (synth.checker_colors(pw).reshape(H, W, 3) * 255.0).astype(np.uint8)
    return d16, rgb


def pose_err(T, gt):
    dR = np.asarray(T, np.float64)[:3, :3] @ gt[:3, :3].T
    return np.degrees(np.linalg.norm(dR - np.eye(3)) / math.sqrt(2.0)), 1e3 * np.linalg.norm(np.asarray(T, np.float64)[:3, 3] - gt[:3, 3])


def gaussians_from(points_w, colors, rots, scales):
    n = points_w.shape[0]
    return dict(xyz=points_w, f_dc=((colors - 0.5) / C0)[:, None, :].contiguous(), f_rest=torch.zeros((n, 0, 3), device=dev),
                opacity=torch.full((n, 1), math.log(0.1 / 0.9), device=dev), scaling=torch.log(scales.clamp_min(1e-3)), rotation=rots)


def mapper_run(store, opt, pose, d16, rgb, iters):
    cam = synth.make_camera(W, H, cfg["fx"], cfg["fy"], pose)
    params = {"means3D": store.params["xyz"], "shs": store.params["f_dc"], "opacities": store.params["opacity"],
              "scales": store.params["scaling"], "rotations": store.params["rotation"]}
    mg = MapperIterationGraph(params, opt, H, W, cam["tanfovx"], cam["tanfovy"], sh_degree=0, capacity=1 << 21, warmup=1)
    gt_img = (torch.from_numpy(rgb).to(dev).permute(2, 0, 1).float() / 255.0).contiguous()
    gt_dep = torch.from_numpy(d16.astype(np.float32) / np.float32(cfg["depth_scale"])).to(dev)[None].contiguous()
    mg.set_view(torch.from_numpy(cam["viewmatrix"]).to(dev), torch.from_numpy(cam["projmatrix"]).to(dev),
                torch.from_numpy(cam["campos"]).to(dev), gt_img, gt_dep)
    mg.capture()
    first = float(mg.step())
    for _ in range(iters - 1):
        last = float(mg.step())
    assert not mg.overflowed(), "duplicate-list capacity too small"
    return first, last


t_start = time.perf_counter()
fe = DepthFrontEnd(H, W, cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], cfg["stride"], cfg["depth_scale"], cfg["depth_trunc"])
reg = pygicp.FastGICP()
reg.set_max_correspondence_distance(cfg["max_corr"])
reg.set_max_knn_distance(99999.0)

# ---------------------------------------------------------------------------------------------- frame 0
d16, rgb = make_frame(poses[0])
pc = fe.make_pointcloud(torch.from_numpy(d16.view(np.int16)).to(dev), torch.from_numpy(rgb).to(dev))
pw = DepthFrontEnd.to_world(pc.points, poses[0]).contiguous()
reg.set_input_target(pw)                                              # device tensor in
trk = pc.trackable_idx.cpu().numpy()
filt = np.zeros(pw.shape[0], np.int32)
filt[trk] = np.arange(1, len(trk) + 1)
reg.set_target_filter(len(trk), filt)
reg.calculate_target_covariance_with_filter()
rots = torch.from_numpy(np.reshape(np.asarray(reg.get_target_rotationsq(), np.float32), (-1, 4)).copy()).to(dev)
scales = torch.from_numpy(np.reshape(np.asarray(reg.get_target_scales(), np.float32), (-1, 3)).copy()).to(dev)
store = GaussianStore(400_000, n_rest=0)
store.append(gaussians_from(pw, pc.colors, rots, scales))
opt = store.attach(FusedAdam, LRS, lr=0.0, eps=1e-15, capturable=True)
l0, l1 = mapper_run(store, opt, poses[0], d16, rgb, 20)
print(f"frame 0: {pw.shape[0]} points -> {store.n} Gaussians; mapper loss {l0:.5f} -> {l1:.5f} in 20 graph replays")
assert l1 < l0

# ---------------------------------------------------------------------------------------------- tracking
pose_est = poses[0].copy()
worst = (0.0, 0.0)
for k in range(1, FRAMES):
    d16, rgb = make_frame(poses[k])
    pc = fe.make_pointcloud(torch.from_numpy(d16.view(np.int16)).to(dev), torch.from_numpy(rgb).to(dev))
    reg.set_input_source(pc.points)                                   # device tensors in, no host round trip
    reg.set_source_trackable(pc.trackable_idx)
    T = reg.align(pose_est)
    idx, d2 = reg.get_source_correspondence()
    ang, mm = pose_err(T, poses[k])
    worst = (max(worst[0], ang), max(worst[1], mm))
    ratio, new_idx = overlap_statistics(torch.from_numpy(d2), 5e-4, 5e-5)
    line = f"frame {k}: {pc.points.shape[0]} points, {reg.iterations} LM iterations, pose error {ang:.4f} deg / {mm:.3f} mm, overlap {ratio:.2f}"
    pose_est = np.asarray(T, np.float64)
    if k % KEY_EVERY == 0:
        # new Gaussians from the points the map does not explain yet; their covariances come from the tracker, rotated into the world
        q_cam = rotation_to_quaternion_xyzw(pose_est[:3, :3]).float().to(dev)
        rots_w = quaternion_multiply(q_cam, reg.get_source_rotationsq_tensor())
        scales_s = reg.get_source_scales_tensor()
        sel = pc.trackable_idx.long()[new_idx.to(dev)]
        pw_k = DepthFrontEnd.to_world(pc.points, pose_est)
        before = store.n
        if sel.numel() > 0:
            store.append(gaussians_from(pw_k[sel].contiguous(), pc.colors[sel], rots_w[sel].contiguous(), scales_s[sel].contiguous()))
        l0, l1 = mapper_run(store, opt, pose_est, d16, rgb, 10)
        with torch.no_grad():                                          # the map goes back to the tracker without leaving the device
            n_t = reg.set_target_from_gaussians(store.params["xyz"], torch.nn.functional.normalize(store.params["rotation"]),
                                                torch.exp(store.params["scaling"]), torch.sigmoid(store.params["opacity"]),
                                                store.trackable_mask, 0.05)
        line += f" | keyframe: +{store.n - before} Gaussians ({store.n}), mapper loss {l0:.5f} -> {l1:.5f}, new target {n_t} points"
    print(line)

print(f"worst pose error {worst[0]:.4f} deg / {worst[1]:.3f} mm over {FRAMES - 1} tracked frames; wall {time.perf_counter() - t_start:.1f} s")
assert worst[0] < 0.05 and worst[1] < 1.5, "tracking drifted"
print("slam demo OK")
