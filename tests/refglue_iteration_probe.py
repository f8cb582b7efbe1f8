"""Child process of tests/test_reference_slam_gpu.py::test_fused_iteration_tracks_the_references_own_statements: a few mapper iterations on one
synthetic keyframe, either through the reference's OWN training statements [REF mp_Mapper.py:219-262] (lifted unmodified into a callable at
build time by oracle/make_refpy.py, byte-code only: render_3 + torch loss chain + loss.backward() + torch.optim.Adam over the reference's
GaussianModel) or through gs_icp_slam_amd.refglue.fused_mapping_iteration over the patched GaussianModel (one hipGraph replay per iteration).
    python tests/refglue_iteration_probe.py <tree> <ref|fused> <out.npz> <iterations>"""
import math
import sys
import types

import numpy as np
import torch

tree, mode, out, iters = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
sys.path.insert(0, tree)
sys.argv = ["x"]
from scene.gaussian_model import GaussianModel          # noqa: E402
from scene.shared_objs import SharedCam                  # noqa: E402
import pygicp                                            # noqa: E402
from gs_icp_slam_amd import synth                        # noqa: E402

dev = "cuda"
cfg = synth.REPLICA
poses = synth.trajectory(12)
rgb, d16 = synth.render_frame(cfg, poses[0])
pts, z, trackable, depth_m = synth.frame_points(cfg, poses[0])
import os                                                # noqa: E402
if os.environ.get("GSICP_PROBE_TRACKABLE_EVERY"):        # a map that holds trackable AND non-trackable Gaussians (the freeze-policy test)
    trackable = trackable[::int(os.environ["GSICP_PROBE_TRACKABLE_EVERY"])]
pw = (pts.astype(np.float64) @ poses[0][:3, :3].T + poses[0][:3, 3]).astype(np.float32)
reg = pygicp.FastGICP()
reg.set_max_knn_distance(99999.0)
reg.set_input_target(pw)
reg.calculate_target_covariance_with_filter()
rots = np.reshape(reg.get_target_rotationsq(), (-1, 4)).copy()
scales = np.maximum(np.reshape(reg.get_target_scales(), (-1, 3)), 1e-4)
idx = synth.downsample_indices(cfg["W"], cfg["H"], cfg["stride"])
colors = (rgb.reshape(-1, 3)[idx].astype(np.float32) / 255.0)[d16.reshape(-1)[idx] != 0]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731

gm = GaussianModel(0)
gm.create_from_pcd2_tensor(t(pw), t(colors), t(rots), t(scales.astype(np.float32)), t(z), t(trackable.astype(np.int64)))
gm.spatial_lr_scale = 2.5
args = types.SimpleNamespace(percent_dense=0.01, position_lr_init=1.6e-6, position_lr_final=1.6e-6, position_lr_delay_mult=0.01, position_lr_max_steps=30000,
                             feature_lr=2.5e-3, opacity_lr=0.05, scaling_lr=5e-3, rotation_lr=1e-3)
gm.training_setup(args)
gm.update_learning_rate(1)
gm.active_sh_degree = gm.max_sh_degree

fov = lambda f, n: 2.0 * math.atan(n / (2.0 * f))   # noqa: E731
cams = []
for k in (0, 6, 11):                       # three keyframe views of the same first-keyframe map
    rgb_k, d16_k = synth.render_frame(cfg, poses[k])
    depth_k = d16_k.astype(np.float32) / np.float32(cfg["depth_scale"])
    cam = SharedCam(fov(cfg["fx"], cfg["W"]), fov(cfg["fy"], cfg["H"]), rgb_k, depth_k, cfg["cx"], cfg["cy"], cfg["fx"], cfg["fy"])
    w2c = np.linalg.inv(poses[k])
    cam.setup_cam(np.ascontiguousarray(poses[k][:3, :3]), np.ascontiguousarray(w2c[:3, 3]), rgb_k, depth_k)
    cam.on_cuda()
    cams.append(cam)

mapper = types.SimpleNamespace(gaussians=gm, pipe=types.SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False),
                               background=torch.zeros(3, device=dev), training_stage=0, lambda_dssim=0.2, train_iter=1, prune_th=2.5, rerun_viewer=False,
                               training=False, iter_shared=[0], total_start_time_viewer=0.0)
if mode == "ref":
    from _lifted_mapping_block import reference_training_block as step
else:
    from gs_icp_slam_amd.refglue import fused_mapping_iteration as fused

    def step(self, cam, gt_image, gt_depth_image):
        return fused(self, cam, gt_image, gt_depth_image), None, None

# train_iter starts at 1: at every train_iter % 200 == 0 the reference prunes AFTER loss.backward() and BEFORE optimizer.step(); prune_points re-creates
# the parameter tensors, the fresh ones carry no .grad, and that step applies nothing (1 iteration in 200).  The fused loop prunes first and steps.
losses = []
for i in range(iters):
    cam = cams[i % len(cams)]
    res = step(mapper, cam, cam.original_image.cuda(), cam.original_depth_image.cuda())
    losses.append(float(res[0].detach()))
    mapper.train_iter += 1
torch.cuda.synchronize()
np.savez(out, losses=np.array(losses), xyz=gm.get_xyz.detach().cpu().numpy(), f_dc=gm._features_dc.detach().cpu().numpy(),
         opacity=gm._opacity.detach().cpu().numpy(), scaling=gm._scaling.detach().cpu().numpy(), rotation=gm._rotation.detach().cpu().numpy(),
         trackable=gm.trackable_mask.detach().cpu().numpy().astype(np.uint8))
print("probe ok", mode, len(losses), gm.get_xyz.shape[0])
