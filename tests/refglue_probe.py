"""Child process of tests/test_reference_slam_gpu.py::test_fused_gaussian_model_equals_the_reference_methods: drives `scene.gaussian_model.GaussianModel` of the
reference tree given on the command line (oracle/_ref/refpy = the untouched byte-code, oracle/_ref/refpy_fused = the same with gs_icp_slam_amd/refglue.py patched
in) through the calls the mapping process makes [REF mp_Mapper.py:131-136, 163-187, 244-245] on seeded inputs and saves what it holds after each."""
import sys
import types

import numpy as np
import torch

tree, out = sys.argv[1], sys.argv[2]
sys.path.insert(0, tree)
sys.argv = ["x"]
from scene.gaussian_model import GaussianModel  # noqa: E402

dev = "cuda"
rng = np.random.default_rng(3)


def batch(n, n_track):
    pts = torch.from_numpy(rng.uniform(-2, 2, (n, 3)).astype(np.float32)).to(dev)
    col = torch.from_numpy(rng.uniform(0, 1, (n, 3)).astype(np.float32)).to(dev)
    q = rng.normal(size=(n, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    sc = torch.from_numpy(np.exp(rng.normal(np.log(0.03), 0.7, (n, 3))).astype(np.float32)).to(dev)
    z = torch.from_numpy(rng.uniform(0.3, 6.0, n).astype(np.float32)).to(dev)
    tr = torch.from_numpy(np.sort(rng.choice(n, n_track, replace=False)).astype(np.int64)).to(dev)
    return pts, col, torch.from_numpy(q).to(dev), sc, z, tr


args = types.SimpleNamespace(percent_dense=0.01, position_lr_init=1.6e-6, position_lr_final=1.6e-8, position_lr_delay_mult=0.01, position_lr_max_steps=30000,
                             feature_lr=2.5e-3, opacity_lr=0.05, scaling_lr=5e-3, rotation_lr=1e-3)
gm = GaussianModel(0)
snaps = {}


def snap(tag):
    for name in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "trackable_mask"):
        snaps[f"{tag}.{name}"] = getattr(gm, name).detach().float().cpu().numpy()
    for name in ("get_xyz", "get_opacity", "get_scaling", "get_rotation", "get_features"):
        snaps[f"{tag}.{name}"] = getattr(gm, name).detach().cpu().numpy()
    snaps[f"{tag}.lrs"] = np.array([g["lr"] for g in gm.optimizer.param_groups], np.float64) if gm.optimizer is not None else np.zeros(0)
    snaps[f"{tag}.groups"] = np.array([g["name"] for g in gm.optimizer.param_groups]) if gm.optimizer is not None else np.zeros(0)


gm.create_from_pcd2_tensor(*batch(5000, 3000))
gm.spatial_lr_scale = 2.5
gm.training_setup(args)
gm.update_learning_rate(1)
gm.active_sh_degree = gm.max_sh_degree
snap("created")
gm.add_from_pcd2_tensor(*batch(3000, 1200))
snap("tracking_keyframe")
p = batch(2000, 1)
gm.add_from_pcd2_tensor(p[0], p[1], p[2], p[3], p[4], [])
snap("mapping_keyframe")
with torch.no_grad():     # what training would have done to some of them: transparent, and too large
    n = gm.get_xyz.shape[0]
    idx = torch.from_numpy(rng.choice(n, n // 5, replace=False)).to(dev)
    gm._opacity[idx] = -8.0
    big = torch.from_numpy(rng.choice(n, n // 20, replace=False)).to(dev)
    gm._scaling[big, 0] = 0.0     # exp(0) = 1 m > 0.1 * 2.5
gm.prune_large_and_transparent(0.005, 2.5)
snap("pruned")
t = gm.get_trackable_gaussians_tensor(0.05)
for i, name in enumerate(("points", "rots", "scales")):
    snaps[f"trackable.{name}"] = t[i].detach().cpu().numpy()
gm.add_from_pcd2_tensor(*batch(1000, 400))
snap("after_prune_keyframe")
np.savez(out, **snaps)
print("probe ok", tree, gm.get_xyz.shape[0])
