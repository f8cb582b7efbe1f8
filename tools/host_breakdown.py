"""Diagnostics: where does a bench step spend HOST time?  (not part of the product or the tests)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gs_icp_slam_amd import synth
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
import pygicp

dev = torch.device("cuda", 0)
cfg = synth.REPLICA
W, H, P = cfg["W"], cfg["H"], 300_000
cam = synth.make_camera(W, H, cfg["fx"], cfg["fy"], synth.DEFAULT_POSE_A)
g = synth.s_map(P, seed=2)
params = {k: torch.from_numpy(g[k]).to(dev).requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities", "shs")}
rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(3, device=dev),
    scale_modifier=1.0, viewmatrix=torch.from_numpy(cam["viewmatrix"]).to(dev), projmatrix=torch.from_numpy(cam["projmatrix"]).to(dev),
    sh_degree=0, campos=torch.from_numpy(cam["campos"]).to(dev), prefiltered=False, debug=False)
rast = GaussianRasterizer(rs)
gt_c = torch.rand(3, H, W, device=dev); gt_d = torch.rand(1, H, W, device=dev)
sp = synth.s_pair(cfg)
pw = sp["points_a"].astype(np.float64) @ sp["pose_a"][:3, :3].T + sp["pose_a"][:3, 3]
reg = pygicp.FastGICP(); reg.set_max_correspondence_distance(0.02); reg.set_max_knn_distance(99999.0)
reg.set_input_target(pw); reg.calculate_target_covariance_with_filter()
f = np.zeros(len(sp["points_b"]), np.int32); f[sp["trackable_b"]] = np.arange(1, len(sp["trackable_b"]) + 1)
T = {}
def tic(): torch.cuda.synchronize(); return time.perf_counter()
def acc(k, t0, sync=True):
    if sync: torch.cuda.synchronize()
    T[k] = T.get(k, 0.0) + time.perf_counter() - t0
N = 40
for it in range(N + 5):
    if it == 5: T.clear()
    t0 = tic(); reg.set_input_source(sp["points_b"]); reg.set_source_filter(len(sp["trackable_b"]), f); acc("trk.set_source", t0)
    t0 = tic(); reg.align(sp["pose_a"]); acc("trk.align", t0)
    t0 = tic(); reg.get_source_correspondence(); acc("trk.get_corr", t0)
    m2 = torch.zeros_like(params["means3D"], requires_grad=True)
    t0 = tic(); d, c, r, u = rast(means3D=params["means3D"], means2D=m2, shs=params["shs"], opacities=params["opacities"], scales=params["scales"], rotations=params["rotations"]); acc("fwd.host", t0, sync=False); acc("fwd.total", t0)
    t0 = tic(); loss = (c - gt_c).abs().mean() + 0.1 * ((d - gt_d) / 10).abs().mean(); acc("loss.host", t0, sync=False); acc("loss.total", t0)
    t0 = tic(); loss.backward(); acc("bwd.host", t0, sync=False); acc("bwd.total", t0)
    for p in params.values(): p.grad = None
for k, v in T.items(): print(f"{k:16s} {1e6 * v / N:9.1f} us/step")
