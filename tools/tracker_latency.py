#!/usr/bin/env python
"""Wall-clock breakdown of one tracker frame (the four pygicp calls of mp_Tracker.py:191-231) with the GPU otherwise idle."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
import pygicp  # noqa: E402
from gs_icp_slam_amd import _lib, synth  # noqa: E402

cfg = synth.REPLICA if "--tum" not in sys.argv else synth.TUM
sp = (synth.s_pair_survey if "--survey" in sys.argv else synth.s_pair)(cfg, noise=("--tum" in sys.argv))   # --survey: SURVEY 8(d)'s pair (7 LM iterations)
pw = sp["points_a"].astype(np.float64) @ sp["pose_a"][:3, :3].T + sp["pose_a"][:3, 3]


def filt(n, tr):
    f = np.zeros(n, np.int32)
    f[tr] = np.arange(1, len(tr) + 1)
    return f


reg = pygicp.FastGICP()
reg.set_max_correspondence_distance(cfg["max_corr"])
reg.set_max_knn_distance(99999.0)
if "--map" in sys.argv:   # --map K: the steady-state configuration — a map-sized target of ~K Gaussians, source = frame 155 of the trajectory, guess = frame 154's pose
    K = int(sys.argv[sys.argv.index("--map") + 1])
    kreg = pygicp.FastGICP()
    kreg.set_max_knn_distance(99999.0)

    def cov_fn(p_):
        kreg.set_input_target(p_)
        kreg.calculate_target_covariance_with_filter()
        return kreg.get_target_rotationsq(), kreg.get_target_scales()
    m = synth.tracker_map(K, cov_fn)
    keep = m["trackable"] & (m["opacity"] > m["opacity_th"])
    poses = synth.trajectory(156)
    pb, _, tb, _ = synth.frame_points(cfg, poses[155])
    sp = dict(sp, points_b=pb, trackable_b=tb, pose_a=poses[154], pose_b=poses[155])
    reg.set_input_target(np.ascontiguousarray(m["points"][keep]))
    reg.set_target_covariances_fromqs(m["rotations"][keep].reshape(-1), m["scales"][keep].reshape(-1))
    print("map-sized target:", int(keep.sum()), "Gaussians")
else:
    reg.set_input_target(pw)
    reg.set_target_filter(len(sp["trackable_a"]), filt(len(pw), sp["trackable_a"]))
    reg.calculate_target_covariance_with_filter()
f_src = filt(len(sp["points_b"]), sp["trackable_b"])
DEVICE = "--device" in sys.argv   # device-resident frame: front-end kernel -> set_input_source(tensor) -> set_source_trackable
if DEVICE:
    from gs_icp_slam_amd.frontend import DepthFrontEnd
    stride = 10 if cfg is synth.REPLICA else 5
    fe = DepthFrontEnd(cfg["H"], cfg["W"], cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], stride, cfg["depth_scale"], 3.0)
    raw = np.clip(np.round(synth.raycast_depth(cfg, sp["pose_b"]) * cfg["depth_scale"]), 0, 65535).astype(np.uint16)
    depth_dev = torch.from_numpy(raw.view(np.int16)).cuda()
names = ["make_pointcloud / set_input_source", "set_source_filter / trackable", "align", "get_source_correspondence"]
acc = np.zeros(4)
N = 300
_lib.profile_enable(True)
for it in range(N + 20):
    if it == 20:
        acc[:] = 0
        _lib.profile_read()
        t_all = time.perf_counter()
    t0 = time.perf_counter()
    if DEVICE:
        pc = fe.make_pointcloud(depth_dev)
        reg.set_input_source(pc.points)
        t1 = time.perf_counter()
        reg.set_source_trackable(pc.trackable_idx)
    else:
        reg.set_input_source(sp["points_b"])
        t1 = time.perf_counter()
        reg.set_source_filter(len(sp["trackable_b"]), f_src)
    t2 = time.perf_counter()
    T = reg.align(sp["pose_a"])
    t3 = time.perf_counter()
    idx, d2 = reg.get_source_correspondence()
    t4 = time.perf_counter()
    acc += [t1 - t0, t2 - t1, t3 - t2, t4 - t3]
wall = (time.perf_counter() - t_all) / N
prof = _lib.profile_read()
print("frame wall %.1f us" % (wall * 1e6))
for n, a in zip(names, acc):
    print("  %-36s %.1f us" % (n, a / N * 1e6))
print("  kernels:", {k: round(1e3 * ms / N, 1) for k, (ms, c) in prof.items() if c > 0})
print("  align stats:", reg.last_align_stats())
print("  knn grid:", reg.knn_stats())

if os.environ.get("GSICP_ALIGN_TRACE"):
    import ctypes
    buf = (ctypes.c_ulonglong * 500)()
    n = _lib.load().gsicp_gicp_align_trace(reg._h, buf, 250)
    names_t = {0: "start", 1: "lin begin", 2: "lin compute done", 3: "solve done", 4: "trial cost done", 5: "spec lin done", 10: "wave+LDS reduce, partial stored",
               11: "barrier passed", 20: "  point loaded", 21: "  nn done", 22: "  maha/J done", 12: "partials summed", 99: "end",
               30: "accept/reject + convergence test + H fill done", 31: "damped H in registers", 32: "LDL^T solve done", 33: "se3 exp done"}
    t0 = buf[1]
    prev = t0
    for i in range(n):
        tag, t = buf[2 * i], buf[2 * i + 1]
        print("  %7.2f us (+%6.2f)  %s" % ((t - t0) / 100.0, (t - prev) / 100.0, names_t.get(tag, tag)))
        prev = t
