"""Front-end: the oracle restatement on a hand-checkable case (CPU), and the HIP kernel against the
oracle bit for bit plus the device-resident tracker frame against the numpy-API frame (GPU)."""
import numpy as np
import pytest
import torch


def test_oracle_pick_table_and_pointcloud_small_case():
    from oracle import frontend_oracle as fo
    H, W, s = 7, 9, 3
    pick, x_pre, y_pre = fo.downsample_filter(H, W, 2.0, 4.0, 1.0, 2.0, s)
    rows = [0, 2, 5]                       # 0, s-1, 2s-1
    cols = [0, 3, 6]
    assert pick.tolist() == [r * W + c for r in rows for c in cols]
    assert np.allclose(x_pre.numpy(), [(c - 1.0) / 2.0 for _ in rows for c in cols])
    assert np.allclose(y_pre.numpy(), [(r - 2.0) / 4.0 for r in rows for _ in cols])
    depth = np.zeros((H, W), np.uint16)
    depth[0, 3], depth[2, 0], depth[5, 6] = 1000, 4000, 2000      # three valid picks, one beyond depth_trunc
    rgb = np.arange(H * W * 3, dtype=np.uint8).reshape(H, W, 3)
    pts, col, z, trk = fo.make_pointcloud(depth, rgb, pick, x_pre, y_pre, 1000.0, 3.0)
    assert z.tolist() == [1.0, 4.0, 2.0] and trk.tolist() == [0, 2]
    assert np.allclose(pts, [[(3 - 1) / 2 * 1, (0 - 2) / 4 * 1, 1], [(0 - 1) / 2 * 4, (2 - 2) / 4 * 4, 4], [(6 - 1) / 2 * 2, (5 - 2) / 4 * 2, 2]])
    assert np.allclose(col[0], rgb[0, 3] / 255.0)


def test_mirror_pick_table_equals_oracle_on_cpu():
    """DepthFrontEnd builds its table with its own (index-arithmetic) formulation; it must equal the restated reference
    construction (meshgrid + gather) exactly, for both benchmark resolutions and a stride that does not divide H."""
    from oracle import frontend_oracle as fo
    from gs_icp_slam_amd import synth
    from gs_icp_slam_amd.frontend import DepthFrontEnd
    for cfg, s in ((synth.REPLICA, 10), (synth.TUM, 5), (synth.TUM, 7)):
        H, W = cfg["H"], cfg["W"]
        pick, x_pre, y_pre = fo.downsample_filter(H, W, cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], s)
        fe = DepthFrontEnd(H, W, cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], s, cfg["depth_scale"], 3.0, device="cpu")
        assert torch.equal(fe.pick_idx_cpu, pick) and torch.equal(fe.x_pre_cpu, x_pre) and torch.equal(fe.y_pre_cpu, y_pre)
        assert fe.x_pre_cpu.dtype == torch.float32 and int(pick.max()) < H * W
    with pytest.raises(RuntimeError):
        fe.make_pointcloud(torch.zeros((cfg["H"], cfg["W"]), dtype=torch.float32))   # no CPU path


def _images(cfg, seed=0, holes=0.05):
    from gs_icp_slam_amd import synth
    depth_m = synth.raycast_depth(cfg, synth.DEFAULT_POSE_A).astype(np.float32)
    rng = np.random.default_rng(seed)
    depth_m[rng.uniform(size=depth_m.shape) < holes] = 0.0
    raw = np.clip(np.round(depth_m * cfg["depth_scale"]), 0, 65535).astype(np.uint16)
    rgb = rng.integers(0, 256, depth_m.shape + (3,), dtype=np.uint8)
    return raw, rgb


@pytest.mark.gpu
@pytest.mark.parametrize("name,as_float", [("replica", False), ("tum", False), ("replica", True)])
def test_frontend_kernel_matches_oracle_bit_for_bit(name, as_float):
    from oracle import frontend_oracle as fo
    from gs_icp_slam_amd import synth
    from gs_icp_slam_amd.frontend import DepthFrontEnd
    cfg = synth.REPLICA if name == "replica" else synth.TUM
    H, W = cfg["H"], cfg["W"]
    raw, rgb = _images(cfg)
    scale = 10 if name == "replica" else 5
    trunc = 3.0 if name == "replica" else cfg["depth_trunc"]
    pick, x_pre, y_pre = fo.downsample_filter(H, W, cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], scale)
    depth_in = raw.astype(np.float32) if as_float else raw
    pts, col, z, trk = fo.make_pointcloud(depth_in, rgb, pick, x_pre, y_pre, cfg["depth_scale"], trunc)
    fe = DepthFrontEnd(H, W, cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], scale, cfg["depth_scale"], trunc)
    assert torch.equal(fe.pick_idx_cpu, pick) and torch.equal(fe.x_pre_cpu, x_pre) and torch.equal(fe.y_pre_cpu, y_pre)
    d_dev = torch.from_numpy(depth_in.view(np.int16) if not as_float else depth_in).cuda()
    pc = fe.make_pointcloud(d_dev, torch.from_numpy(rgb).cuda())
    assert 0 < len(trk) < len(z) < len(pick)                      # holes and the truncation both bite
    assert np.array_equal(pc.points.cpu().numpy(), pts) and np.array_equal(pc.z_values.cpu().numpy(), z)
    assert np.array_equal(pc.colors.cpu().numpy(), col) and np.array_equal(pc.trackable_idx.cpu().numpy(), trk)
    pc2 = fe.make_pointcloud(d_dev)                               # without colours
    assert pc2.colors is None and torch.equal(pc2.points, pc.points)
    with pytest.raises(RuntimeError):
        fe.make_pointcloud(torch.from_numpy(depth_in.astype(np.float32)))   # host tensor


@pytest.mark.gpu
def test_device_resident_frame_equals_numpy_api_frame():
    """front-end -> set_input_source(device) -> set_source_trackable -> align == the reference's numpy call sequence."""
    import pygicp
    from oracle import frontend_oracle as fo
    from gs_icp_slam_amd import synth
    from gs_icp_slam_amd.frontend import DepthFrontEnd
    from tests.test_gicp_gpu import filt, world
    cfg = synth.REPLICA
    sp = synth.s_pair(cfg)
    raw = np.clip(np.round(synth.raycast_depth(cfg, sp["pose_b"]) * cfg["depth_scale"]), 0, 65535).astype(np.uint16)
    rgb = np.zeros(raw.shape + (3,), np.uint8)
    pick, x_pre, y_pre = fo.downsample_filter(cfg["H"], cfg["W"], cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], 10)
    pts, _, _, trk = fo.make_pointcloud(raw, rgb, pick, x_pre, y_pre, cfg["depth_scale"], 3.0)

    def target(reg):
        reg.set_max_correspondence_distance(cfg["max_corr"])
        pw = world(sp["points_a"], sp["pose_a"])
        reg.set_input_target(pw)
        reg.set_target_filter(len(sp["trackable_a"]), filt(len(pw), sp["trackable_a"]))
        reg.calculate_target_covariance_with_filter()

    ra = pygicp.FastGICP()
    target(ra)
    ra.set_input_source(pts)
    f = np.zeros(len(pts), np.int32)
    f[trk] = np.arange(1, len(trk) + 1)
    ra.set_source_filter(len(trk), f)
    Ta = ra.align(sp["pose_a"])
    ia, da = ra.get_source_correspondence()

    rb = pygicp.FastGICP()
    target(rb)
    fe = DepthFrontEnd(cfg["H"], cfg["W"], cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], 10, cfg["depth_scale"], 3.0)
    pc = fe.make_pointcloud(torch.from_numpy(raw.view(np.int16)).cuda())
    rb.set_input_source(pc.points)
    rb.set_source_trackable(pc.trackable_idx)
    Tb = rb.align(sp["pose_a"])
    ib, db = rb.get_source_correspondence()
    assert len(ia) == len(trk) and np.array_equal(Ta, Tb) and np.array_equal(ia, ib) and np.array_equal(da, db)
    w = DepthFrontEnd.to_world(pc.points, Tb).cpu().numpy()
    assert np.allclose(w, pts @ Tb[:3, :3].T + Tb[:3, 3], atol=1e-5)
