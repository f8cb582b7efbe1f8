"""GPU tests of the device-pointer overloads of the tracker: each must reproduce, bit for bit, what the
reference's host round trip (GPU -> CPU tensor -> numpy -> pygicp) produces — the numpy path itself is pinned against the oracle
in test_gicp_gpu.py."""
import numpy as np
import pytest
import torch

from gs_icp_slam_amd import synth
from tests.test_gicp_gpu import filt, world

pytestmark = pytest.mark.gpu


def _frame(reg, sp, cfg, device_inputs):
    import pygicp  # noqa: F401
    reg.set_max_correspondence_distance(cfg["max_corr"])
    reg.set_max_knn_distance(99999.0)
    pw = world(sp["points_a"], sp["pose_a"]).astype(np.float32)
    reg.set_input_target(pw)
    reg.set_target_filter(len(sp["trackable_a"]), filt(len(pw), sp["trackable_a"]))
    reg.calculate_target_covariance_with_filter()
    rots = np.reshape(reg.get_target_rotationsq(), (-1, 4)).copy()
    scales = np.reshape(reg.get_target_scales(), (-1, 3)).copy()
    if device_inputs:
        reg.set_input_target(torch.from_numpy(pw).cuda())
        reg.set_target_covariances_fromqs(torch.from_numpy(rots).cuda(), torch.from_numpy(scales).cuda())
        reg.set_input_source(torch.from_numpy(sp["points_b"]).cuda())
    else:
        reg.set_input_target(pw)
        reg.set_target_covariances_fromqs(rots.flatten(), scales.flatten())
        reg.set_input_source(sp["points_b"])
    reg.set_source_filter(len(sp["trackable_b"]), filt(len(sp["points_b"]), sp["trackable_b"]))
    T = reg.align(sp["pose_a"])
    idx, d2 = reg.get_source_correspondence()
    return T, idx, d2


def test_device_tensor_inputs_equal_numpy_inputs():
    import pygicp
    cfg = synth.REPLICA
    sp = synth.s_pair(cfg)
    Ta, ia, da = _frame(pygicp.FastGICP(), sp, cfg, device_inputs=False)
    reg = pygicp.FastGICP()
    Tb, ib, db = _frame(reg, sp, cfg, device_inputs=True)
    assert np.array_equal(Ta, Tb) and np.array_equal(ia, ib) and np.array_equal(da, db)
    # covariance export as device tensors == the numpy getters
    q = reg.get_source_rotationsq_tensor()
    s = reg.get_source_scales_tensor()
    torch.cuda.synchronize()
    assert np.array_equal(q.cpu().numpy().ravel(), np.asarray(reg.get_source_rotationsq()))
    assert np.array_equal(s.cpu().numpy().ravel(), np.asarray(reg.get_source_scales()))


def test_target_from_gaussians_equals_the_reference_host_round_trip():
    """get_trackable_gaussians_tensor -> .cpu() -> numpy -> set_input_target + set_target_covariances_fromqs
    [REF scene/gaussian_model.py:207-215; scene/shared_objs.py:81-126; mp_Tracker.py:284-289] vs one device call."""
    import pygicp
    cfg = synth.REPLICA
    sp = synth.s_pair(cfg)
    pw = world(sp["points_a"], sp["pose_a"]).astype(np.float32)
    n = len(pw)
    rng = np.random.default_rng(4)
    # a "map": the keyframe's points in random order, diluted with far-away and low-opacity Gaussians the selection must drop
    P = 3 * n
    perm = rng.permutation(P)
    xyz = np.zeros((P, 3), np.float32)
    xyz[perm[:n]] = pw
    xyz[perm[n:]] = pw[rng.integers(0, n, P - n)] + rng.normal(0, 0.5, (P - n, 3)).astype(np.float32)
    opacity = np.full((P, 1), 0.05, np.float32)
    opacity[perm[:n]] = rng.uniform(0.3, 0.99, (n, 1)).astype(np.float32)
    opacity[perm[n:2 * n]] = rng.uniform(0.3, 0.99, (n, 1)).astype(np.float32)
    mask = np.zeros(P, bool)
    mask[perm[:n]] = True
    mask[perm[2 * n:]] = True                      # trackable but transparent -> dropped by the opacity test
    quat = rng.normal(size=(P, 4)).astype(np.float32)
    quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    scl = np.abs(rng.normal(0.03, 0.01, (P, 3))).astype(np.float32) + 1e-3
    scl[:, 2] *= 0.05
    t = {k: torch.from_numpy(v).cuda() for k, v in dict(xyz=xyz, q=quat, s=scl, o=opacity).items()}
    tmask = torch.from_numpy(mask).cuda()
    th = 0.2

    def finish(reg):
        reg.set_max_correspondence_distance(cfg["max_corr"])
        reg.set_input_source(sp["points_b"])
        reg.set_source_filter(len(sp["trackable_b"]), filt(len(sp["points_b"]), sp["trackable_b"]))
        T = reg.align(sp["pose_a"])
        idx, d2 = reg.get_source_correspondence()
        return T, idx, d2

    # reference-style host path
    keep = torch.logical_and((t["o"] > th).squeeze(-1), tmask)
    tp, tr, ts = t["xyz"][keep].cpu().numpy(), t["q"][keep].cpu().numpy(), t["s"][keep].cpu().numpy()
    ra = pygicp.FastGICP()
    ra.set_max_correspondence_distance(cfg["max_corr"])
    ra.set_input_target(tp)
    ra.set_target_covariances_fromqs(tr.flatten(), ts.flatten())
    Ta, ia, da = finish(ra)
    # device path
    rb = pygicp.FastGICP()
    rb.set_max_correspondence_distance(cfg["max_corr"])
    count = rb.set_target_from_gaussians(t["xyz"], t["q"], t["s"], t["o"], tmask, th)
    assert count == int(keep.sum()) == n
    Tb, ib, db = finish(rb)
    assert np.array_equal(ia, ib) and np.array_equal(da, db) and np.array_equal(Ta, Tb)
    # no mask, nothing selected, size mismatch
    assert rb.set_target_from_gaussians(t["xyz"], t["q"], t["s"], t["o"], None, th) == int((t["o"] > th).sum())
    assert rb.set_target_from_gaussians(t["xyz"], t["q"], t["s"], t["o"], tmask, 2.0) == 0
    with pytest.raises(RuntimeError):
        rb.set_target_from_gaussians(t["xyz"], t["q"][:-1], t["s"], t["o"], tmask, th)
