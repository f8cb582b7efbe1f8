"""CPU: the GICP / kNN oracles against known answers, internal cross-checks and the committed golden vectors."""
import os

import numpy as np
import pytest

import oracle
from gs_icp_slam_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def pose_err(T, gt):
    dR = np.asarray(T, np.float64)[:3, :3] @ gt[:3, :3].T
    return np.degrees(np.linalg.norm(dR - np.eye(3)) / np.sqrt(2.0)), 1e3 * np.linalg.norm(np.asarray(T, np.float64)[:3, 3] - gt[:3, 3])


@pytest.mark.parametrize("name", ["gicp_replica", "gicp_tum"])
def test_oracle_reproduces_golden(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    reg = oracle.OracleGICP()
    reg.set_max_correspondence_distance(float(z["max_corr"]))
    reg.set_max_knn_distance(99999.0)
    reg.set_input_target(z["target"])
    reg.calculate_target_covariance_with_filter()
    np.testing.assert_allclose(reg.get_target_scales(), z["target_scales"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(reg.get_target_covariances(), z["target_cov"], rtol=1e-9, atol=1e-12)
    reg.set_input_source(z["source"])
    T = reg.align(z["init"])
    idx, d2 = reg.get_source_correspondence()
    np.testing.assert_allclose(T, z["T"], atol=1e-6)
    assert np.array_equal(idx, z["corr_idx"]) and np.array_equal(d2, z["corr_d2"])
    ang, mm = pose_err(T, z["gt"])
    assert ang < 0.5 and mm < 10.0     # sub-sampled, noisy (TUM) case: sanity only — the exact values are pinned above


def test_known_rigid_motion_is_recovered():
    rng = np.random.default_rng(0)
    tgt = np.concatenate([np.c_[rng.uniform(-1, 1, (1500, 2)), np.zeros(1500)], np.c_[rng.uniform(-1, 1, 1500), np.zeros(1500), rng.uniform(0, 1, 1500)],
                          np.c_[np.zeros(1500), rng.uniform(-1, 1, 1500), rng.uniform(0, 1, 1500)]])
    motion = synth.se3((0.6, -0.9, 0.4), (0.01, -0.015, 0.02))
    src = ((tgt - motion[:3, 3]) @ motion[:3, :3]).astype(np.float32)
    reg = oracle.OracleGICP()
    reg.set_max_correspondence_distance(0.2)
    reg.set_input_target(tgt)
    reg.set_input_source(src)
    T = reg.align(np.eye(4))
    ang, mm = pose_err(T, motion)
    assert ang < 0.01 and mm < 0.1, (ang, mm)


def test_kdtree_matches_brute_force_and_gate():
    rng = np.random.default_rng(1)
    tgt = rng.uniform(-1, 1, (3000, 3)).astype(np.float32)
    src = rng.uniform(-1, 1, (500, 3)).astype(np.float32)
    reg = oracle.OracleGICP()
    reg.set_max_correspondence_distance(0.1)
    reg.set_max_iterations(1)
    reg.set_input_target(tgt)
    reg.set_input_source(src)
    reg.align(np.eye(4))          # correspondences are those of the first linearisation (identity pose)
    idx, d2 = reg.get_source_correspondence()
    D = ((src[:, None, :] - tgt[None, :, :]) ** 2).astype(np.float32)
    bf = ((D[..., 0] + D[..., 1]) + D[..., 2])
    nn = bf.argmin(1)
    np.testing.assert_array_equal(d2, bf[np.arange(500), nn])
    gate = np.float32(0.1) * np.float32(0.1)
    assert np.array_equal(idx >= 0, d2 < gate) and np.array_equal(idx[idx >= 0], nn[idx >= 0])


def test_covariance_export_round_trip_and_filters():
    sp = synth.s_pair(synth.TUM)
    pts = sp["points_a"][::5]
    reg = oracle.OracleGICP()
    reg.set_regularization_method(0)                     # NONE: cost covariance = raw covariance
    reg.set_input_target(pts)
    reg.calculate_target_covariance_with_filter()
    raw = reg.get_target_covariances()
    q = reg.get_target_rotationsq().reshape(-1, 4).astype(np.float64)
    s = reg.get_target_scales().reshape(-1, 3).astype(np.float64)
    assert np.all(np.diff(s, axis=1) <= 1e-12) and np.allclose(np.linalg.norm(q, axis=1), 1, atol=1e-6)
    x, y, z, r = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z),
                  2 * (y * z - r * x), 2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    rec = R @ (s[:, :, None] ** 2 * np.swapaxes(R, 1, 2))
    full = raw[:, [0, 1, 2, 1, 3, 4, 2, 4, 5]].reshape(-1, 3, 3)
    np.testing.assert_allclose(rec, full, atol=5e-8)     # scales are sqrt(eigenvalues) of the raw covariance, frame = (x,y,z,w) quaternion
    assert np.all(np.linalg.det(R) > 0.999)
    # set_target_covariances_fromqs with NONE reproduces the same covariances
    reg.set_target_covariances_fromqs(q.astype(np.float32).ravel(), s.astype(np.float32).ravel())
    np.testing.assert_allclose(reg.get_target_covariances(), raw, atol=5e-8)
    # source filter: only trackable points produce correspondences, in rank order
    reg.set_input_source(pts[:50])
    f = np.zeros(50, np.int32)
    f[[4, 9, 30]] = [1, 2, 3]
    reg.set_source_filter(3, f)
    reg.set_max_correspondence_distance(0.05)
    reg.align(np.eye(4))
    idx, d2 = reg.get_source_correspondence()
    assert idx.tolist() == [4, 9, 30] and np.all(d2 == 0)


def test_knn_oracle_golden_and_definition():
    z = np.load(os.path.join(GOLD, "knn.npz"))
    got = oracle.knn_dist2(z["points"])
    np.testing.assert_allclose(got, z["dist2"], rtol=1e-6)
    p = z["points"].astype(np.float64)
    D = ((p[:, None] - p[None]) ** 2).sum(-1)
    np.fill_diagonal(D, np.inf)
    np.testing.assert_allclose(got, np.sort(D, 1)[:, :3].mean(1), rtol=1e-5)


def test_knn_oracle_is_invariant_to_point_order():
    """distCUDA2's mean squared distance to the 3 nearest neighbours is a property of the point SET: permuting the
    input permutes the output.  The HIP kernel is compared with this oracle in tests/test_knn_gpu.py."""
    import oracle
    rng = np.random.default_rng(3)
    pts = rng.normal(size=(500, 3)).astype(np.float32)
    perm = rng.permutation(len(pts))
    a, b = oracle.knn_dist2(pts), oracle.knn_dist2(pts[perm])
    np.testing.assert_allclose(b, a[perm], rtol=1e-6, atol=0)
