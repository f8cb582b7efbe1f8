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


def _sym(c6):
    c = np.zeros((len(c6), 3, 3))
    c[:, 0, 0], c[:, 0, 1], c[:, 0, 2], c[:, 1, 1], c[:, 1, 2], c[:, 2, 2] = c6.T
    c[:, 1, 0], c[:, 2, 0], c[:, 2, 1] = c[:, 0, 1], c[:, 0, 2], c[:, 1, 2]
    return c


def test_converged_pose_is_a_stationary_point_of_the_published_gicp_objective():
    """Independent of the oracle's Jacobians, its LM damping and its linear solve: at the pose the oracle converges to (tiny epsilons), with the
    correspondences and Mahalanobis matrices of that pose held fixed as fast_gicp does within an iteration, the published objective
    sum_i d_i^T (C_B,i + R C_A,i R^T)^-1 d_i, d_i = b_i - T a_i, written out here in numpy, has a vanishing gradient (central differences over
    the six twist coordinates) and does not decrease under small random perturbations — on NOISY clouds with anisotropic covariances, where a
    wrong weighting or a wrong residual sign would still recover a noise-free known motion but not sit on this stationary point."""
    rng = np.random.default_rng(7)
    n = 2500
    tgt = np.concatenate([np.c_[rng.uniform(-1, 1, (n, 2)), np.zeros(n)], np.c_[rng.uniform(-1, 1, n), np.zeros(n), rng.uniform(0, 1, n)],
                          np.c_[np.zeros(n), rng.uniform(-1, 1, n), rng.uniform(0, 1, n)]]) + 0.004 * rng.standard_normal((3 * n, 3))
    motion = synth.se3((0.5, -0.7, 0.3), (0.008, -0.01, 0.012))
    pick = rng.permutation(3 * n)[:3000]
    src = ((tgt[pick] + 0.004 * rng.standard_normal((3000, 3)) - motion[:3, 3]) @ motion[:3, :3]).astype(np.float32)
    tgt = tgt.astype(np.float32)
    gate = 0.08
    reg = oracle.OracleGICP()
    reg.set_max_correspondence_distance(gate)
    reg.set_max_knn_distance(99999.0)
    reg.set_max_iterations(200)
    reg.set_rotation_epsilon(1e-10)
    reg.set_transformation_epsilon(1e-10)
    reg.set_input_target(tgt)
    reg.calculate_target_covariance_with_filter()
    reg.set_input_source(src)
    reg.calculate_source_covariance()
    T = np.asarray(reg.align(np.eye(4)), np.float64)
    idx, d2 = reg.get_source_correspondence()
    CA, CB = _sym(reg.get_source_covariances()), _sym(reg.get_target_covariances())
    used = np.flatnonzero((idx >= 0) & (d2 < gate * gate * 0.98))          # clear of the gate's edge (the pose returned is rounded to float32)
    assert len(used) > 2000
    a, b = src[used].astype(np.float64), tgt[idx[used]].astype(np.float64)
    R = T[:3, :3]
    M = np.linalg.inv(CB[idx[used]] + R @ CA[used] @ R.T)

    def cost(xi):
        w, v = xi[:3], xi[3:]
        th = np.linalg.norm(w)
        K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        dR = np.eye(3) if th == 0 else np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K
        d = b - ((a @ R.T + T[:3, 3]) @ dR.T + v)
        return float(np.einsum("ni,nij,nj->", d, M, d))

    def grad(at):
        g = np.zeros(6)
        for k in range(6):
            e = np.zeros(6); e[k] = 1e-6
            g[k] = (cost(at + e) - cost(at - e)) / 2e-6
        return g

    g0 = grad(np.zeros(6))
    g_off = grad(np.array([1e-3, -1e-3, 1e-3, 1e-3, 1e-3, -1e-3]))          # the scale of the gradient a millimetre / millirad away
    assert np.linalg.norm(g0) < 1e-5 * np.linalg.norm(g_off), (g0, g_off)     # measured 3.6e-7; weighting by C_B alone gives 0.15
    c0 = cost(np.zeros(6))
    for _ in range(20):
        assert cost(2e-4 * rng.standard_normal(6)) >= c0 * (1 - 1e-9)
    ang, mm = pose_err(T, motion)
    assert ang < 0.05 and mm < 1.0, (ang, mm)
