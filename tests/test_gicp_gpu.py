"""GPU parity: HIP GICP tracker (through the drop-in pygicp API / C ABI) vs the CPU oracle, plus known-answer tests.

Bars: nearest-neighbour indices bit-exact (fp32 search with a fixed evaluation order on both sides); squared
distances bit-exact; covariances / scales / poses to fp64-reduction-order tolerance (1e-9 relative on covariances,
1e-6 on the float-rounded 4x4)."""
import numpy as np
import pytest

from gs_icp_slam_amd import synth

pytestmark = pytest.mark.gpu


def world(points, pose):
    return points.astype(np.float64) @ pose[:3, :3].T + pose[:3, 3]


def filt(n, trackable):
    f = np.zeros(n, np.int32)
    f[trackable] = np.arange(1, len(trackable) + 1)
    return f


def drive(reg, sp, cfg, use_fromqs=False):
    """The reference tracker's call sequence for frame 0 + one tracked frame (mp_Tracker.py:109-110,157-169,191-200,231)."""
    reg.set_max_correspondence_distance(cfg["max_corr"])
    reg.set_max_knn_distance(99999.0)
    pw = world(sp["points_a"], sp["pose_a"])
    reg.set_input_target(pw)
    reg.set_target_filter(len(sp["trackable_a"]), filt(len(pw), sp["trackable_a"]))
    reg.calculate_target_covariance_with_filter()
    rots = np.reshape(reg.get_target_rotationsq(), (-1, 4))
    scales = np.reshape(reg.get_target_scales(), (-1, 3))
    if use_fromqs:
        reg.set_input_target(pw.astype(np.float32))
        reg.set_target_covariances_fromqs(rots.flatten(), scales.flatten())
    reg.set_input_source(sp["points_b"])
    reg.set_source_filter(len(sp["trackable_b"]), filt(len(sp["points_b"]), sp["trackable_b"]))
    T = reg.align(sp["pose_a"])
    idx, d2 = reg.get_source_correspondence()
    srots = np.reshape(np.array(reg.get_source_rotationsq()), (-1, 4))
    sscales = np.reshape(np.array(reg.get_source_scales()), (-1, 3))
    return dict(T=T, idx=idx, d2=d2, rots=rots, scales=scales, srots=srots, sscales=sscales)


def quat_cov(q, s):
    x, y, z, r = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z),
                  2 * (y * z - r * x), 2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    return R @ (s[:, :, None] ** 2 * np.swapaxes(R, 1, 2))


def pose_err(T, gt):
    dR = T[:3, :3].astype(np.float64) @ gt[:3, :3].T
    ang = np.degrees(np.linalg.norm(dR - np.eye(3)) / np.sqrt(2.0))   # small-angle rotation error; acos(trace) loses precision near 0
    return ang, 1e3 * np.linalg.norm(T[:3, 3] - gt[:3, 3])


@pytest.mark.parametrize("name", ["replica", "tum"])
@pytest.mark.parametrize("fromqs", [False, True])
def test_align_matches_oracle(name, fromqs):
    import oracle
    import pygicp
    cfg = synth.REPLICA if name == "replica" else synth.TUM
    sp = synth.s_pair(cfg, noise=(name == "tum"))
    po = drive(oracle.OracleGICP(), sp, cfg, fromqs)
    reg = pygicp.FastGICP()
    pp = drive(reg, sp, cfg, fromqs)
    print(name, "fromqs", fromqs, reg.last_align_stats(), "pose err (deg, mm)", pose_err(pp["T"], sp["pose_b"]))
    # exported Gaussians: sign-invariant comparison through the reconstructed covariance
    np.testing.assert_allclose(pp["scales"], po["scales"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(quat_cov(pp["rots"], pp["scales"]), quat_cov(po["rots"], po["scales"]), rtol=0, atol=2e-7)
    np.testing.assert_allclose(pp["sscales"], po["sscales"], rtol=2e-5, atol=1e-7)
    assert np.allclose(np.linalg.norm(pp["rots"], axis=1), 1.0, atol=1e-5)
    # correspondences (as of the last linearisation): bit-exact indices and distances
    assert pp["idx"].shape == po["idx"].shape == (len(sp["trackable_b"]),)
    assert np.array_equal(pp["idx"], po["idx"]), f"{(pp['idx'] != po['idx']).sum()} correspondence indices differ"
    assert np.array_equal(pp["d2"], po["d2"]), f"max |d2 diff| {np.abs(pp['d2'] - po['d2']).max()}"
    np.testing.assert_allclose(pp["T"], po["T"], rtol=0, atol=1e-6)
    ang, mm = pose_err(pp["T"], sp["pose_b"])
    assert ang < 0.05 and mm < (1.0 if name == "replica" else 5.0)


@pytest.mark.parametrize("name", ["replica", "tum"])
def test_survey_spair_matches_oracle(name):
    """SURVEY 8(d)'s S-pair verbatim (room centre, looking +z, 1 deg about y + 2 cm along x).  Whatever basin the optimiser
    falls into, HIP and the oracle must fall into the SAME one: correspondences bit-exact, pose to 1e-6.  The distance to the
    ground-truth motion is reported (it is the basin limit of the algorithm on this geometry, not a property of the port)."""
    import oracle
    import pygicp
    cfg = synth.REPLICA if name == "replica" else synth.TUM
    sp = synth.s_pair_survey(cfg, noise=(name == "tum"))
    oreg, reg = oracle.OracleGICP(), pygicp.FastGICP()
    po, pp = drive(oreg, sp, cfg), drive(reg, sp, cfg)
    st = reg.last_align_stats()
    ang, mm = pose_err(pp["T"], sp["pose_b"])
    ango, mmo = pose_err(po["T"], sp["pose_b"])
    print(f"survey S-pair {name}: HIP {st}, oracle iterations {oreg.stats()}, error vs ground truth HIP {ang:.4f} deg / {mm:.2f} mm, "
          f"oracle {ango:.4f} deg / {mmo:.2f} mm")
    assert np.array_equal(pp["idx"], po["idx"]), f"{(pp['idx'] != po['idx']).sum()} correspondence indices differ"
    assert np.array_equal(pp["d2"], po["d2"])
    np.testing.assert_allclose(pp["T"], po["T"], rtol=0, atol=1e-6)
    assert st["iterations"] == oreg.iterations


def test_finite_max_knn_distance():
    """set_max_knn_distance with a radius that actually bites [REF mp_Tracker.py:110 passes 99999]: neighbours beyond it are dropped
    from the covariance estimate, on both sides alike — exported scales / covariances and the resulting pose agree."""
    import oracle
    import pygicp
    cfg = synth.REPLICA
    sp = synth.s_pair(cfg)
    outs = []
    for reg in (oracle.OracleGICP(), pygicp.FastGICP()):
        reg.set_max_correspondence_distance(cfg["max_corr"])
        reg.set_max_knn_distance(0.08)        # point spacing is 3-7 cm: most 20-neighbourhoods lose members, some keep < 3
        pw = world(sp["points_a"], sp["pose_a"])
        reg.set_input_target(pw)
        reg.set_target_filter(len(sp["trackable_a"]), filt(len(pw), sp["trackable_a"]))
        reg.calculate_target_covariance_with_filter()
        s = np.reshape(reg.get_target_scales(), (-1, 3))
        q = np.reshape(reg.get_target_rotationsq(), (-1, 4))
        reg.set_input_source(sp["points_b"])
        reg.set_source_filter(len(sp["trackable_b"]), filt(len(sp["points_b"]), sp["trackable_b"]))
        T = reg.align(sp["pose_a"])
        idx, d2 = reg.get_source_correspondence()
        outs.append(dict(s=s, q=q, T=T, idx=idx, d2=d2))
    o, p = outs
    # the radius must have changed something relative to the unlimited run
    ref = pygicp.FastGICP()
    ref.set_input_target(world(sp["points_a"], sp["pose_a"]))
    ref.calculate_target_covariance_with_filter()
    s_unl = np.reshape(ref.get_target_scales(), (-1, 3))
    assert (np.abs(s_unl - p["s"]).max(1) > 1e-4).mean() > 0.3
    np.testing.assert_allclose(p["s"], o["s"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(quat_cov(p["q"], p["s"]), quat_cov(o["q"], o["s"]), rtol=0, atol=2e-7)
    assert np.array_equal(p["idx"], o["idx"]) and np.array_equal(p["d2"], o["d2"])
    np.testing.assert_allclose(p["T"], o["T"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("method", ["NONE", "MIN_EIG", "NORMALIZED_MIN_EIG", "FROBENIUS"])
@pytest.mark.parametrize("fromqs", [False, True])
def test_non_plane_regularisation_matches_oracle(method, fromqs):
    """Every regularisation the option accepts besides the default PLANE, through both covariance sources (k-NN and
    set_target_covariances_fromqs)."""
    import oracle
    import pygicp
    cfg = synth.TUM
    sp = synth.s_pair(cfg, noise=True)
    code = {"NONE": 0, "MIN_EIG": 1, "NORMALIZED_MIN_EIG": 2, "PLANE": 3, "FROBENIUS": 4}[method]
    oreg, reg = oracle.OracleGICP(), pygicp.FastGICP()
    oreg.set_regularization_method(code)
    reg.set_regularization_method(method)
    po, pp = drive(oreg, sp, cfg, fromqs), drive(reg, sp, cfg, fromqs)
    print(method, "fromqs", fromqs, reg.last_align_stats(), "pose err", pose_err(pp["T"], sp["pose_b"]))
    assert np.array_equal(pp["idx"], po["idx"]), f"{(pp['idx'] != po['idx']).sum()} correspondence indices differ"
    assert np.array_equal(pp["d2"], po["d2"])
    np.testing.assert_allclose(pp["T"], po["T"], rtol=0, atol=2e-6)


def test_variance_scale_semantics_round_trips_and_matches_oracle():
    """set_scale_semantics("variance") (SURVEY 8a unknown, exposed as an option): exported scales are eigenvalues, fromqs consumes
    them as they are.  The covariance that round-trips through (rotationsq, scales) -> fromqs must be the one the std-dev semantics
    round-trips, so poses agree between the two semantics; and each semantics agrees with the oracle's instance of it."""
    import oracle
    import pygicp
    cfg = synth.TUM
    sp = synth.s_pair(cfg, noise=True)
    out = {}
    for mode in ("stddev", "variance"):
        oreg, reg = oracle.OracleGICP(), pygicp.FastGICP()
        oreg.set_scale_semantics(mode)
        reg.set_scale_semantics(mode)
        out[mode] = (drive(oreg, sp, cfg, True), drive(reg, sp, cfg, True))
        po, pp = out[mode]
        np.testing.assert_allclose(pp["scales"], po["scales"], rtol=4e-5, atol=1e-9)
        assert np.array_equal(pp["idx"], po["idx"]) and np.array_equal(pp["d2"], po["d2"])
        np.testing.assert_allclose(pp["T"], po["T"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(out["variance"][1]["scales"], out["stddev"][1]["scales"].astype(np.float64) ** 2, rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(out["variance"][1]["T"], out["stddev"][1]["T"], rtol=0, atol=2e-5)   # float32 export of s vs s^2 rounds differently


def test_lost_grid_barrier_recovers_on_the_device():
    """The persistent align kernel synchronises its workgroups with a hand-rolled grid barrier.  When a barrier gives up (a workgroup that
    never became resident under a saturating co-tenant, or the abort flag — injected here through the test hook), the registration is re-run
    as ONE workgroup (no grid barrier to lose) instead of returning a half-optimised pose; the next frame runs multi-workgroup again."""
    import pygicp
    cfg = synth.REPLICA
    sp = synth.s_pair(cfg)
    reg = pygicp.FastGICP()
    good = drive(reg, sp, cfg)
    assert reg.last_align_stats()["barrier_retries"] == 0
    reg._debug_abort_next_align()
    reg.set_input_source(sp["points_b"])
    reg.set_source_filter(len(sp["trackable_b"]), filt(len(sp["points_b"]), sp["trackable_b"]))
    T = reg.align(sp["pose_a"])
    st = reg.last_align_stats()
    assert st["barrier_retries"] == 1 and not st["failed"] and st["converged"]
    np.testing.assert_allclose(T, good["T"], rtol=0, atol=1e-6)
    idx, d2 = reg.get_source_correspondence()
    assert np.array_equal(idx, good["idx"]) and np.array_equal(d2, good["d2"])
    T2 = reg.align(sp["pose_a"])                       # barrier state was reset: the next launch is a normal one
    assert reg.last_align_stats()["barrier_retries"] == 1
    np.testing.assert_allclose(T2, good["T"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("k,max_iter", [(10, 64), (33, 64), (20, 3)])
def test_other_neighbourhood_sizes_and_iteration_caps_match_oracle(k, max_iter):
    """Upstream options the reference leaves at their defaults (set_correspondence_randomness = k of the k-NN covariances,
    set_max_iterations): other values must track the oracle just the same, including a run that is cut off before convergence."""
    import oracle
    import pygicp
    cfg = synth.TUM
    sp = synth.s_pair(cfg, noise=True)
    out = []
    for reg in (oracle.OracleGICP(), pygicp.FastGICP()):
        reg.set_correspondence_randomness(k)
        reg.set_max_iterations(max_iter)
        out.append((drive(reg, sp, cfg), reg))
    (po, oreg), (pp, reg) = out
    st = reg.last_align_stats()
    assert st["iterations"] == oreg.iterations
    if max_iter == 3:
        assert st["iterations"] == 3 and not st["converged"]
    np.testing.assert_allclose(pp["scales"], po["scales"], rtol=2e-5, atol=1e-7)
    assert np.array_equal(pp["idx"], po["idx"]) and np.array_equal(pp["d2"], po["d2"])
    np.testing.assert_allclose(pp["T"], po["T"], rtol=0, atol=1e-6)


def test_known_answer_rigid_motion():
    """Source = target moved by a known SE(3): GICP must recover it (no sampling difference, wide gate)."""
    import pygicp
    sp = synth.s_pair(synth.TUM)
    pw = world(sp["points_a"], sp["pose_a"])
    motion = synth.se3((0.4, -0.7, 0.3), (0.012, -0.008, 0.015))
    src = (pw - motion[:3, 3]) @ motion[:3, :3]          # motion^-1 applied
    reg = pygicp.FastGICP()
    reg.set_max_correspondence_distance(0.1)
    reg.set_input_target(pw)
    reg.set_input_source(src.astype(np.float32))
    T = reg.align(np.eye(4))
    ang, mm = pose_err(T, motion)
    assert ang < 0.01 and mm < 0.2, (ang, mm)


def test_default_gate_bruteforce_path_and_float64_input():
    import oracle
    import pygicp
    rng = np.random.default_rng(0)
    tgt = rng.uniform(-1, 1, size=(700, 3))
    motion = synth.se3((1.0, 2.0, -1.0), (0.01, 0.02, -0.01))
    src = ((tgt - motion[:3, 3]) @ motion[:3, :3])[:500]
    res = []
    for reg in (oracle.OracleGICP(), pygicp.FastGICP()):
        reg.set_input_target(tgt)            # float64 in, as at mp_Tracker.py:157
        reg.set_input_source(src.astype(np.float32))
        T = reg.align(np.eye(4))
        idx, d2 = reg.get_source_correspondence()
        res.append((T, idx, d2))
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    np.testing.assert_allclose(res[0][0], res[1][0], atol=1e-6)
    assert pose_err(res[1][0], motion)[1] < 0.5


def test_filters_and_errors():
    import pygicp
    reg = pygicp.FastGICP()
    with pytest.raises(RuntimeError):
        reg.align(np.eye(4))                           # nothing set
    pts = np.random.default_rng(1).uniform(-1, 1, (100, 3)).astype(np.float32)
    reg.set_input_target(pts)
    with pytest.raises(RuntimeError):
        reg.set_target_covariances_fromqs(np.zeros(8, np.float32), np.zeros(6, np.float32))   # wrong sizes
    with pytest.raises(RuntimeError):
        reg.set_input_source(np.zeros((5, 2), np.float32))
    reg.set_input_source(pts[:40])
    f = np.zeros(40, np.int32); f[[3, 7, 9]] = [1, 2, 3]
    reg.set_source_filter(3, f)
    reg.set_max_correspondence_distance(0.05)
    reg.align(np.eye(4))
    idx, d2 = reg.get_source_correspondence()
    assert idx.tolist() == [3, 7, 9] and np.all(d2 == 0.0)


def test_repeated_frames_on_one_object_equal_fresh_objects():
    """Round 3's two launch savings must not change a bit: (i) `set_input_*` + `set_*_filter` are one deferred, fused ingest launch — also when
    no filter call follows (the first consumer flushes the cloud), when the cloud is replaced before it was consumed, and when the filter
    arrives alone; (ii) after an object's first `get_source_correspondence`, `align` enqueues the correspondence kernels itself — the second
    frame of an object therefore takes the other path than its first, and both must equal what a fresh object returns, with the getter
    called once, twice or not at all in between, and across a gate change."""
    import pygicp
    sp1 = synth.s_pair(synth.REPLICA, motion=synth.se3((0.0, 1.0, 0.0), (0.02, 0.0, 0.0)))       # part of the source beyond the gate
    sp2 = synth.s_pair(synth.REPLICA, motion=synth.se3((0.3, -0.4, 0.2), (0.004, -0.003, 0.002)))

    def frame(reg, sp, gate, with_filter=True, ask=1):
        if getattr(reg, "_gate_set", None) != gate:                   # like the reference: the gate is set once, not per frame
            reg.set_max_correspondence_distance(gate)
            reg._gate_set = gate
        reg.set_input_source(sp["points_b"])
        if with_filter:
            reg.set_source_filter(len(sp["trackable_b"]), filt(len(sp["points_b"]), sp["trackable_b"]))
        T = reg.align(sp["pose_a"])
        got = [reg.get_source_correspondence() for _ in range(ask)]
        for g_ in got[1:]:
            assert np.array_equal(g_[0], got[0][0]) and np.array_equal(g_[1], got[0][1])
        return T, (got[0] if got else None)

    def fresh(sp, gate, with_filter=True):
        reg = pygicp.FastGICP()
        reg.set_max_knn_distance(99999.0)
        pw = world(sp1["points_a"], sp1["pose_a"])
        reg.set_input_target(np.zeros((7, 3)))                        # replaced before anything consumed it
        reg.set_input_target(pw)
        reg.set_target_filter(len(sp1["trackable_a"]), filt(len(pw), sp1["trackable_a"]))
        reg.calculate_target_covariance_with_filter()
        return reg, frame(reg, sp, gate, with_filter)

    def same(a, b):
        return np.array_equal(np.asarray(a[0]), np.asarray(b[0])) and np.array_equal(a[1][0], b[1][0]) and np.array_equal(a[1][1], b[1][1])

    g1, g2 = synth.REPLICA["max_corr"], 2.5 * synth.REPLICA["max_corr"]
    reg, first = fresh(sp1, g1)
    assert (first[1][0] < 0).mean() > 0.2                             # the exact-distance kernels have work to do
    assert same(frame(reg, sp2, g1, ask=2), fresh(sp2, g1)[1])        # second frame: the kernels ride behind the LM kernel
    frame(reg, sp1, g1, ask=0)                                        # a frame whose correspondences nobody asks for
    assert same(frame(reg, sp1, g1), first)
    assert same(frame(reg, sp2, g2), fresh(sp2, g2)[1])               # another gate: grid rebuilt, speculation follows
    sp3 = dict(sp2, points_b=np.ascontiguousarray(sp2["points_b"][sp2["trackable_b"]]))        # a small cloud, every point trackable
    unfiltered = fresh(sp3, g1, with_filter=False)[1]
    assert same(frame(reg, sp3, g1, with_filter=False), unfiltered)   # no filter call: the first consumer ingests the cloud
    reg.set_max_correspondence_distance(g2)                           # gate changed between align and the getter: recomputed on demand
    idx, d2 = reg.get_source_correspondence()
    assert len(idx) == len(sp3["points_b"]) and (idx >= 0).sum() >= (unfiltered[1][0] >= 0).sum()


def test_distances_beyond_gate_are_exact():
    """The reference exports the raw nearest-neighbour distance even when it exceeds the gate (thresholds at
    mp_Tracker.py:235 sit above max_corr^2)."""
    import oracle
    import pygicp
    sp = synth.s_pair(synth.REPLICA, motion=synth.se3((0.0, 1.0, 0.0), (0.02, 0.0, 0.0)))
    out = []
    for reg in (oracle.OracleGICP(), pygicp.FastGICP()):
        r = drive(reg, sp, synth.REPLICA)
        out.append(r)
    assert (out[0]["idx"] < 0).mean() > 0.2
    assert np.array_equal(out[0]["d2"], out[1]["d2"]) and np.array_equal(out[0]["idx"], out[1]["idx"])


def _clouds():
    rng = np.random.default_rng(11)
    plane = np.c_[rng.uniform(-2, 2, (4000, 2)), 0.002 * rng.standard_normal(4000)]
    volume = rng.uniform(-1, 1, (3000, 3))
    clusters = np.r_[rng.normal(0, 0.05, (600, 3)), rng.normal(0, 0.05, (15, 3)) + [40.0, 0, 0], rng.normal(0, 0.3, (300, 3)) + [0, 9.0, 0]]
    line = np.c_[np.linspace(0, 5, 700), np.zeros(700), np.zeros(700)] + 1e-4 * rng.standard_normal((700, 3))
    lattice = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(12)), -1).reshape(-1, 3) * 0.1   # many exactly tied distances
    dup = np.r_[volume[:500], volume[:500], volume[:40]]                                                      # exact duplicates
    tiny = rng.uniform(-1, 1, (10, 3))                                                                       # fewer points than k
    return dict(plane=plane, volume=volume, clusters=clusters, line=line, lattice=lattice, dup=dup, tiny=tiny)


@pytest.mark.parametrize("name", ["plane", "volume", "clusters", "line", "lattice", "dup", "tiny"])
def test_knn_covariances_match_oracle_on_awkward_clouds(name):
    """The grid k-NN (ring growth, whole-grid and scan-everything fallbacks, distance ties broken by index) must select the
    same neighbours as the oracle's exhaustive search: scales and covariances agree to fp64 summation-order tolerance."""
    import oracle
    import pygicp
    pts = _clouds()[name].astype(np.float32)
    out = []
    for reg in (pygicp.FastGICP(), oracle.OracleGICP()):
        reg.set_max_knn_distance(99999.0)
        reg.set_input_source(pts)
        reg.calculate_source_covariance()
        q = np.reshape(np.array(reg.get_source_rotationsq()), (-1, 4))
        s = np.reshape(np.array(reg.get_source_scales()), (-1, 3))
        out.append((q, s))
    (qg, sg), (qo, so) = out
    assert sg.shape == so.shape == (len(pts), 3)
    np.testing.assert_allclose(sg, so, rtol=2e-5, atol=1e-7)
    cg, co = quat_cov(qg.astype(np.float64), sg.astype(np.float64)), quat_cov(qo.astype(np.float64), so.astype(np.float64))
    np.testing.assert_allclose(cg, co, atol=2e-6 * max(1.0, float(np.abs(co).max())))


# ---------------------------------------------------------------------------------------------- map-sized targets (steady state)
def _oracle_knn_export(pw):
    import oracle
    r = oracle.OracleGICP()
    r.set_max_knn_distance(99999.0)
    r.set_input_target(pw)
    r.calculate_target_covariance_with_filter()
    return r.get_target_rotationsq(), r.get_target_scales()


_MAP_CACHE = {}


def _tracker_map(K):
    if K not in _MAP_CACHE:
        _MAP_CACHE.clear()                       # one map at a time (2 M rows at K = 1e6)
        _MAP_CACHE[K] = synth.tracker_map(K, _oracle_knn_export)
    return _MAP_CACHE[K]


def _source_frame(fid):
    cfg = synth.REPLICA
    poses = synth.trajectory(fid + 1)
    pts, _, trackable, _ = synth.frame_points(cfg, poses[fid])
    return poses, pts, trackable


@pytest.mark.parametrize("back", [1, 8])
@pytest.mark.parametrize("K", [100_000, 1_000_000])
def test_align_against_map_sized_target(K, back):
    """The tracker's STEADY-STATE configuration [REF mp_Tracker.py:282-288; scene/gaussian_model.py:207-215; gs_icp_slam.py:86 (capacity 10 M)]:
    after the first tracking keyframe the target is the MAP's trackable Gaussians — here ~K of them out of 2K rows (32 keyframes of the
    synthetic trajectory, random order, k-NN quaternions / shrunk scales, half the rows non-trackable or at / below the opacity threshold) — handed
    over as `set_input_target(points f32)` + `set_target_covariances_fromqs(rots.flatten(), scales.flatten())`.  The source is one 8 280-point
    frame between two keyframes; the initial guess is the pose `back` frames earlier (7 mm / 56 mm away).
    Three routes to the same target must agree with the oracle and with each other:
      (i)   the reference's route: rows selected on the host, numpy in;
      (ii)  all 2K rows + `set_target_filter` (the selection as a filter): correspondence indices then refer to the unselected array;
      (iii) `set_target_from_gaussians`: selection + covariances on the device.
    Bars: correspondence indices and squared distances bit-exact (also beyond the gate), pose <= 1e-6, same LM iteration count."""
    import oracle
    import pygicp
    import torch
    cfg = synth.REPLICA
    m = _tracker_map(K)
    keep = m["trackable"] & (m["opacity"] > m["opacity_th"])
    sel = np.where(keep)[0]
    assert abs(len(sel) - K) < 0.02 * K
    fid = 155
    poses, src, trackable = _source_frame(fid)
    f_src = filt(len(src), trackable)
    init = poses[fid - back]
    tp, tr, ts = m["points"][sel], m["rotations"][sel], m["scales"][sel]

    def frame(reg):
        reg.set_input_source(src)
        reg.set_source_filter(len(trackable), f_src)
        T = reg.align(init)
        idx, d2 = reg.get_source_correspondence()
        return T, idx, d2

    def host_route(reg):
        reg.set_max_correspondence_distance(cfg["max_corr"])
        reg.set_max_knn_distance(99999.0)
        reg.set_input_target(tp)
        reg.set_target_covariances_fromqs(tr.flatten(), ts.flatten())
        return frame(reg)

    oreg, reg = oracle.OracleGICP(), pygicp.FastGICP()
    To, io, do = host_route(oreg)
    Tp, ip, dp = host_route(reg)
    st = reg.last_align_stats()
    ang, mm = pose_err(Tp, poses[fid])
    print(f"map-sized target K={len(sel)} back={back}: HIP {st}, oracle {oreg.stats()}, pose error {ang:.5f} deg / {mm:.3f} mm, "
          f"in-gate {np.mean(do < cfg['max_corr'] ** 2):.3f}")
    assert ip.shape == io.shape == (len(trackable),)
    assert np.array_equal(ip, io), f"{(ip != io).sum()} correspondence indices differ"
    assert np.array_equal(dp, do), f"max |d2 diff| {np.abs(dp - do).max()}"
    np.testing.assert_allclose(Tp, To, rtol=0, atol=1e-6)
    assert st["iterations"] == oreg.iterations and st["barrier_retries"] == 0
    assert ang < 0.01 and mm < 0.5

    # (ii) the selection as a target filter over all rows
    f_tgt = filt(len(m["points"]), sel)
    res = []
    for r2 in (oracle.OracleGICP(), pygicp.FastGICP()):
        r2.set_max_correspondence_distance(cfg["max_corr"])
        r2.set_input_target(m["points"])
        r2.set_target_filter(len(sel), f_tgt)
        r2.set_target_covariances_fromqs(m["rotations"].flatten(), m["scales"].flatten())
        res.append(frame(r2))
    (T2o, i2o, d2o), (T2p, i2p, d2p) = res
    assert np.array_equal(i2p, i2o) and np.array_equal(d2p, d2o)
    np.testing.assert_allclose(T2p, T2o, rtol=0, atol=1e-6)
    hit = ip >= 0
    assert np.array_equal(i2p >= 0, hit) and np.array_equal(i2p[hit], sel[ip[hit]]) and np.array_equal(d2p, dp)   # same match, index into the full array
    np.testing.assert_array_equal(T2p, Tp)

    # (iii) the device hand-off
    dev = "cuda"
    r3 = pygicp.FastGICP()
    r3.set_max_correspondence_distance(cfg["max_corr"])
    n = r3.set_target_from_gaussians(torch.from_numpy(m["points"]).to(dev), torch.from_numpy(m["rotations"]).to(dev),
                                     torch.from_numpy(m["scales"]).to(dev), torch.from_numpy(m["opacity"]).to(dev),
                                     trackable_mask=torch.from_numpy(m["trackable"]).to(dev), opacity_th=m["opacity_th"])
    assert n == len(sel)
    T3, i3, d3 = frame(r3)
    assert np.array_equal(i3, ip) and np.array_equal(d3, dp)
    np.testing.assert_array_equal(T3, Tp)


def test_map_sized_target_with_points_beyond_the_gate():
    """A map-sized target that covers only PART of what the source sees (the first 6 keyframes of the trajectory, viewed from frame 200): a
    large share of the source has no neighbour inside the gate, so the miss list, the dense exact-NN grid over 1e5 targets and the export
    all work at scale; distances beyond the gate are the raw nearest-neighbour d2, as the reference's thresholds need [REF mp_Tracker.py:235]."""
    import oracle
    import pygicp
    cfg = synth.REPLICA
    cloud = synth.tracker_map_cloud(100_000, n_keyframes=6, seed=9)
    rots, scales = _oracle_knn_export(cloud["points"])
    fid = 200
    poses, src, trackable = _source_frame(fid)
    f_src = filt(len(src), trackable)
    out = []
    for reg in (oracle.OracleGICP(), pygicp.FastGICP()):
        reg.set_max_correspondence_distance(cfg["max_corr"])
        reg.set_input_target(cloud["points"])
        reg.set_target_covariances_fromqs(rots, scales)
        reg.set_input_source(src)
        reg.set_source_filter(len(trackable), f_src)
        T = reg.align(poses[fid - 1])
        out.append((T,) + tuple(reg.get_source_correspondence()))
    (To, io, do), (Tp, ip, dp) = out
    frac = float((io < 0).mean())
    print(f"partial map: {frac:.3f} of the source beyond the gate, max d2 {do.max():.4f}")
    assert 0.1 < frac < 0.95
    assert np.array_equal(ip, io) and np.array_equal(dp, do)
    np.testing.assert_allclose(Tp, To, rtol=0, atol=1e-6)


def test_knn_covariances_of_a_map_sized_cloud():
    """`calculate_target_covariance_with_filter` above the single-workgroup grid build (> 32 768 points: the six-launch chip-wide counting sort) on a
    1e5-point multi-keyframe cloud: exported scales / covariances equal the oracle's kd-tree k-NN to summation-order tolerance."""
    import pygicp
    cloud = synth.tracker_map_cloud(100_000, n_keyframes=6, seed=9)
    qo, so = _oracle_knn_export(cloud["points"])
    reg = pygicp.FastGICP()
    reg.set_max_knn_distance(99999.0)
    reg.set_input_target(cloud["points"])
    reg.calculate_target_covariance_with_filter()
    qg = np.reshape(reg.get_target_rotationsq(), (-1, 4)).astype(np.float64)
    sg = np.reshape(reg.get_target_scales(), (-1, 3)).astype(np.float64)
    qo, so = np.reshape(qo, (-1, 4)).astype(np.float64), np.reshape(so, (-1, 3)).astype(np.float64)
    print("knn stats", reg.knn_stats())
    np.testing.assert_allclose(sg, so, rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(quat_cov(qg, sg), quat_cov(qo, so), rtol=0, atol=2e-7)


def test_tum_configuration_at_map_scale():
    """the steady-state tracker under the TUM configuration [REF tum.sh:135-142: gate 0.03 m, stride 5 -> 12 416-point frames at
    640x480, depth truncated at 3 m, trackable_opacity_th 0.09] with SENSOR NOISE on both sides: the map = ~1e5 selected Gaussians of 32 noisy
    TUM-shaped keyframes (k-NN covariances of the noisy clouds, 2K rows, the opacity threshold of tum.sh), the source = a noisy frame with 15 %
    holes.  Three routes (host arrays / device hand-off) against the oracle: indices and squared distances bit-exact (also beyond the gate), pose
    <= 1e-6, same LM iteration count; a second case starts 8 frames back."""
    import oracle
    import pygicp
    import torch
    cfg = synth.TUM
    m = synth.tracker_map(170_000, _oracle_knn_export, cfg=cfg, opacity_th=0.09, noise=True, seed=6)      # ~1e5 pass the 3 m truncation + the selection
    keep = m["trackable"] & (m["opacity"] > m["opacity_th"])
    sel = np.where(keep)[0]
    tp, tr, ts = m["points"][sel], m["rotations"][sel], m["scales"][sel]
    fid = 155
    poses = synth.trajectory(fid + 1)
    src, _, trackable, _ = synth.frame_points(cfg, poses[fid], noise_seed=77, holes=0.15)
    assert 9000 < len(src) <= 12416 and 80_000 < len(sel) < 130_000
    f_src = filt(len(src), trackable)
    for back in (1, 8):
        init = poses[fid - back]

        def frame(reg):
            reg.set_input_source(src)
            reg.set_source_filter(len(trackable), f_src)
            T = reg.align(init)
            idx, d2 = reg.get_source_correspondence()
            return T, idx, d2
        out = []
        for reg in (oracle.OracleGICP(), pygicp.FastGICP()):
            reg.set_max_correspondence_distance(cfg["max_corr"])
            reg.set_max_knn_distance(99999.0)
            reg.set_input_target(tp)
            reg.set_target_covariances_fromqs(tr.flatten(), ts.flatten())
            out.append(frame(reg) + (reg,))
        (To, io, do, oreg), (Tp, ip, dp, reg) = out
        st = reg.last_align_stats()
        ang, mm = pose_err(Tp, poses[fid])
        print(f"TUM configuration, K={len(sel)} back={back}: HIP {st}, oracle iterations {oreg.iterations}, pose error {ang:.4f} deg / {mm:.2f} mm, "
              f"in-gate {np.mean(do < cfg['max_corr'] ** 2):.3f}, index {reg.target_index_stats()}")
        assert np.array_equal(ip, io), f"{(ip != io).sum()} correspondence indices differ"
        assert np.array_equal(dp, do), f"max |d2 diff| {np.abs(dp - do).max()}"
        np.testing.assert_allclose(Tp, To, rtol=0, atol=1e-6)
        assert st["iterations"] == oreg.iterations and st["barrier_retries"] == 0
        assert 0.9 < np.mean(do < cfg["max_corr"] ** 2) < 1.0 and len(trackable) < 0.8 * len(src)     # a few misses; a third of the frame beyond the 3 m truncation
        r3 = pygicp.FastGICP()
        r3.set_max_correspondence_distance(cfg["max_corr"])
        n = r3.set_target_from_gaussians(torch.from_numpy(m["points"]).cuda(), torch.from_numpy(m["rotations"]).cuda(), torch.from_numpy(m["scales"]).cuda(),
                                         torch.from_numpy(m["opacity"]).cuda(), trackable_mask=torch.from_numpy(m["trackable"]).cuda(), opacity_th=m["opacity_th"])
        assert n == len(sel)
        T3, i3, d3 = frame(r3)
        assert np.array_equal(i3, ip) and np.array_equal(d3, dp)
        np.testing.assert_array_equal(T3, Tp)


def test_second_index_level_is_exact_on_a_very_dense_target(monkeypatch):
    """the SECOND level of the target index (built when an occupied gate-sized cell holds >= 64 points — first reached by
    real maps at ~3e6 Gaussians, above every other test) against the kd-tree oracle.  Target: 120 000 points on three dense, mutually
    perpendicular wall patches (a 0.6 x 0.4 m patch + two strips: ~300 points per coarse cell), isotropic covariances through fromqs.  Source: 3 000
    points at four kinds of distance from the surface — inside the fine radius (answered by level 1), between the fine radius and the gate (level 1
    misses, the gate-sized level answers), just beyond the gate, and far away — so both the fine-level search and the fall-through run.  Bars: the
    index has two levels; correspondence indices and squared distances bit-equal to the oracle (also beyond the gate) and to the same object with
    GSICP_INDEX_LEVELS=1; the pose within 1e-6 of the oracle's; a second target on the SAME object (sort / temp buffers shared by the two builds
    of a two-level index, then reused) stays exact."""
    import oracle
    import pygicp
    rng = np.random.default_rng(42)
    gate = 0.02

    def cloud(n, shift):      # three mutually perpendicular dense patches (a corner): the registration is well conditioned
        na, nb = n * 3 // 5, n // 5
        a = np.stack([rng.uniform(0, 0.6, na), rng.uniform(0, 0.4, na), 2.0 + 2e-4 * rng.standard_normal(na)], 1)
        b = np.stack([0.6 + 2e-4 * rng.standard_normal(nb), rng.uniform(0, 0.4, nb), rng.uniform(1.9, 2.0, nb)], 1)
        c = np.stack([rng.uniform(0, 0.6, n - na - nb), 0.4 + 2e-4 * rng.standard_normal(n - na - nb), rng.uniform(1.9, 2.0, n - na - nb)], 1)
        return (np.concatenate([a, b, c]) + shift).astype(np.float32)

    def source(tgt, n):
        base = tgt[rng.choice(len(tgt), n, replace=False)].astype(np.float64)
        kind = rng.integers(0, 4, n)
        off = np.where(kind == 0, rng.uniform(0, 0.002, n), np.where(kind == 1, rng.uniform(0.004, 0.018, n),
                                                                      np.where(kind == 2, rng.uniform(0.0205, 0.03, n), rng.uniform(0.1, 0.5, n))))
        d = rng.standard_normal((n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        return (base + d * off[:, None]).astype(np.float32), kind

    def run(reg, tgt, src):
        reg.set_max_correspondence_distance(gate)
        reg.set_input_target(tgt)
        q = np.tile(np.array([0, 0, 0, 1], np.float32), (len(tgt), 1))
        reg.set_target_covariances_fromqs(q.flatten(), np.full((len(tgt), 3), 0.01, np.float32).flatten())
        reg.set_input_source(src)
        T = reg.align(np.eye(4))
        idx, d2 = reg.get_source_correspondence()
        return np.asarray(T), np.asarray(idx), np.asarray(d2)

    monkeypatch.delenv("GSICP_INDEX_LEVELS", raising=False)
    reg = pygicp.FastGICP()
    for trial, shift in enumerate((np.zeros(3), np.array([0.013, -0.007, 0.021]))):       # the second target reuses the object's buffers
        tgt = cloud(120_000, shift)
        src, kind = source(tgt, 3000)
        To, io, do = run(oracle.OracleGICP(), tgt, src)
        Tp, ip, dp = run(reg, tgt, src)
        st = reg.target_index_stats()
        print(f"two-level index, trial {trial}: {st}; in-gate {np.mean(do < gate * gate):.3f}, answered inside the fine radius "
              f"{np.mean(do < st['levels'][-1]['radius_m'] ** 2):.3f}")
        assert len(st["levels"]) == 2 and st["levels"][1]["radius_m"] < gate, st
        assert (do < st["levels"][1]["radius_m"] ** 2).mean() > 0.1 and ((do >= st["levels"][1]["radius_m"] ** 2) & (do < gate * gate)).mean() > 0.1
        assert np.array_equal(ip, io), f"{(ip != io).sum()} correspondence indices differ from the oracle"
        assert np.array_equal(dp, do), f"max |d2 diff| {np.abs(dp - do).max()}"
        np.testing.assert_allclose(Tp, To, rtol=0, atol=1e-6)
        monkeypatch.setenv("GSICP_INDEX_LEVELS", "1")
        one = pygicp.FastGICP()
        T1, i1, d1 = run(one, tgt, src)
        assert len(one.target_index_stats()["levels"]) == 1
        assert np.array_equal(i1, ip) and np.array_equal(d1, dp)
        np.testing.assert_allclose(T1, Tp, rtol=0, atol=1e-6)
        monkeypatch.delenv("GSICP_INDEX_LEVELS", raising=False)
