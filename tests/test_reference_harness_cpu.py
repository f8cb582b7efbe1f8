"""CPU checks of the harness that runs the reference's UNMODIFIED system on the drop-ins (tools/run_reference_slam.py): the byte-compiled
reference tree imports with the stand-in third-party modules, the synthetic sequence is written in Replica's on-disk layout and reads
back through the stand-ins exactly as the reference's loaders read it, the trajectory stays inside the analytic room, and the environment
shim restores the two NumPy-1 behaviours the reference's host files rely on.  (The run itself needs a GPU: tests/test_reference_slam_gpu.py.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUBS = os.path.join(ROOT, "tests", "refstubs")


def _reference_dir():
    sys.path.insert(0, ROOT)
    from tools.run_reference_slam import find_reference
    if os.path.isdir("/root/reference"):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "oracle", "make_refpy.py")], stdout=subprocess.DEVNULL)
    ref = os.path.join(ROOT, "oracle", "_ref", "refpy")
    if not os.path.exists(os.path.join(ref, "mp_Tracker.pyc")):
        ref = find_reference()
    if ref is None:
        pytest.skip("no reference tree on this machine")
    return ref


def test_bytecode_reference_imports_with_the_stand_ins():
    ref = _reference_dir()
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, STUBS]))
    code = ("import sys; sys.path.insert(0, %r); sys.argv = ['x']\n"
            "import gs_icp_slam_unlimit as m, mp_Tracker, mp_Mapper, scene.shared_objs as so\n"
            "assert m.Tracker.__module__ == 'mp_Tracker_unlimit' and m.Mapper is mp_Mapper.Mapper\n"
            "import pygicp, diff_gaussian_rasterization, simple_knn._C\n"
            "assert mp_Tracker.pygicp is pygicp\n"
            "print('imports ok', so.SharedGaussians.__name__)\n") % ref
    r = subprocess.run([sys.executable, "-W", "ignore", "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "imports ok SharedGaussians" in r.stdout, r.stderr[-3000:]


def test_synthetic_sequence_has_replicas_layout_and_reads_back_through_the_stand_ins(tmp_path):
    sys.path.insert(0, ROOT)
    from tools.make_synth_dataset import write_dataset
    from gs_icp_slam_amd import synth
    out = str(tmp_path / "seq")
    cfg, poses = write_dataset(out, frames=2, shape="tum")          # 640x480: quick to ray-cast
    assert sorted(os.listdir(os.path.join(out, "images"))) == ["frame000000.jpg", "frame000001.jpg"]
    assert sorted(os.listdir(os.path.join(out, "depth_images"))) == ["depth000000.png", "depth000001.png"]
    cam = open(os.path.join(out, "caminfo.txt")).readlines()[2].split()      # the line the reference parses [REF gs_icp_slam.py:52-63]
    assert (int(cam[0]), int(cam[1]), cam[8]) == (640, 480, "replica") and float(cam[6]) == 5000.0
    traj = np.loadtxt(os.path.join(out, "traj.txt")).reshape(-1, 4, 4)      # [REF utils/traj_utils.py:41-52]
    np.testing.assert_allclose(traj, np.stack(poses), rtol=0, atol=1e-12)
    sys.path.insert(0, STUBS)
    try:
        import cv2
        import open3d as o3d
        rgb = cv2.imread(os.path.join(out, "images", "frame000000.jpg"))              # [REF mp_Tracker.py:350]
        depth = np.array(o3d.io.read_image(os.path.join(out, "depth_images", "depth000000.png")))   # [REF mp_Tracker.py:351]
        assert cv2.imread(os.path.join(out, "nope.jpg")) is None
        raw = cv2.imread(os.path.join(out, "depth_images", "depth000000.png"), cv2.IMREAD_UNCHANGED)   # [REF mp_Mapper.py:362]
    finally:
        sys.path.remove(STUBS)
        for name in ("cv2", "open3d", "open3d.io"):
            sys.modules.pop(name, None)
    want_rgb, want_d16 = synth.render_frame(cfg, poses[0])
    assert depth.dtype == np.uint16 and np.array_equal(depth, want_d16) and np.array_equal(raw, want_d16)
    assert rgb.shape == (480, 640, 3) and rgb.dtype == np.uint8
    assert np.abs(rgb[..., ::-1].astype(np.int32) - want_rgb.astype(np.int32)).mean() < 3.0       # JPEG, B G R order
    back = np.ascontiguousarray(rgb[..., ::-1])
    assert np.abs(back.astype(np.int32) - want_rgb.astype(np.int32)).max() < 80


def test_trajectory_stays_inside_the_room_for_any_length():
    from gs_icp_slam_amd import synth
    poses = synth.trajectory(3000)
    pos = np.stack([p[:3, 3] for p in poses])
    assert np.all(pos > synth.ROOM_LO + 0.3) and np.all(pos < synth.ROOM_HI - 0.3)
    for lo, hi in synth.CUBOIDS:
        assert not np.any(np.all((pos > lo - 0.2) & (pos < hi + 0.2), axis=1)), "the camera passes through a cuboid"
    step = np.linalg.norm(np.diff(pos, axis=0), axis=1)
    assert step.max() < 0.008
    rot = [np.degrees(np.arccos(np.clip((np.trace(poses[k][:3, :3].T @ poses[k + 1][:3, :3]) - 1) / 2, -1, 1))) for k in range(0, 2999, 7)]
    assert max(rot) < 0.3


def test_numpy1_shim_restores_what_the_reference_relies_on():
    code = ("import numpy as np, torch\n"
            "assert np.unicode_ is np.str_\n"
            "r = np.linalg.inv(torch.eye(4) * 2)\n"
            "assert type(r) is np.ndarray and r.transpose().shape == (4, 4) and abs(r[0, 0] - 0.5) < 1e-7\n"
            "print('shim ok')\n")
    r = subprocess.run([sys.executable, "-W", "ignore", "-c", code], env=dict(os.environ, PYTHONPATH=STUBS), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "shim ok" in r.stdout, r.stderr[-2000:]


def test_tum_layout_is_read_by_the_references_own_tum_loader(tmp_path):
    """The reference's TUM branch — TrajManager('tum', ...) -> tum_load_poses -> parse_list (np.unicode_) -> associate_frames
    [REF utils/traj_utils.py:63-137] — executed UNMODIFIED (byte-code) on a synthetic sequence in TUM's on-disk layout: every frame is
    kept (30 Hz > 1/32 s), each frame is associated with its own depth image and its own ground-truth sample (not one of the in-between
    100 Hz samples), and the poses come back as written."""
    ref = _reference_dir()
    sys.path.insert(0, ROOT)
    from tools.make_synth_dataset import write_dataset, tum_stamp
    out = str(tmp_path / "tumseq")
    cfg, poses = write_dataset(out, frames=4, shape="tum", layout="tum")
    assert sorted(os.listdir(out)) == ["caminfo.txt", "depth", "depth.txt", "groundtruth.txt", "rgb", "rgb.txt"]
    assert open(os.path.join(out, "caminfo.txt")).readlines()[2].split()[8] == "tum"
    code = ("import sys, json; sys.path.insert(0, %r)\n"
            "from utils.traj_utils import TrajManager\n"
            "tm = TrajManager('tum', %r)\n"
            "print('RESULT ' + json.dumps(dict(poses=tm.gt_poses.tolist(), color=tm.color_paths, depth=tm.depth_paths)))\n") % (ref, out)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, STUBS]), MPLBACKEND="Agg")
    r = subprocess.run([sys.executable, "-W", "ignore", "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    got = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][0][7:])
    assert [os.path.basename(p) for p in got["color"]] == [tum_stamp(i) + ".png" for i in range(4)]
    assert [os.path.basename(p) for p in got["depth"]] == [tum_stamp(i, 0.011) + ".png" for i in range(4)]
    np.testing.assert_allclose(np.array(got["poses"]), np.stack(poses), rtol=0, atol=2e-8)      # %.9f text + quaternion round trip
    sys.path.insert(0, STUBS)
    try:
        import cv2
        import open3d as o3d
        from gs_icp_slam_amd import synth
        rgb = cv2.imread(got["color"][1])                                            # [REF mp_Tracker.py:355]
        depth = np.array(o3d.io.read_image(got["depth"][1]))                         # [REF mp_Tracker.py:356]
    finally:
        sys.path.remove(STUBS)
        for name in ("cv2", "open3d", "open3d.io"):
            sys.modules.pop(name, None)
    want_rgb, want_d16 = synth.render_frame(cfg, poses[1])
    assert np.array_equal(depth, want_d16) and np.array_equal(rgb[..., ::-1], want_rgb)     # PNG: lossless


def test_flag_sets_follow_the_dataset_not_the_harness_mode(tmp_path):
    """A REAL TUM path gets tum.sh's flags [REF tum.sh:135-142], a Replica path replica.sh's [REF replica.sh:135-142]."""
    sys.path.insert(0, ROOT)
    from tools import run_reference_slam as h
    tum_cfg, rep_cfg = tmp_path / "tum.txt", tmp_path / "rep.txt"
    tum_cfg.write_text("## camera parameters\nW H fx fy cx cy depth_scale depth_trunc dataset_type\n640 480 517.3 516.5 318.6 255.3 5000.0 3.0 tum\n")
    rep_cfg.write_text("## camera parameters\nW H fx fy cx cy depth_scale depth_trunc dataset_type\n1200 680 600.0 600.0 599.5 339.5 6553.5 12.0 replica\n")
    assert h.flags_for(str(tum_cfg)) == dict(keyframe_th=0.81, knn_maxd=99999.0, overlapped_th=1e-3, max_correspondence_distance=0.03,
                                             trackable_opacity_th=0.09, overlapped_th2=1e-3, downsample_rate=5)
    assert h.flags_for(str(rep_cfg)) == dict(keyframe_th=0.7, knn_maxd=99999.0, overlapped_th=5e-4, max_correspondence_distance=0.02,
                                             trackable_opacity_th=0.05, overlapped_th2=5e-5, downsample_rate=10)
    assert h.flags_for(str(rep_cfg), shape="tum") == h.TUM_FLAGS       # TUM-shaped sensor written in Replica's layout


def test_subset_of_a_cached_sequence_is_the_sequence_prefix(tmp_path):
    sys.path.insert(0, ROOT)
    from tools.make_synth_dataset import write_dataset, subset_dataset
    full, part = str(tmp_path / "full"), str(tmp_path / "part")
    write_dataset(full, frames=3, shape="tum")
    subset_dataset(full, 2, part)
    assert sorted(os.listdir(os.path.join(part, "images"))) == ["frame000000.jpg", "frame000001.jpg"]
    assert len(open(os.path.join(part, "traj.txt")).readlines()) == 2
    assert open(os.path.join(part, "images", "frame000001.jpg"), "rb").read() == open(os.path.join(full, "images", "frame000001.jpg"), "rb").read()


def test_fused_transform_edits_exactly_the_documented_statements():
    """oracle/make_refpy.py --fused (INTEGRATION.md 6-8's few-line edits as an AST transform, byte-code only): the edited modules import with the
    stand-ins, carry the calls into gs_icp_slam_amd/refglue.py where the reference's statements stood, and everything else of the mapping /
    tracking functions is still there.  (The run needs a GPU: tests/test_reference_slam_gpu.py::test_fused_rows_inside_...)"""
    _reference_dir()      # builds both trees where /root/reference exists
    fused = os.path.join(ROOT, "oracle", "_ref", "refpy_fused")
    if not os.path.exists(os.path.join(fused, "mp_Mapper.pyc")):
        pytest.skip("oracle/_ref/refpy_fused not built on this machine")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, STUBS]))
    code = ("import sys, types; sys.path.insert(0, %r); sys.argv = ['x']\n"
            "import mp_Mapper, mp_Tracker, mp_Tracker_unlimit, scene.gaussian_model as gm, scene.shared_objs as so\n"
            "from gs_icp_slam_amd import refglue\n"
            "def names(f):\n"
            "    out, todo = set(), [f.__code__]\n"
            "    while todo:\n"
            "        c = todo.pop(); out |= set(c.co_names); todo += [k for k in c.co_consts if isinstance(k, types.CodeType)]\n"
            "    return out\n"
            "m = names(mp_Mapper.Mapper.mapping)\n"
            "assert 'fused_mapping_iteration' in m and mp_Mapper.fused_mapping_iteration is refglue.fused_mapping_iteration\n"
            "assert not ({'render_3', 'l1_loss', 'ssim', 'backward', 'zero_grad'} & m), m      # the replaced statements are gone ...\n"
            "assert {'add_from_pcd2_tensor', 'get_trackable_gaussians_tensor', 'mapping_cams', 'new_keyframes', 'train_iter'} <= m   # ... the loop around them is not\n"
            "for T in (mp_Tracker.Tracker, mp_Tracker_unlimit.Tracker):\n"
            "    t = names(T.tracking)\n"
            "    assert 'get_values_tensor' in t and 'get_values_np' not in t and 'set_target_covariances_fromqs' in t and T._gsicp_fused\n"
            "assert gm.GaussianModel._gsicp_fused and gm.GaussianModel.create_from_pcd2_tensor.__module__ == 'gs_icp_slam_amd.refglue'\n"
            "assert gm.GaussianModel.save_ply.__module__ == 'scene.gaussian_model'            # untouched methods stay the reference's\n"
            "assert so.SharedTargetPoints._gsicp_fused and so.SharedGaussians.__init__.__module__ == 'scene.shared_objs'\n"
            "print('fused ok')\n") % fused
    r = subprocess.run([sys.executable, "-W", "ignore", "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fused ok" in r.stdout, r.stderr[-3000:]


def test_fused_policy_and_pacing_host_logic(monkeypatch):
    """gs_icp_slam_amd/refglue.py's policy switch and the `budget` pacing loop, without a GPU: the default is `free` (the reference's optimiser); `freeze` (opt-in) covers xyz / scaling / rotation with
    no pacing; `budget` holds the loop to k optimiser steps per tracked frame (the shared frame counter), lets an iteration through at once when the
    tracker raises a keyframe flag (it blocks on the mapper there [REF mp_Tracker.py:285-286]) and never waits longer than GSICP_FUSED_MAX_WAIT_MS."""
    import threading
    import time
    import types
    from gs_icp_slam_amd import refglue
    for k in ("GSICP_FUSED_POLICY", "GSICP_FUSED_ITERS_PER_FRAME", "GSICP_FUSED_FREEZE_GROUPS", "GSICP_FUSED_MIN_PERIOD_MS", "GSICP_FUSED_BURST", "GSICP_FUSED_MAX_WAIT_MS"):
        monkeypatch.delenv(k, raising=False)
    assert refglue.fused_policy() == dict(name="free", freeze_groups=(), iters_per_frame=0.0)       # default: nothing frozen, nothing paced
    monkeypatch.setenv("GSICP_FUSED_POLICY", "freeze")
    p = refglue.fused_policy()
    assert p["name"] == "freeze" and set(p["freeze_groups"]) == {"xyz", "scaling", "rotation"} and p["iters_per_frame"] == 0.0
    assert "DEVIATES" in refglue.POLICY_NOTES["freeze"] and "DEVIATES" not in refglue.POLICY_NOTES["free"]
    monkeypatch.setenv("GSICP_FUSED_POLICY", "nonsense")
    with pytest.raises(RuntimeError):
        refglue.fused_policy()
    monkeypatch.setenv("GSICP_FUSED_POLICY", "budget")
    assert refglue.fused_policy()["iters_per_frame"] == refglue.DEFAULT_ITERS_PER_FRAME and refglue.fused_policy()["freeze_groups"] == ()

    monkeypatch.setenv("GSICP_FUSED_BURST", "2")
    monkeypatch.setenv("GSICP_FUSED_MAX_WAIT_MS", "400")
    frames, eod, tkf, mkf = [0], [0], [0], [0]
    mapper = types.SimpleNamespace(iter_shared=frames, end_of_dataset=eod, is_tracking_keyframe_shared=tkf, is_mapping_keyframe_shared=mkf)
    gm = types.SimpleNamespace()

    def paced(steps_done):
        gm.__dict__["_gsicp_steps"] = steps_done
        t0 = time.perf_counter()
        refglue._pace(mapper, gm)
        return time.perf_counter() - t0
    assert paced(0) < 0.05 and paced(3) < 0.05            # burst 2 + 2 x (0 + 1) = 4 steps allowed while frame 0 is being tracked
    # over budget: waits until the tracker has consumed another frame
    threading.Timer(0.08, lambda: frames.__setitem__(0, 1)).start()
    dt = paced(4)
    assert 0.05 < dt < 0.35, dt
    # over budget, but the tracker raises a tracking keyframe: the iteration goes through at once
    threading.Timer(0.05, lambda: tkf.__setitem__(0, 1)).start()
    dt = paced(100)
    assert 0.02 < dt < 0.3, dt
    tkf[0] = 0
    # a stalled frame counter cannot hang the mapper
    monkeypatch.setenv("GSICP_FUSED_MAX_WAIT_MS", "60")
    dt = paced(100)
    assert 0.04 < dt < 0.3, dt
    assert gm.__dict__.get("_gsicp_paced", 0) >= 3
    # a mapper object without the shared flags (the iteration probe of the tests): no pacing at all
    assert refglue._pace(types.SimpleNamespace(iter_shared=[0]), gm) is None
