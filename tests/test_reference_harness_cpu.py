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
