#!/usr/bin/env python
"""Writes a synthetic RGB-D sequence in Replica's on-disk layout (what the reference's loaders read
[REF mp_Tracker.py:341-354; utils/traj_utils.py:41-52; gs_icp_slam.py:135-144]):

    <out>/images/frame000000.jpg ...     8-bit RGB
    <out>/depth_images/depth000000.png   16-bit depth, metres x depth_scale
    <out>/traj.txt                       one row-major 4x4 camera-to-world pose per line
    <out>/caminfo.txt                    the three-line camera config the reference's --config expects

or, with --layout tum, in the TUM RGB-D benchmark's layout — the branch the reference takes when the camera config says `tum`
[REF mp_Tracker.py:353-359; gs_icp_slam.py:143-149; utils/traj_utils.py:63-137]:

    <out>/rgb/<timestamp>.png            8-bit RGB
    <out>/depth/<timestamp>.png          16-bit depth, metres x 5000
    <out>/rgb.txt, depth.txt             "timestamp path" per line behind three comment lines, as the benchmark ships them
    <out>/groundtruth.txt                "timestamp tx ty tz qx qy qz qw" (camera-to-world), 100 Hz like the motion-capture track, so the
                                         reference's nearest-timestamp association [REF utils/traj_utils.py:121-137] has real work to do
    <out>/caminfo.txt                    third line ends in `tum`

The scene is the analytic room of gs_icp_slam_amd/synth.py (there is no dataset in this image and no network).
    python tools/make_synth_dataset.py OUT [--frames 30] [--shape replica|tum] [--layout replica|tum] [--noise]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gs_icp_slam_amd import synth  # noqa: E402


TUM_T0, TUM_DT = 1305031452.791720, 1.0 / 30.0     # fr1_desk starts at this stamp; 30 Hz (> 1/32 s apart: the reference keeps every frame)


def tum_stamp(i, offset=0.0):
    return "%.6f" % (TUM_T0 + i * TUM_DT + offset)


def _write_frame(job):
    from PIL import Image
    out, shape, i, pose, noise, quality, layout = job
    cfg = synth.REPLICA if shape == "replica" else synth.TUM
    rgb, d16 = synth.render_frame(cfg, pose, noise_seed=(100 + i) if noise else None, holes=0.15 if noise else 0.0)
    if layout == "tum":
        Image.fromarray(rgb, "RGB").save(os.path.join(out, "rgb", tum_stamp(i) + ".png"))
        Image.fromarray(d16).save(os.path.join(out, "depth", tum_stamp(i, 0.011) + ".png"))    # the depth camera stamps ~11 ms later, as in fr1
    else:
        Image.fromarray(rgb, "RGB").save(os.path.join(out, "images", f"frame{i:06d}.jpg"), quality=quality)
        Image.fromarray(d16).save(os.path.join(out, "depth_images", f"depth{i:06d}.png"))
    return i


def _write_tum_lists(out, poses):
    from scipy.spatial.transform import Rotation
    n = len(poses)
    with open(os.path.join(out, "rgb.txt"), "w") as fh:
        fh.write("# color images\n# file: 'synthetic analytic room'\n# timestamp filename\n")
        for i in range(n):
            fh.write(f"{tum_stamp(i)} rgb/{tum_stamp(i)}.png\n")
    with open(os.path.join(out, "depth.txt"), "w") as fh:
        fh.write("# depth maps\n# file: 'synthetic analytic room'\n# timestamp filename\n")
        for i in range(n):
            fh.write(f"{tum_stamp(i, 0.011)} depth/{tum_stamp(i, 0.011)}.png\n")
    # ground truth: the frames' own poses plus two in-between samples each (~100 Hz); the in-between samples are a little off the frame
    # poses, so a loader that associated a frame with the wrong stamp would show up in the ATE
    with open(os.path.join(out, "groundtruth.txt"), "w") as fh:
        fh.write("# ground truth trajectory\n# file: 'synthetic analytic room'\n# timestamp tx ty tz qx qy qz qw\n")
        for i in range(n):
            for sub in range(3):
                a = sub / 3.0
                nxt = poses[min(i + 1, n - 1)]
                t = (1 - a) * poses[i][:3, 3] + a * nxt[:3, 3]
                q = Rotation.from_matrix(poses[i][:3, :3]).as_quat()     # xyzw [REF utils/traj_utils.py:54-61 Rotation.from_quat]
                fh.write(tum_stamp(i, sub * TUM_DT / 3.0) + " " + " ".join("%.9f" % v for v in list(t) + list(q)) + "\n")


def write_dataset(out, frames=30, shape="replica", noise=False, quality=95, layout="replica", speed=1.0, jitter=0.0):
    cfg = synth.REPLICA if shape == "replica" else synth.TUM
    for sub in (("rgb", "depth") if layout == "tum" else ("images", "depth_images")):
        os.makedirs(os.path.join(out, sub), exist_ok=True)
    poses = synth.trajectory(frames, speed=speed, jitter=jitter)
    jobs = [(out, shape, i, pose, noise, quality, layout) for i, pose in enumerate(poses)]
    workers = max(1, min(16, (os.cpu_count() or 2) // 2, frames))
    if workers > 1:   # the CPU ray-caster costs ~1 s per 1200x680 frame
        from concurrent.futures import ProcessPoolExecutor
        with ProcessPoolExecutor(max_workers=workers) as ex:
            list(ex.map(_write_frame, jobs))
    else:
        for j in jobs:
            _write_frame(j)
    if layout == "tum":
        _write_tum_lists(out, poses)
    else:
        with open(os.path.join(out, "traj.txt"), "w") as fh:
            for pose in poses:
                fh.write(" ".join(repr(float(v)) for v in pose.reshape(-1)) + "\n")
    with open(os.path.join(out, "caminfo.txt"), "w") as fh:   # third line is the one parsed [REF gs_icp_slam.py:52-63]
        fh.write("## camera parameters (synthetic room, %s-shaped)\nW H fx fy cx cy depth_scale depth_trunc dataset_type\n" % shape)
        fh.write(f"{cfg['W']} {cfg['H']} {cfg['fx']} {cfg['fy']} {cfg['cx']} {cfg['cy']} {cfg['depth_scale']} {cfg['depth_trunc']} {layout}\n")
    return cfg, poses


def subset_dataset(src, n, dst):
    """The first n frames of a Replica-layout dataset as a directory of symlinks (one ray-cast sequence serves runs of several lengths)."""
    for sub in ("images", "depth_images"):
        os.makedirs(os.path.join(dst, sub), exist_ok=True)
        for name in sorted(os.listdir(os.path.join(src, sub)))[:n]:
            link = os.path.join(dst, sub, name)
            if not os.path.lexists(link):
                os.symlink(os.path.join(os.path.abspath(src), sub, name), link)
    with open(os.path.join(src, "traj.txt")) as fh:
        lines = fh.readlines()[:n]
    with open(os.path.join(dst, "traj.txt"), "w") as fh:
        fh.writelines(lines)
    with open(os.path.join(src, "caminfo.txt")) as fi, open(os.path.join(dst, "caminfo.txt"), "w") as fo:
        fo.write(fi.read())
    return dst


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--shape", choices=["replica", "tum"], default="replica")
    ap.add_argument("--layout", choices=["replica", "tum"], default=None, help="on-disk layout (default: the shape's own)")
    ap.add_argument("--noise", action="store_true")
    a = ap.parse_args()
    write_dataset(a.out, a.frames, a.shape, a.noise, layout=a.layout or a.shape)
    print(f"wrote {a.frames} frames to {a.out}")
