#!/usr/bin/env python
"""Writes a synthetic RGB-D sequence in Replica's on-disk layout (what the reference's loaders read
[REF mp_Tracker.py:341-354; utils/traj_utils.py:41-52; gs_icp_slam.py:135-144]):

    <out>/images/frame000000.jpg ...     8-bit RGB
    <out>/depth_images/depth000000.png   16-bit depth, metres x depth_scale
    <out>/traj.txt                       one row-major 4x4 camera-to-world pose per line
    <out>/caminfo.txt                    the three-line camera config the reference's --config expects

The scene is the analytic room of gs_icp_slam_amd/synth.py (there is no dataset in this image and no network).
    python tools/make_synth_dataset.py OUT [--frames 30] [--shape replica|tum] [--noise]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gs_icp_slam_amd import synth  # noqa: E402


def _write_frame(job):
    from PIL import Image
    out, shape, i, pose, noise, quality = job
    cfg = synth.REPLICA if shape == "replica" else synth.TUM
    rgb, d16 = synth.render_frame(cfg, pose, noise_seed=(100 + i) if noise else None, holes=0.15 if noise else 0.0)
    Image.fromarray(rgb, "RGB").save(os.path.join(out, "images", f"frame{i:06d}.jpg"), quality=quality)
    Image.fromarray(d16).save(os.path.join(out, "depth_images", f"depth{i:06d}.png"))
    return i


def write_dataset(out, frames=30, shape="replica", noise=False, quality=95):
    cfg = synth.REPLICA if shape == "replica" else synth.TUM
    os.makedirs(os.path.join(out, "images"), exist_ok=True)
    os.makedirs(os.path.join(out, "depth_images"), exist_ok=True)
    poses = synth.trajectory(frames)
    jobs = [(out, shape, i, pose, noise, quality) for i, pose in enumerate(poses)]
    workers = max(1, min(16, (os.cpu_count() or 2) // 2, frames))
    if workers > 1:   # the CPU ray-caster costs ~1 s per 1200x680 frame
        from concurrent.futures import ProcessPoolExecutor
        with ProcessPoolExecutor(max_workers=workers) as ex:
            list(ex.map(_write_frame, jobs))
    else:
        for j in jobs:
            _write_frame(j)
    with open(os.path.join(out, "traj.txt"), "w") as fh:
        for pose in poses:
            fh.write(" ".join(repr(float(v)) for v in pose.reshape(-1)) + "\n")
    with open(os.path.join(out, "caminfo.txt"), "w") as fh:   # third line is the one parsed [REF gs_icp_slam.py:52-63]
        fh.write("## camera parameters (synthetic room, %s-shaped)\nW H fx fy cx cy depth_scale depth_trunc dataset_type\n" % shape)
        fh.write(f"{cfg['W']} {cfg['H']} {cfg['fx']} {cfg['fy']} {cfg['cx']} {cfg['cy']} {cfg['depth_scale']} {cfg['depth_trunc']} replica\n")
    return cfg, poses


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--shape", choices=["replica", "tum"], default="replica")
    ap.add_argument("--noise", action="store_true")
    a = ap.parse_args()
    write_dataset(a.out, a.frames, a.shape, a.noise)
    print(f"wrote {a.frames} frames to {a.out}")
