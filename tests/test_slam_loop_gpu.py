"""End-to-end functional test: tools/slam_demo.py — front-end kernel, device-tensor tracker inputs, align, overlap statistics,
keyframe map growth through GaussianStore, mapper iterations as hipGraph replays, device-side map -> tracker hand-off — on synthetic
frames with a known trajectory.  The script asserts sub-millimetre tracking and a falling mapper loss itself."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_resident_slam_loop_tracks_a_known_trajectory():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "slam_demo.py"), "5"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and "slam demo OK" in r.stdout, tail
    assert r.stdout.count("keyframe:") == 2, tail
