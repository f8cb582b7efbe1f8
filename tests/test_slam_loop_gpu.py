"""End-to-end functional test: tools/slam_demo.py — front-end kernel, device-tensor tracker inputs, align, overlap statistics, the
reference's keyframe rules, map growth / pruning through GaussianStore(stable=True), mapper iterations as replays of ONE captured hipGraph
for the whole run, device-side map -> tracker hand-off — on synthetic frames with a known trajectory.  The script asserts sub-1.5 mm
tracking, a falling mapper loss and zero graph re-captures itself."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_resident_slam_loop_one_graph_for_the_whole_run():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "slam_demo.py"), "52", "--iters", "5", "--prune-every", "120"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0 and "slam demo OK" in r.stdout, tail
    m = re.search(r"(\d+) tracking \+ (\d+) mapping keyframes, (\d+) prune\(s\), (\d+) mapper iterations, (\d+) Gaussians, graph captures (\d+), re-captures (\d+)", r.stdout)
    assert m, tail
    n_tr, n_map, prunes, iters, n_g, captures, recaptures = map(int, m.groups())
    assert n_tr + n_map >= 5 and prunes >= 1 and captures == 1 and recaptures == 0, tail
    print(r.stdout[-1500:])
