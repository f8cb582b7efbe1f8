"""CPU: the driver's bench contract at the command-line level (no GPU needed): `python bench.py --gpus N --steps K --warmup W` must parse,
and the tools bench.py spawns must at least compile."""
import os
import py_compile
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_accepts_the_contract_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    for flag in ("--gpus", "--steps", "--warmup", "--repeats", "--lockstep", "--no-graph", "--pair", "--only"):
        assert flag in out.stdout, flag


def test_scripts_the_bench_and_the_capture_depend_on_compile():
    for rel in ("bench.py", "__graft_entry__.py", "tools/rccl_graph_probe.py", "tools/collect_profiles.py", "tools/tracker_latency.py",
                "tools/run_reference_slam.py", "tools/slam_demo.py"):
        py_compile.compile(os.path.join(ROOT, rel), doraise=True)
