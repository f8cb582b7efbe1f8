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


def test_gpus_n_without_a_launcher_starts_n_ranks():
    """`python bench.py --gpus 2 ...` — the driver's plain form — must start two ranks by itself (it silently ran one).  The probe
    mode stops each rank after the rendezvous, before any device work, so this runs without a GPU."""
    import json
    env = dict(os.environ, GSICP_BENCH_SPAWN_PROBE="1", GSICP_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["ranks"] == [0, 1] and res["distinct_processes"] == 2 and res["steps"] == 3 and res["warmup"] == 1


def test_under_torchrun_the_environment_decides_and_nothing_is_respawned():
    import json
    env = dict(os.environ, GSICP_BENCH_SPAWN_PROBE="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert res["n_gpus"] == 1 and res["distinct_processes"] == 1


def test_contract_line_stays_compact_on_a_worst_case_record():
    """the one JSON line had grown to 22-25 KB, the driver kept its last 7.9 KB and parsed nothing.  `compact_line` is what bench.py prints: fed with
    the LARGEST record on file (round 5's full line, sixteen legs, four system runs) it must stay under 6 000 characters, be one line, and carry the contract keys —
    `roofline.frac` and `cpu_baseline.value` among them; the full record goes to the legs file the line names."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_driver_cmd.json")))
    assert len(json.dumps(full)) > 20000
    full["fused_policy"] = {"a": "x" * 500}
    full["config"]["workload"] = full["config"]["workload"] * 3
    line = json.dumps(bench.compact_line(full, "/root/repo/bench_legs.json"))
    assert len(line) < 6000 and "\n" not in line, len(line)
    res = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
              "cpu_baseline", "legs_file"):
        assert k in res, k
    assert res["value"] == full["value"] and res["ms_per_step"] == full["ms_per_step"] and len(res["config"]["workload"]) <= 300
    assert res["roofline"]["frac"] == full["roofline"]["frac"] and res["roofline"]["bound"] == "hbm" and "traffic" in res["roofline"]
    assert res["cpu_baseline"]["value"] == full["cpu_baseline"]["value"] and res["cpu_baseline"]["kind"] == "port" and len(res["cpu_baseline"]["sample"]) <= 260
    assert res["render_bwd_ms_per_iter"] == full["render_bwd_ms_per_iter"] and res["render_bwd_ms_per_iter_trained_map"] > 0
    assert res["system_fps"] == full["system_fps"] and res["ate_cm"] == full["ate_cm"] and res["step_tum_ms"] == full["legs"]["step_tum"]["ms_per_step"]
    assert "legs" not in res
    # an empty record (a --only run, legs off) still yields a line
    small = bench.compact_line({"metric": "m", "value": 1.0, "config": {"workload": "w"}}, "/tmp/x.json")
    assert small["roofline"] is None and small["cpu_baseline"] is None and small["config"]["workload"] == "w"
