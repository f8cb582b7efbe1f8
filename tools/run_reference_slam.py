#!/usr/bin/env python
"""Runs the reference's OWN, unmodified two-process system — `gs_icp_slam_unlimit.py` (or `gs_icp_slam.py`, capped at 30 FPS)
spawning mp_Tracker + mp_Mapper [REF gs_icp_slam.py:121-131] — with this repo's drop-in `pygicp`, `diff_gaussian_rasterization`
and `simple_knn` packages on PYTHONPATH, and captures the statistics it prints: System FPS / ATE RMSE [REF mp_Tracker.py:333-334]
and PSNR / SSIM [REF mp_Mapper.py:422].

    python tools/run_reference_slam.py --synthetic 30                     # writes a Replica-layout synthetic sequence first
    python tools/run_reference_slam.py --dataset /data/Replica/room0 --config <reference>/configs/Replica/caminfo.txt

The reference tree is taken from --reference, else /root/reference, else oracle/_ref/refpy (sourceless byte-code compiled from
/root/reference by oracle/make_refpy.py, because the GPU box has no /root/reference).  Nothing in it is edited; it is executed with
`python <reference>/gs_icp_slam_unlimit.py[c] <the flags of replica_unlimit.sh>`.  Third-party packages this image lacks
(cv2, open3d, rerun, torchmetrics, plyfile) are served by the stand-ins in tests/refstubs, appended LAST to PYTHONPATH.
Prints one JSON line {"system_fps": .., "ate_rmse_cm": .., "psnr": .., ...}; with no dataset and no --synthetic: "not measured".
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find_reference(explicit=None, fused=False):
    if fused:      # the AST-transformed byte-code of oracle/make_refpy.py --fused (INTEGRATION.md 6-8's edits applied; gs_icp_slam_amd/refglue.py)
        cand = os.path.join(ROOT, "oracle", "_ref", "refpy_fused")
        return cand if os.path.exists(os.path.join(cand, "mp_Tracker.pyc")) else None
    for cand in (explicit, os.environ.get("GSICP_REFERENCE"), "/root/reference", os.path.join(ROOT, "oracle", "_ref", "refpy")):
        if cand and (os.path.exists(os.path.join(cand, "mp_Tracker.py")) or os.path.exists(os.path.join(cand, "mp_Tracker.pyc"))):
            return cand
    return None


# the flag sets the reference's own launch scripts pass
REPLICA_FLAGS = dict(keyframe_th=0.7, knn_maxd=99999.0, overlapped_th=5e-4, max_correspondence_distance=0.02, trackable_opacity_th=0.05,
                     overlapped_th2=5e-5, downsample_rate=10)      # [REF replica.sh:135-142; replica_unlimit.sh:135-142]
TUM_FLAGS = dict(keyframe_th=0.81, knn_maxd=99999.0, overlapped_th=1e-3, max_correspondence_distance=0.03, trackable_opacity_th=0.09,
                 overlapped_th2=1e-3, downsample_rate=5)           # [REF tum.sh:135-142; tum_unlimit.sh]


def dataset_type(config):
    """`replica` / `tum`: the ninth token of the camera config's third line — what the reference itself branches on [REF gs_icp_slam.py:52-72]."""
    try:
        with open(config) as fh:
            return fh.readlines()[2].split()[8].strip().lower()
    except Exception:
        return "replica"


def flags_for(config, shape=None):
    """Flags by the DATASET (a real fr1_desk path gets tum.sh's flags, not Replica's).  A TUM-shaped sequence written in Replica's layout
    (`--shape tum --layout replica`) still gets tum.sh's flags: they belong to the sensor (640x480, stride 5, 3 cm gate), not to the file layout."""
    return dict(TUM_FLAGS if (dataset_type(config) == "tum" or shape == "tum") else REPLICA_FLAGS)


def run(reference, dataset, config, output, unlimit=True, timeout=600, extra_flags=(), flags=None, trace_dir=None, omp_threads=1, compiled_pygicp=False, compiled_ext=False):
    name = "gs_icp_slam_unlimit" if unlimit else "gs_icp_slam"
    script = os.path.join(reference, name + ".py")
    if not os.path.exists(script):
        script += "c"
    f = flags_for(config)
    f.update(flags or {})
    cmd = [sys.executable, "-W", "ignore", script, "--dataset_path", dataset, "--config", config, "--output_path", output]
    for k, v in f.items():
        cmd += [f"--{k}", str(v)]
    cmd += list(extra_flags)
    env = dict(os.environ)
    # --compiled-pygicp: `import pygicp` resolves to integration/pygicp.<abi>.so (PyInit_pygicp, the pybind11 binding over the C ABI) instead of the
    # ctypes mirror package at the repo root
    # --compiled-ext: additionally `diff_gaussian_rasterization` and `simple_knn._C` resolve to the packages around the compiled torch-extension
    # module `_C` (integration/torch_ext/, built from integration/torch_ext_pybind.cpp): all three native boundaries are then extension modules
    front = ([os.path.join(ROOT, "integration", "torch_ext")] if compiled_ext else []) + ([os.path.join(ROOT, "integration")] if compiled_pygicp or compiled_ext else [])
    env["PYTHONPATH"] = os.pathsep.join(front + [ROOT] + [p for p in env.get("PYTHONPATH", "").split(os.pathsep) if p] +
                                        [os.path.join(ROOT, "tests", "refstubs")])
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MPLBACKEND", "Agg")
    env["GSICP_ANNOUNCE"] = "1"
    env.setdefault("GSICP_ATE_DETAIL", "1")       # tests/refstubs/sitecustomize.py: print the true RMSE next to the reference's mean statistic
    # Three processes x torch's default intra-op pool (one thread per hardware thread, spinning after every parallel region) exhaust a
    # container's cgroup CPU quota within the first ~20 ms of every 100 ms scheduler period and the whole system then stalls for the
    # rest of it (measured: tracker frames and mapper iterations both alternate 10 ms / 90 ms).  The per-frame CPU work of the reference is
    # a handful of small torch ops; give each process a one-thread pool (measured on the 16-CPU-quota MI355X box, 96 synthetic frames:
    # default pool 8-10 FPS, 4 threads 15 FPS, 1 thread 84 FPS).  Environment only — no reference file is touched.
    if omp_threads > 0:
        env.setdefault("OMP_NUM_THREADS", str(omp_threads))
        env.setdefault("MKL_NUM_THREADS", str(omp_threads))
    if trace_dir:
        trace_dir = os.path.abspath(trace_dir)     # the reference runs with its own tree as working directory
        os.makedirs(trace_dir, exist_ok=True)
        env["GSICP_CALL_TRACE"] = trace_dir
    t0 = time.time()
    p = subprocess.Popen(cmd, cwd=reference, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
    try:
        out, _ = p.communicate(timeout=timeout)
        timed_out = False
    except subprocess.TimeoutExpired:
        import signal
        os.killpg(p.pid, signal.SIGKILL)     # the exact process group started above
        out, _ = p.communicate()
        timed_out = True
    res = dict(returncode=p.returncode, timed_out=timed_out, wall_s=round(time.time() - t0, 2), script=os.path.basename(script),
               reference=reference, loaded_so=sorted(set(re.findall(r"GSICP_LOADED (\S+)", out))),
               pygicp_binding=("compiled pybind11 module (integration/pygicp.*.so)" if re.search(r"via=compiled-pygicp", out) else "ctypes mirror (pygicp/)"),
               raster_binding=("compiled torch extension (integration/torch_ext/diff_gaussian_rasterization/_C.*.so)" if re.search(r"via=compiled-torch-ext", out)
                               else "ctypes mirror (diff_gaussian_rasterization/)"),
               processes_that_loaded_it=len(set(re.findall(r"GSICP_LOADED \S+ pid=(\d+)", out))))
    for key, pat in (("system_fps", r"System FPS:\s*([-\d.eE+naninf]+)"), ("ate_rmse_cm", r"ATE RMSE:\s*([-\d.eE+naninf]+)"),
                     ("psnr", r"PSNR:\s*([-\d.eE+naninf]+)"), ("ssim", r"SSIM:\s*([-\d.eE+naninf]+)")):
        m = re.search(pat, out)
        res[key] = float(m.group(1)) if m else None
    # GSICP_ATE_DETAIL=1 (tests/refstubs/sitecustomize.py): the reference's "ATE RMSE" is the MEAN aligned translation error [REF mp_Tracker.py:479];
    # the true RMSE / median / maximum of the same per-frame errors are printed next to it
    m = re.search(r"GSICP_FUSED_MAPPER iterations (\d+) median_ms ([-\d.eE+]+) mean_ms ([-\d.eE+]+) p90_ms ([-\d.eE+]+) captures (\d+) gaussians (\d+)", out)
    if m:
        res["fused_mapper"] = dict(iterations=int(m.group(1)), median_ms_per_iteration=float(m.group(2)), mean_ms_per_iteration=float(m.group(3)),
                                   p90_ms_per_iteration=float(m.group(4)), graph_captures=int(m.group(5)), gaussians=int(m.group(6)))
        m = re.search(r"gpu_median_ms ([-\d.eE+na]+) gpu_p90_ms ([-\d.eE+na]+) paced_waits (\d+) iters_per_frame (\S+) policy (\S+)", out)
        if m:      # median_ms_per_iteration is the loop's CADENCE (pacing included); gpu_* is the device time of set_view + the graph replay
            res["fused_mapper"].update(gpu_median_ms_per_iteration=float(m.group(1)), gpu_p90_ms_per_iteration=float(m.group(2)),
                                       paced_waits=int(m.group(3)), iters_per_frame_budget=float(m.group(4)), policy=m.group(5))
    m = re.search(r"ATE detail: true_rmse_cm ([-\d.eE+]+) mean_cm ([-\d.eE+]+) median_cm ([-\d.eE+]+) max_cm ([-\d.eE+]+)", out)
    if m:
        res.update(ate_true_rmse_cm=float(m.group(1)), ate_mean_cm=float(m.group(2)), ate_median_cm=float(m.group(3)), ate_max_cm=float(m.group(4)),
                   ate_statistic_printed_by_the_reference="mean of the aligned translation errors (labelled 'ATE RMSE')")
    return res, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=None)
    ap.add_argument("--dataset", default=None)
    ap.add_argument("--config", default=None)
    ap.add_argument("--output", default=None)
    ap.add_argument("--synthetic", type=int, default=0, help="write and use a synthetic Replica-layout sequence of this many frames")
    ap.add_argument("--shape", choices=["replica", "tum"], default="replica")
    ap.add_argument("--layout", choices=["replica", "tum"], default=None, help="on-disk layout of the synthetic sequence (default: the shape's own; "
                    "`tum` = rgb/ depth/ rgb.txt depth.txt groundtruth.txt and dataset tag `tum`, the reference's TUM loader branch)")
    ap.add_argument("--cache", default=None, help="directory that keeps synthetic sequences between runs: a run of N frames re-uses (symlinks) the "
                    "first N frames of a longer cached sequence of the same kind instead of ray-casting again")
    ap.add_argument("--noise", action="store_true")
    ap.add_argument("--speed", type=float, default=1.0, help="synthetic sequence: motion per frame relative to the default (~7 mm / 0.25 deg); 2 = a fast hand-held sweep")
    ap.add_argument("--jitter", type=float, default=0.0, help="synthetic sequence: hand tremor per frame (sigma in metres of translation, x 10 in degrees of rotation)")
    ap.add_argument("--limit30", action="store_true", help="run gs_icp_slam.py (tracker capped at 30 FPS [REF mp_Tracker.py:323]) instead of the _unlimit variant")
    ap.add_argument("--timeout", type=float, default=600.0)
    ap.add_argument("--trace", default=None, help="directory for the drop-in call trace (GSICP_CALL_TRACE): one file per process")
    ap.add_argument("--omp-threads", type=int, default=1, help="OMP_NUM_THREADS for the reference's processes (0 = leave the environment alone)")
    ap.add_argument("--log", default=None, help="write the reference's full stdout here")
    ap.add_argument("--fused", action="store_true", help="run the reference with INTEGRATION.md 6-8's few-line edits applied (oracle/make_refpy.py --fused: fused mapper "
                    "iteration as one hipGraph over a GaussianStore, device-resident target hand-off, front-end kernel) instead of the untouched files")
    ap.add_argument("--policy", choices=["free", "freeze", "budget"], default=None, help="--fused: GSICP_FUSED_POLICY of the fused mapper (gs_icp_slam_amd/refglue.py; "
                    "default `free` = the reference's optimiser; `freeze` / `budget` deviate from it)")
    ap.add_argument("--compiled-pygicp", action="store_true", help="let the reference's `import pygicp` resolve to the compiled pybind11 module "
                    "integration/pygicp.<abi>.so instead of the ctypes mirror package")
    ap.add_argument("--compiled-ext", action="store_true", help="all three native boundaries as compiled extension modules: --compiled-pygicp plus "
                    "`diff_gaussian_rasterization` / `simple_knn._C` from integration/torch_ext/ (the pybind11 torch extension `_C` over the C ABI)")
    a = ap.parse_args()
    if a.policy:
        os.environ["GSICP_FUSED_POLICY"] = a.policy
    ref = find_reference(a.reference, fused=a.fused)
    if ref is None:
        print(json.dumps({"status": "not measured", "why": "no reference tree (/root/reference or oracle/_ref/refpy) on this machine"}))
        return 0
    tmp = None
    flags = {}
    if a.synthetic > 0:
        sys.path.insert(0, ROOT)
        from tools.make_synth_dataset import write_dataset
        from tools.make_synth_dataset import subset_dataset
        layout = a.layout or a.shape
        tmp = tempfile.mkdtemp(prefix="gsicp_synth_")
        cached = None
        if a.cache and layout == "replica":
            kind = f"{a.shape}_{'noisy' if a.noise else 'clean'}_s{a.speed:g}_j{a.jitter:g}_"
            os.makedirs(a.cache, exist_ok=True)
            have = sorted((int(d[len(kind):]), d) for d in os.listdir(a.cache) if d.startswith(kind) and d[len(kind):].isdigit())
            fit = [d for n, d in have if n >= a.synthetic]
            if fit:
                cached = os.path.join(a.cache, fit[0])
            else:
                cached = os.path.join(a.cache, kind + str(a.synthetic))
                write_dataset(cached, a.synthetic, a.shape, a.noise, layout=layout, speed=a.speed, jitter=a.jitter)
            subset_dataset(cached, a.synthetic, tmp)
        else:
            write_dataset(tmp, a.synthetic, a.shape, a.noise, layout=layout, speed=a.speed, jitter=a.jitter)
        a.dataset, a.config = tmp, os.path.join(tmp, "caminfo.txt")
        flags = flags_for(a.config, a.shape)
    if not a.dataset or not os.path.isdir(a.dataset):
        print(json.dumps({"status": "not measured", "why": f"dataset {a.dataset!r} not present on this machine"}))
        return 0
    if not a.config:
        a.config = os.path.join(ref, "configs", "Replica", "caminfo.txt")
    out_dir = a.output or tempfile.mkdtemp(prefix="gsicp_out_")
    res, log = run(ref, a.dataset, a.config, out_dir, unlimit=not a.limit30, timeout=a.timeout, flags=flags, trace_dir=a.trace, omp_threads=a.omp_threads,
                   compiled_pygicp=a.compiled_pygicp, compiled_ext=a.compiled_ext)
    res.update(status="measured" if res["returncode"] == 0 and res["system_fps"] is not None else "failed", dataset=a.dataset,
               data="synthetic" if a.synthetic else "real", frames=a.synthetic or None,
               synthetic_motion=(dict(speed=a.speed, jitter_m=a.jitter, sensor_noise=bool(a.noise)) if a.synthetic else None), dataset_type=dataset_type(a.config),
               flags=flags or flags_for(a.config), entry="gs_icp_slam.py (30 FPS cap)" if a.limit30 else "gs_icp_slam_unlimit.py",
               variant="FUSED: INTEGRATION.md 6-8's edits applied by oracle/make_refpy.py --fused" if a.fused else "untouched reference files")
    if a.log:
        with open(a.log, "w") as fh:
            fh.write(log)
    if res["status"] != "measured":
        sys.stderr.write(log[-6000:])
    print(json.dumps(res))
    return 0 if res["status"] == "measured" else 1


if __name__ == "__main__":
    sys.exit(main())
