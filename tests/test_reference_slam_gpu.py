"""The reference's own two-process system, UNMODIFIED, on top of this repo's drop-in packages.

`gs_icp_slam_unlimit.py` builds the shared objects, spawns mp_Tracker + mp_Mapper [REF gs_icp_slam.py:81-131], the tracker
process drives `pygicp.FastGICP`, the mapper process drives `diff_gaussian_rasterization` through render_3 + torch loss +
torch.optim.Adam [REF mp_Mapper.py:219-248], `SharedGaussians` device tensors cross the process boundary by HIP-IPC
[REF scene/shared_objs.py:72-104], and at the end the reference prints System FPS / ATE RMSE / PSNR.  The data is a synthetic
sequence written in Replica's on-disk layout (no dataset in this image); third-party packages the image lacks are stand-ins
(tests/refstubs).  The reference tree is /root/reference when present, else the byte-code oracle/make_refpy.py compiled from it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, timeout=420):
    sys.path.insert(0, ROOT)
    from tools.run_reference_slam import find_reference
    if find_reference() is None:
        pytest.skip("no reference tree on this machine (neither /root/reference nor oracle/_ref/refpy)")
    log = os.path.join(ROOT, "gpurun_out", "reference_slam_%s.log" % "_".join(a.strip("-") for a in extra if a.startswith("--")))
    os.makedirs(os.path.dirname(log), exist_ok=True)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_reference_slam.py"), "--log", log, "--timeout", str(timeout - 60)] + extra,
                       capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines, f"harness printed no result line\nstdout: {p.stdout[-2000:]}\nstderr: {p.stderr[-6000:]}"
    res = json.loads(lines[-1])
    print("reference run:", json.dumps(res))
    assert res["status"] == "measured", f"reference run failed: {res}\n{p.stderr[-6000:]}"
    return res


def test_untouched_reference_two_process_run_replica_shaped():
    """300 Replica-shaped frames through gs_icp_slam.py (the 30-FPS-capped entry point [REF mp_Tracker.py:323-324]): ten seconds of mapping, so
    the map has to CONVERGE through the reference's own optimiser on the drop-in backward.  Thresholds sit within 20 % of what this run
    measured on an MI355X in round 3 (profiles/r03_reference_run_limit30_300.json: System FPS 30.0, ATE 0.01 cm, PSNR 33.78 dB, SSIM 0.971)."""
    res = _run(["--synthetic", "300", "--limit30"])
    assert res["processes_that_loaded_it"] >= 3, "parent, tracker process and mapper process must each load libgsicp_hip.so"
    assert 24.0 < res["system_fps"] <= 30.5, res["system_fps"]
    assert res["ate_rmse_cm"] <= 0.03, f"ATE {res['ate_rmse_cm']} cm on a noise-free synthetic sequence"    # printed x100: centimetres, two decimals
    assert res["psnr"] is not None and res["psnr"] > 27.0, f"PSNR {res['psnr']} dB: the map did not converge"
    assert res["ssim"] > 0.78, res["ssim"]


def test_untouched_reference_two_process_run_tum_layout_and_flags():
    """The reference's TUM branch end to end: dataset tag `tum`, rgb/ depth/ rgb.txt depth.txt groundtruth.txt read by its own association
    loader [REF utils/traj_utils.py:63-137; mp_Tracker.py:353-359], the flags of tum.sh [REF tum.sh:135-142], sensor-noise model + 15 % holes,
    30-FPS-capped entry point.  Thresholds within ~20-30 % of the round-3 measurement (profiles/r03_reference_run_tum_layout60.json:
    System FPS 29.99, ATE 0.23 cm, PSNR 21.29 dB, SSIM 0.948 after two seconds of mapping)."""
    res = _run(["--synthetic", "60", "--shape", "tum", "--noise", "--limit30"])
    assert res["dataset_type"] == "tum" and res["flags"]["trackable_opacity_th"] == 0.09 and res["flags"]["overlapped_th2"] == 1e-3
    assert res["processes_that_loaded_it"] >= 3
    assert 24.0 < res["system_fps"] <= 30.5, res["system_fps"]
    assert res["ate_rmse_cm"] < 0.30, res["ate_rmse_cm"]     # the real fr1_desk figure of the paper is 2.7 cm; this is the analytic room + the noise model
    assert res["psnr"] > 17.0 and res["ssim"] > 0.76, (res["psnr"], res["ssim"])


def test_untouched_reference_runs_on_compiled_extension_modules_at_all_three_boundaries():
    """`import pygicp` [REF mp_Tracker.py:10] resolving to the COMPILED pybind11 module (PyInit_pygicp, integration/pygicp_pybind.cpp) and
    `diff_gaussian_rasterization` / `simple_knn._C` [REF gaussian_renderer/__init__.py:14, scene/gaussian_model.py:20] to the packages around the
    COMPILED torch extension `_C` (integration/torch_ext_pybind.cpp) instead of the ctypes mirrors: the unmodified reference's tracker drives the
    first through the numpy API (it pickles into the spawned process [REF gs_icp_slam.py:121-127]), its mapper renders and back-propagates through
    the second, and the trajectory is tracked as with the mirrors."""
    res = _run(["--synthetic", "40", "--compiled-ext"])
    assert res["pygicp_binding"].startswith("compiled"), res
    assert res["raster_binding"].startswith("compiled"), res
    assert res["processes_that_loaded_it"] >= 2          # tracker (pygicp at import) and mapper (_C at its first render)
    assert res["ate_rmse_cm"] <= 0.03, res["ate_rmse_cm"]


def test_fused_rows_inside_the_reference_two_process_system():
    """SURVEY 8(f)'s rows EXECUTED inside the reference's own two processes: INTEGRATION.md 6-8's few-line edits are applied
    to the reference's files as an AST transform at build time (oracle/make_refpy.py --fused -> oracle/_ref/refpy_fused, byte-code only) and call
    gs_icp_slam_amd/refglue.py: the map in a GaussianStore(stable=True), FusedAdam(capturable), the iteration as ONE hipGraph replay captured once for
    the whole run, the tracker's new target as device tensors through HIP-IPC, the front-end kernel.  Same sequence and entry point as the untouched
    run of the test above.  Thresholds within ~20-30 % of the MI355X measurement of round 4 (400 frames unlimited: mapper iteration 0.57 ms median
    against 16.8 ms untouched, System FPS 135 against 113, ATE 0.01 cm both, PSNR 35.2 dB against 28.4)."""
    sys.path.insert(0, ROOT)
    from tools.run_reference_slam import find_reference
    if find_reference(fused=True) is None:
        pytest.skip("oracle/_ref/refpy_fused not built (python oracle/make_refpy.py where /root/reference exists)")
    res = _run(["--synthetic", "300", "--limit30", "--fused"])
    assert res["variant"].startswith("FUSED") and res["reference"].endswith("refpy_fused")
    assert res["processes_that_loaded_it"] >= 3
    assert 24.0 < res["system_fps"] <= 30.5, res["system_fps"]
    assert res["ate_rmse_cm"] <= 0.03, res["ate_rmse_cm"]
    fm = res["fused_mapper"]
    assert fm["graph_captures"] == 1, fm                       # keyframe growth and pruning never re-captured
    assert fm["median_ms_per_iteration"] < 1.2, fm             # 16.8 ms in the untouched system
    assert res["psnr"] > 33.0 and res["ssim"] > 0.95, (res["psnr"], res["ssim"])     # untouched, same run: 33.8 dB / 0.971
