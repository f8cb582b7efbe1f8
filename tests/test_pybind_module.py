"""The COMPILED `pygicp` module (integration/pygicp_pybind.cpp -> PyInit_pygicp, built by gs_icp_slam_amd/build.py): CPU checks that it is a real
extension module exporting every method the reference's trackers call; the GPU test runs a frame through it and through the ctypes mirror and
compares bit for bit (both are thin bindings over the same C ABI)."""
import importlib.util
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INTEG = os.path.join(ROOT, "integration")


def _load():
    so = [f for f in os.listdir(INTEG) if f.startswith("pygicp.") and f.endswith(".so")]
    if not so:
        subprocess.check_call([sys.executable, "-m", "gs_icp_slam_amd.build"], cwd=ROOT)
        so = [f for f in os.listdir(INTEG) if f.startswith("pygicp.") and f.endswith(".so")]
    assert so, "integration/pygicp.*.so was not built"
    spec = importlib.util.spec_from_file_location("pygicp", os.path.join(INTEG, so[0]))    # a compiled module named pygicp: PyInit_pygicp
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_compiled_module_exports_every_method_the_reference_calls():
    mod = _load()
    assert mod.__file__.endswith(".so") and hasattr(mod, "FastGICP")
    names = {"set_max_correspondence_distance", "set_max_knn_distance", "set_input_target", "set_input_source", "set_target_filter", "set_source_filter",
             "calculate_target_covariance_with_filter", "get_target_rotationsq", "get_target_scales", "get_source_rotationsq", "get_source_scales",
             "set_target_covariances_fromqs", "align", "get_source_correspondence"}
    for f in ("mp_Tracker.py", "mp_Tracker_unlimit.py"):
        path = os.path.join("/root/reference", f)
        if os.path.exists(path):
            names |= set(re.findall(r"self\.reg\.([A-Za-z_0-9]+)\s*\(", open(path).read()))
    missing = sorted(n for n in names if not callable(getattr(mod.FastGICP, n, None)))
    assert not missing, missing


def _configured(mod):
    reg = mod.FastGICP()
    reg.set_max_correspondence_distance(0.5)     # overwritten by the caller after unpickling; the point is that the object survives pickling
    return reg


@pytest.mark.gpu
def test_compiled_module_equals_the_ctypes_mirror_and_pickles():
    import pickle
    import pygicp as mirror                      # the product's default binding (ctypes)
    from gs_icp_slam_amd import synth
    mod = _load()
    sys.modules.setdefault("pygicp_compiled_for_pickle", mod)
    sp = synth.s_pair(synth.REPLICA)
    pw = sp["points_a"].astype(np.float64) @ sp["pose_a"][:3, :3].T + sp["pose_a"][:3, 3]

    def filt(n, tr):
        f = np.zeros(n, np.int32)
        f[tr] = np.arange(1, len(tr) + 1)
        return f

    def unpickled():
        real, sys.modules["pygicp"] = sys.modules.get("pygicp"), mod      # pickle looks the class up by module name
        try:
            return pickle.loads(pickle.dumps(_configured(mod)))
        finally:
            sys.modules["pygicp"] = real
    outs = []
    for make in (mirror.FastGICP, mod.FastGICP, unpickled):
        reg = make()
        reg.set_max_correspondence_distance(0.02)
        reg.set_max_knn_distance(99999.0)
        reg.set_input_target(pw)                                                  # float64 [REF mp_Tracker.py:157]
        reg.set_target_filter(len(sp["trackable_a"]), filt(len(pw), sp["trackable_a"]))
        reg.calculate_target_covariance_with_filter()
        rq, sc = np.asarray(reg.get_target_rotationsq()), np.asarray(reg.get_target_scales())
        reg.set_input_source(sp["points_b"])                                      # float32 [REF mp_Tracker.py:191]
        reg.set_source_filter(len(sp["trackable_b"]), filt(len(sp["points_b"]), sp["trackable_b"]))
        T = np.asarray(reg.align(sp["pose_a"]))
        idx, d2 = reg.get_source_correspondence()
        srq = np.asarray(reg.get_source_rotationsq())
        outs.append((rq, sc, T, np.asarray(idx), np.asarray(d2), srq))
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert a.dtype == b.dtype and np.array_equal(a, b)
    assert outs[1][2].shape == (4, 4) and outs[1][2].dtype == np.float32
    with pytest.raises(RuntimeError):
        mod.FastGICP().set_input_target(np.zeros((5, 2), np.float32))
