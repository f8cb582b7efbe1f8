"""CPU: the C-ABI shared library loads, exports every symbol include/gsicp_hip.h declares, and the product path fails
loudly (no CPU fallback) when there is no HIP device.  No compute is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "gsicp_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gsicp_[a-z0-9_]+)\s*\(", txt)) - {"gsicp_resize_fn"})


def test_library_exports_every_declared_symbol():
    from gs_icp_slam_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build with `python -m gs_icp_slam_amd.build`"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/gsicp_hip.h but not exported"
    assert set(_lib.SIGNATURES) == set(syms), set(_lib.SIGNATURES) ^ set(syms)
    assert _lib.load().gsicp_abi_version() == 5


def test_drop_in_packages_expose_reference_names():
    import diff_gaussian_rasterization as dgr
    import pygicp
    import simple_knn._C as knn
    fields = dgr.GaussianRasterizationSettings._fields[:12]
    assert fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
                      "sh_degree", "campos", "prefiltered", "debug")           # REF gaussian_renderer/__init__.py:244-257
    for m in ("set_max_correspondence_distance", "set_max_knn_distance", "set_input_target", "set_input_source", "set_target_filter",
              "set_source_filter", "calculate_target_covariance_with_filter", "get_target_rotationsq", "get_target_scales",
              "get_source_rotationsq", "get_source_scales", "set_target_covariances_fromqs", "align", "get_source_correspondence"):
        assert callable(getattr(pygicp.FastGICP, m)), m                          # REF mp_Tracker.py:53-308
    assert callable(knn.distCUDA2) and callable(dgr.GaussianRasterizer.markVisible)


def test_product_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present; this test checks the no-device behaviour")
    import pygicp
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from simple_knn._C import distCUDA2
    with pytest.raises(RuntimeError):
        pygicp.FastGICP()
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros(8, 3))
    rs = GaussianRasterizationSettings(image_height=16, image_width=16, tanfovx=1.0, tanfovy=1.0, bg=torch.zeros(3), scale_modifier=1.0,
                                       viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=0, campos=torch.zeros(3), prefiltered=False,
                                       debug=False)
    with pytest.raises(RuntimeError):
        GaussianRasterizer(rs)(means3D=torch.zeros(2, 3), means2D=torch.zeros(2, 3), shs=torch.zeros(2, 1, 3), opacities=torch.ones(2, 1),
                               scales=torch.ones(2, 3), rotations=torch.tensor([[0, 0, 0, 1.0]] * 2))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gs_icp_slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), f
                assert "liboracle" not in src, f
