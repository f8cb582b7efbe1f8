#!/usr/bin/env python
"""Measured experiment behind a design decision (north_star: "MFMA only for the batched 3x3 covariance contractions — choices evidenced"):
Sigma = R diag(s^2) R^T for P Gaussians on the vector ALU (one thread per Gaussian) vs on the matrix cores (v_mfma_f32_4x4x1_16B_f32, four
lanes per Gaussian).  Prints one JSON line; results agree to fp32 rounding.  Kernels: gs_icp_slam_amd/csrc/experiments/mfma_cov3.hip."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "gs_icp_slam_amd", "libgsicp_experiments.so"))
lib.gsicp_exp_cov3.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
lib.gsicp_exp_cov3.restype = ctypes.c_int
res = {}
for P in (8280, 300_000, 1_000_000):
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.nn.functional.normalize(torch.randn((P, 4), device="cuda", generator=g))
    s = torch.exp(torch.randn((P, 3), device="cuda", generator=g) * 0.5 - 4.0)
    outs, us = [], []
    for variant in (0, 1):
        out = torch.zeros((P, 6), device="cuda")
        t = ctypes.c_float(0)
        rc = lib.gsicp_exp_cov3(P, q.data_ptr(), s.data_ptr(), out.data_ptr(), variant, 200, ctypes.byref(t),
                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        outs.append(out)
        us.append(round(t.value, 3))
    err = float((outs[0] - outs[1]).abs().max() / outs[0].abs().max())
    res[str(P)] = {"valu_us": us[0], "mfma_4x4x1_16B_us": us[1], "max_rel_diff": err,
                   "bytes": 52 * P, "valu_GBps": round(52 * P / us[0] / 1e3, 1), "mfma_GBps": round(52 * P / us[1] / 1e3, 1)}
print(json.dumps({"experiment": "Sigma = R diag(s^2) R^T, fp32, VALU vs MFMA 4x4x1 (16 blocks)", "results": res}))
