#!/usr/bin/env python
"""Tracking-keyframe hand-off: the reference's host round trip vs the device call, same result.
  host  : get_trackable_gaussians_tensor (boolean indexing) -> .cpu() x3 -> numpy -> set_input_target + set_target_covariances_fromqs
  device: FastGICP.set_target_from_gaussians(...)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import pygicp  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
rng = np.random.default_rng(0)
xyz = torch.from_numpy(rng.uniform(-3, 3, (P, 3)).astype(np.float32)).cuda()
q = torch.nn.functional.normalize(torch.from_numpy(rng.normal(size=(P, 4)).astype(np.float32)).cuda())
s = torch.from_numpy((np.abs(rng.normal(0.03, 0.01, (P, 3))) + 1e-3).astype(np.float32)).cuda()
o = torch.from_numpy(rng.uniform(0, 1, (P, 1)).astype(np.float32)).cuda()
mask = torch.from_numpy(rng.uniform(0, 1, P) < 0.5).cuda()
th = 0.2
reg_h, reg_d = pygicp.FastGICP(), pygicp.FastGICP()


def host():
    keep = torch.logical_and((o > th).squeeze(-1), mask)
    tp, tr, ts = xyz[keep].cpu(), q[keep].cpu(), s[keep].cpu()
    reg_h.set_input_target(tp.numpy())
    reg_h.set_target_covariances_fromqs(tr.numpy().flatten(), ts.numpy().flatten())
    return tp.shape[0]


def device():
    return reg_d.set_target_from_gaussians(xyz, q, s, o, mask, th)


for name, fn in (("host round trip", host), ("device call", device)):
    for _ in range(3):
        n = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    print(f"{name:16s} P = {P}, {n} target points: {(time.perf_counter() - t0) / 20 * 1e3:8.3f} ms")
