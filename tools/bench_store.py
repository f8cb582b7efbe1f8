#!/usr/bin/env python
"""Map growth / pruning: reference-style torch.cat / boolean indexing of every parameter and Adam moment
vs the capacity-based GaussianStore, same values."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tests.test_store_gpu import LRS, RefModel, _rows, _step  # noqa: E402
from gs_icp_slam_amd.gaussian_store import GaussianStore  # noqa: E402

P, K = 300_000, 8_000
gen = torch.Generator(device="cuda").manual_seed(0)
first = _rows(P, 0, gen)
tm = torch.ones(P, dtype=torch.bool, device="cuda")
ref = RefModel(first, tm)
store = GaussianStore(P + 40 * K)
store.append(first, tm)
opt = store.attach(torch.optim.Adam, LRS, lr=0.0, eps=1e-15)
_step(ref.p, ref.opt, 1)
_step(store.params, opt, 1)
new = _rows(K, 0, gen)
tmk = torch.ones(K, dtype=torch.bool, device="cuda")


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def ref_cycle():
    ref.cat(new, tmk)
    ref.prune(torch.arange(ref.p["xyz"].shape[0], device="cuda") >= P)


def store_cycle():
    store.append(new, tmk)
    store.prune(torch.arange(store.n, device="cuda") >= P)


print(f"append {K} + prune back to {P}:  reference-style {timeit(ref_cycle):.3f} ms   GaussianStore {timeit(store_cycle):.3f} ms")
