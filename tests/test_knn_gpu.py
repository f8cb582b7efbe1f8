import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P", [1, 3, 4, 257, 5000])
def test_dist2_matches_oracle(P):
    import torch
    import oracle
    from simple_knn._C import distCUDA2
    pts = np.random.default_rng(P).normal(size=(P, 3)).astype(np.float32)
    got = distCUDA2(torch.from_numpy(pts).cuda()).cpu().numpy()
    want = oracle.knn_dist2(pts)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-12)
    perm = np.random.default_rng(1).permutation(P)
    got_p = distCUDA2(torch.from_numpy(pts[perm]).cuda()).cpu().numpy()
    np.testing.assert_allclose(got_p, got[perm], rtol=2e-6, atol=1e-12)   # permutation invariance


def test_dist2_rejects_cpu_tensor():
    import torch
    from simple_knn._C import distCUDA2
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros(4, 3))


def test_dist2_duplicates_surfaces_and_full_size():
    """Coincident points count as neighbours at distance 0; a surface-shaped cloud at the map's size (P = 300 k, the S-map means) runs the
    grid search (not an O(P^2) scan) and still equals the exhaustive definition on a sample of queries."""
    import time
    import torch
    import oracle
    from simple_knn._C import distCUDA2
    from gs_icp_slam_amd import synth
    rng = np.random.default_rng(5)
    base = rng.normal(size=(400, 3)).astype(np.float32)
    pts = np.concatenate([base, base[:150], base[:40]])               # duplicates and triplicates
    got = distCUDA2(torch.from_numpy(pts).cuda()).cpu().numpy()
    np.testing.assert_allclose(got, oracle.knn_dist2(pts), rtol=2e-6, atol=1e-12)
    assert (got[:40] == 0).sum() == 0 and np.all(got[:40] < got[200:400].mean())   # two coincident copies + one real neighbour
    big = synth.s_map(300_000, seed=2)["means3D"]
    t = torch.from_numpy(big).cuda()
    distCUDA2(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got_big = distCUDA2(t)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"distCUDA2 at P = 300 000: {1e3 * dt:.3f} ms")
    assert dt < 0.01                                                   # the O(P^2) scan this replaces took several ms of GPU time
    sample = rng.choice(300_000, 300, replace=False)
    d = ((big[sample, None, :].astype(np.float32) - big[None, :, :]) ** 2).sum(-1, dtype=np.float32)
    d[np.arange(300), sample] = np.inf
    want = np.sort(d, axis=1)[:, :3].sum(1) / 3.0
    np.testing.assert_allclose(got_big.cpu().numpy()[sample], want, rtol=1e-5)


def test_dist2_from_two_streams_back_to_back():
    """the scratch of distCUDA2 is per device and guarded by an event, so calls on DIFFERENT streams (the first still queued, the second
    growing the buffers) neither race on the buffers nor free memory a queued kernel reads."""
    import oracle
    import torch
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(5)
    clouds = [rng.normal(size=(n, 3)).astype(np.float32) for n in (3000, 90000, 700, 40000)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = []
    for i, c in enumerate(clouds):
        with torch.cuda.stream(streams[i % 2]):
            t = torch.from_numpy(c).cuda(non_blocking=True)
            outs.append((t, distCUDA2(t)))           # nothing synchronises between the calls
    torch.cuda.synchronize()
    for c, (_t, o) in zip(clouds, outs):
        np.testing.assert_allclose(o.cpu().numpy(), oracle.knn_dist2(c), rtol=1e-5, atol=1e-9)


@pytest.mark.gpu
def test_wave_sorting_network_and_lane_exchanges():
    """The k-NN kernels keep their candidate lists with a 64-lane bitonic network made of DPP / permlane exchanges on packed
    (distance bits << 32 | id) keys.  Known answers: every exchange pattern returns the value of lane l ^ J, and the network sorts any
    64 pairs — random, heavy ties (broken by id), idle pairs (FLT_MAX, INT_MAX), already sorted, reversed — exactly as a lexicographic sort."""
    import ctypes
    from gs_icp_slam_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3)
    fmax, imax = np.float32(np.finfo(np.float32).max), np.int32(2**31 - 1)
    cases = []
    cases.append((rng.random(64, dtype=np.float32) * 10, rng.permutation(64).astype(np.int32)))
    cases.append((rng.integers(0, 3, 64).astype(np.float32), rng.permutation(1000)[:64].astype(np.int32)))          # three distinct distances
    cases.append((np.zeros(64, np.float32), rng.permutation(64).astype(np.int32)[::-1].copy()))                     # all tied: order by id
    d = rng.random(64, dtype=np.float32); i = rng.permutation(64).astype(np.int32); d[20:] = fmax; i[20:] = imax      # mostly idle lanes
    cases.append((d, i))
    cases.append((np.arange(64, dtype=np.float32), np.arange(64, dtype=np.int32)))
    cases.append((np.arange(64, dtype=np.float32)[::-1].copy(), np.arange(64, dtype=np.int32)))
    cases.append((np.float32(1e-30) * rng.random(64, dtype=np.float32), rng.permutation(2**20)[:64].astype(np.int32)))   # denormal-range distances
    for d, i in cases:
        od, oi, ox = np.empty(64, np.float32), np.empty(64, np.int32), np.empty((7, 64), np.int32)
        rc = lib.gsicp_debug_wave_sort(d.ctypes.data_as(ctypes.c_void_p), i.ctypes.data_as(ctypes.c_void_p), od.ctypes.data_as(ctypes.c_void_p),
                                       oi.ctypes.data_as(ctypes.c_void_p), ox.ctypes.data_as(ctypes.c_void_p))
        _lib.check(rc, "gsicp_debug_wave_sort")
        lanes = np.arange(64)
        for row, J in enumerate((1, 2, 4, 8, 15, 16, 32)):
            assert np.array_equal(ox[row], 3 * (lanes ^ J) + 1), f"exchange pattern {J}"
        order = np.lexsort((i, d))
        assert np.array_equal(od, d[order]) and np.array_equal(oi, i[order])
