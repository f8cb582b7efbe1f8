"""CPU, 2 processes over gloo: the tile-sharded mapper path (gs_icp_slam_amd/sharded.py) must reproduce the single-process
result — same image on every rank, and gradients that sum to the unsharded gradients.  The collectives and autograd
plumbing are the product's; the per-rank rasteriser is an oracle-backed stand-in (test infrastructure) because the HIP
kernels need a GPU.  The same property is checked on real hardware in tests/test_raster_gpu.py (tile sharding test)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gs_icp_slam_amd import synth
from tests import util


def _tile_mask(W, H, mod, rem):
    from gs_icp_slam_amd.sharded import tile_owner_mask      # the ownership rule of csrc/raster_common.hpp in torch: round-robin super-tiles or a band
    return tile_owner_mask(W, H, mod, rem)


class _OracleRasterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, shs, opacities, scales, rotations, rs):
        g = dict(means3D=means3D.detach().numpy(), shs=shs.detach().numpy(), opacities=opacities.detach().numpy(),
                 scales=scales.detach().numpy(), rotations=rotations.detach().numpy())
        cam = dict(viewmatrix=rs.viewmatrix.numpy(), projmatrix=rs.projmatrix.numpy(), campos=rs.campos.numpy(), tanfovx=rs.tanfovx,
                   tanfovy=rs.tanfovy, W=rs.image_width, H=rs.image_height)
        o = util.oracle_forward(g, cam, rs.bg.numpy(), 0)
        mask = _tile_mask(rs.image_width, rs.image_height, rs.tile_mod, rs.tile_rem)
        ctx.g, ctx.cam, ctx.rs, ctx.mask = g, cam, rs, mask
        color = torch.from_numpy(o["color"]) * mask
        depth = torch.from_numpy(o["depth"])[None] * mask
        return depth, color, torch.from_numpy(o["radii"]), torch.from_numpy(o["is_used"])

    @staticmethod
    def backward(ctx, g_depth, g_color, *_):
        gc = (g_color * ctx.mask).numpy()
        gd = (g_depth[0] * ctx.mask).numpy()
        b = util.oracle_backward(ctx.g, ctx.cam, ctx.rs.bg.numpy(), gc, gd, 0)
        t = torch.from_numpy
        return (t(b["dL_dmeans3D"]), t(b["dL_dmeans2D"]), t(b["dL_dsh"]), t(b["dL_dopacity"])[:, None], t(b["dL_dscales"]),
                t(b["dL_drots"]), None)


class _OracleRasterizer(torch.nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.rs = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
        return _OracleRasterFn.apply(means3D, means2D, shs, opacities, scales, rotations, self.rs)


def _settings(cam):
    from gs_icp_slam_amd.rasterizer import GaussianRasterizationSettings
    return GaussianRasterizationSettings(
        image_height=cam["H"], image_width=cam["W"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.tensor([0.1, 0.2, 0.3]),
        scale_modifier=1.0, viewmatrix=torch.from_numpy(cam["viewmatrix"]), projmatrix=torch.from_numpy(cam["projmatrix"]), sh_degree=0,
        campos=torch.from_numpy(cam["campos"]), prefiltered=False, debug=False)


def _run(rasterizer, g, target):
    t = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in g.items()}
    m2d = torch.zeros_like(t["means3D"], requires_grad=True)
    depth, color, radii, used = rasterizer(means3D=t["means3D"], means2D=m2d, shs=t["shs"], opacities=t["opacities"], scales=t["scales"],
                                           rotations=t["rotations"])
    loss = (color - target[:3]).abs().mean() + 0.1 * (depth - target[3:]).abs().mean()
    loss.backward()
    return color.detach(), depth.detach(), {k: v.grad.clone() for k, v in t.items()}, used


def _run_flagged(rasterizer, g, target, trip):
    """One forward / backward in which this rank's duplicate-count guard is tripped (trip=True) or not."""
    t = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in g.items()}
    m2d = torch.zeros_like(t["means3D"], requires_grad=True)
    depth, color, radii, used = rasterizer(means3D=t["means3D"], means2D=m2d, shs=t["shs"], opacities=t["opacities"], scales=t["scales"],
                                           rotations=t["rotations"])
    rasterizer.holder.guard = (torch.tensor([11 if trip else 9], dtype=torch.int32), 10)     # count 11 > capacity 10 on the tripping rank
    ((color - target[:3]).abs().mean() + 0.1 * (depth - target[3:]).abs().mean()).backward()


def _scene():
    g = synth.random_gaussians(120, seed=4)
    g["means3D"][:30, 2] = -2.0        # a quarter of the map is behind the camera: culled rows must not travel in the gradient all-reduce
    return g


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gs_icp_slam_amd.sharded import ShardedGaussianRasterizer
    cam = synth.make_camera(80, 48, 64.0, 64.0)
    g = _scene()
    target = torch.from_numpy(np.random.default_rng(0).random((4, 48, 80)).astype(np.float32))
    sh = ShardedGaussianRasterizer(_settings(cam), rasterizer_cls=_OracleRasterizer)            # defaults: compacted gradient all-reduce
    color, depth, grads, used = _run(sh, g, target)
    vol_compact = sh.holder.last_volume_bytes
    dense = ShardedGaussianRasterizer(_settings(cam), rasterizer_cls=_OracleRasterizer, compact_grads=False, sync_is_used=True)
    color_d, depth_d, grads_d, used_d = _run(dense, g, target)
    assert torch.equal(color, color_d) and torch.equal(depth, depth_d)
    for k in grads:   # the compacted all-reduce (visible rows only) must equal the dense one bit for bit
        assert torch.equal(grads[k], grads_d[k]), k
    assert vol_compact < dense.holder.last_volume_bytes
    # static block (what a captured hipGraph replays): same sums, fixed volume, and an overflow on ONE rank is seen by both
    n_vis = int((sh.holder.radii > 0).sum())
    static = ShardedGaussianRasterizer(_settings(cam), rasterizer_cls=_OracleRasterizer, vis_capacity=n_vis + 5)
    color_s, depth_s, grads_s, _ = _run(static, g, target)
    assert torch.equal(color, color_s) and torch.equal(depth, depth_s)
    for k in grads:
        assert torch.equal(grads_s[k], grads_d[k]), k
    assert static.holder.last_volume_bytes == ((n_vis + 5) * 17 + 1 + 4) * 4 and int(static.holder.overflow.item()) == 0   # rows | flag | 4 loss words
    flagged = ShardedGaussianRasterizer(_settings(cam), rasterizer_cls=_OracleRasterizer, vis_capacity=n_vis + 5)
    _run_flagged(flagged, g, target, trip=(rank == 1))      # only rank 1's duplicate lists "overflow"
    assert int(flagged.holder.overflow.item()) == 1, "an overflow on rank 1 must reach rank 0 through the flag word"
    short = ShardedGaussianRasterizer(_settings(cam), rasterizer_cls=_OracleRasterizer, vis_capacity=n_vis - 1)
    _run(short, g, target)                                   # fewer rows than visible Gaussians: flagged on both
    assert int(short.holder.overflow.item()) == 1
    q.put((rank, color.numpy(), depth.numpy(), {k: v.numpy() for k, v in grads.items()}, used.numpy(), used_d.numpy(),
           vol_compact, dense.holder.last_volume_bytes))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=240) for _ in procs], key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cam = synth.make_camera(80, 48, 64.0, 64.0)
    g = _scene()
    target = torch.from_numpy(np.random.default_rng(0).random((4, 48, 80)).astype(np.float32))
    color, depth, grads, used = _run(_OracleRasterizer(_settings(cam)), g, target)
    for rank, c, d, gr, u_local, u_synced, vol_c, vol_d in outs:
        assert np.array_equal(c, color.numpy()) and np.array_equal(d, depth.numpy()), f"rank {rank}: image differs"
        assert np.array_equal(u_synced, used.numpy())
        assert vol_c < vol_d
        for k in grads:
            np.testing.assert_allclose(gr[k], grads[k].numpy(), rtol=1e-5, atol=1e-7 * (np.abs(grads[k].numpy()).max() + 1e-30))
    for k in grads:   # both ranks hold identical (all-reduced) gradients
        assert np.array_equal(outs[0][3][k], outs[1][3][k])
    assert np.array_equal(np.maximum(outs[0][4], outs[1][4]), used.numpy())   # per-rank is_used flags OR to the full ones


def _kf_cam(rank):
    return synth.make_camera(80, 48, 64.0, 64.0, synth.se3((0.0, 4.0 * rank, 0.0), (0.05 * rank, 0.0, 0.0)))


def _kf_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gs_icp_slam_amd.sharded import KeyframeParallelRasterizer
    g = _scene()
    target = torch.from_numpy(np.random.default_rng(10 + rank).random((4, 48, 80)).astype(np.float32))
    kp = KeyframeParallelRasterizer(_settings(_kf_cam(rank)), rasterizer_cls=_OracleRasterizer)
    guard = kp.overflow_guard()
    color, depth, grads, used = _run(kp, g, target)
    vol = kp.holder.last_volume_bytes
    flag0 = int(guard[0].item())
    # a rank whose duplicate lists overflowed: the flag word travels with the gradients and both ranks see it
    kp2 = KeyframeParallelRasterizer(_settings(_kf_cam(rank)), rasterizer_cls=_OracleRasterizer)
    guard2 = kp2.overflow_guard()
    t = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in g.items()}
    d2, c2, _, _ = kp2(means3D=t["means3D"], means2D=torch.zeros_like(t["means3D"], requires_grad=True), shs=t["shs"], opacities=t["opacities"],
                       scales=t["scales"], rotations=t["rotations"])
    kp2.holder.guard = (torch.tensor([11 if rank == 1 else 9], dtype=torch.int32), 10)
    (c2.mean() + d2.mean()).backward()
    q.put((rank, color.numpy(), {k: v.numpy() for k, v in grads.items()}, vol, flag0, int(guard2[0].item())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_keyframe_parallel_sums_the_views_gradients():
    """Throughput mode (SURVEY 8e alternative): each rank renders its own keyframe; the parameters' gradients on every rank equal the SUM of
    the two views' single-process gradients; each rank keeps its own image; one dense all-reduce of 14 floats per Gaussian + a flag word."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_kf_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=240) for _ in procs], key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = _scene()
    single = []
    for rank in range(2):
        target = torch.from_numpy(np.random.default_rng(10 + rank).random((4, 48, 80)).astype(np.float32))
        single.append(_run(_OracleRasterizer(_settings(_kf_cam(rank))), g, target))
    P = g["means3D"].shape[0]
    for rank, color, grads, vol, flag0, flag_tripped in outs:
        assert np.array_equal(color, single[rank][0].numpy()), f"rank {rank} must render ITS OWN view"
        assert vol == (P * 14 + 1) * 4 and flag0 == 0 and flag_tripped == 1
        for k in grads:
            want = single[0][2][k].numpy() + single[1][2][k].numpy()
            np.testing.assert_allclose(grads[k], want, rtol=1e-6, atol=1e-7 * (np.abs(want).max() + 1e-30))
    for k in outs[0][2]:
        assert np.array_equal(outs[0][2][k], outs[1][2][k])
    assert not np.array_equal(outs[0][1], outs[1][1])


# ---------------------------------------------------------------------------------------------------------------- band mode (round 5)
def _band_worker(rank, world, port, q, H, bands):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gs_icp_slam_amd.sharded import BAND_HALO, ShardedGaussianRasterizer, tile_owner_mask
    W = 80
    cam = synth.make_camera(W, H, 64.0, 64.0)
    g = _scene()
    target = torch.from_numpy(np.random.default_rng(0).random((4, H, W)).astype(np.float32))
    sh = ShardedGaussianRasterizer(_settings(cam), rasterizer_cls=_OracleRasterizer, bands=bands)
    own = tile_owner_mask(W, H, *sh.loss_shard())
    t = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in g.items()}
    m2d = torch.zeros_like(t["means3D"], requires_grad=True)
    depth, color, radii, used = sh(means3D=t["means3D"], means2D=m2d, shs=t["shs"], opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
    # every rank takes the loss of ITS OWN band (as the sharded loss kernel does); the shares sum to the full-image loss
    n = float(3 * H * W)
    loss = (((color - target[:3]).abs() * own).sum() / n) + 0.1 * (((depth - target[3:]).abs() * own).sum() / float(H * W))
    loss.backward()
    q.put((rank, color.detach().numpy(), depth.detach().numpy(), own.numpy(), {k: v.grad.numpy() for k, v in t.items()}, float(loss.detach()),
           sh.holder.last_image_bytes, sh.bands, sh.holder.last_halo_bytes_received))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,H,bands", [(2, 48, "equal"), (3, 128, [0, 1, 3, 4]), (8, 256, "equal")])
def test_band_sharding_exchanges_halos_not_images(world, H, bands):
    """contiguous bands of super-tile rows + a halo exchange instead of round-robin super-tiles + an all-gather of the image.
    On every rank the image equals the single-process image BIT FOR BIT on its band and on the BAND_HALO rows above and below it (everything its loss
    blocks read); the bytes a rank contributes are 2 x BAND_HALO x W x 4 floats whatever the image height, and it receives
    its two neighbours' chunks only (neighbour send / recv inside one all_to_all_single), whatever the world size; the ranks' own-band losses sum to the
    full loss and the all-reduced gradients equal the single-process gradients."""
    from gs_icp_slam_amd.sharded import BAND_HALO
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_band_worker, args=(r, world, port, q, H, bands)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=300) for _ in procs], key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    W = 80
    cam = synth.make_camera(W, H, 64.0, 64.0)
    g = _scene()
    target = torch.from_numpy(np.random.default_rng(0).random((4, H, W)).astype(np.float32))
    color, depth, grads, used = _run(_OracleRasterizer(_settings(cam)), g, target)
    color, depth = color.numpy(), depth.numpy()
    covered = np.zeros((H, W), bool)
    total = 0.0
    for rank, c, d, own, gr, loss, nbytes, b, received in outs:
        assert nbytes == 2 * 4 * BAND_HALO * W * 4, nbytes                       # this rank's contribution to the exchange: independent of H
        # ... and what it RECEIVES: its two neighbours' rows only (one neighbour at the top and the bottom of the image), whatever the world size
        # (round 5's all_gather received world x 2 chunks: VERDICT r5 weak 12)
        assert received == ((rank > 0) + (rank + 1 < world)) * 4 * BAND_HALO * W * 4, (rank, received)
        rows = np.where(own.any(1))[0]
        y0, y1 = rows.min(), rows.max() + 1
        assert (y0, y1) == (32 * b[rank], min(H, 32 * b[rank + 1])) and own[y0:y1].all() and not covered[own].any()
        covered |= own
        lo, hi = max(0, y0 - BAND_HALO), min(H, y1 + BAND_HALO)
        assert np.array_equal(c[:, lo:hi], color[:, lo:hi]) and np.array_equal(d[:, lo:hi], depth[:, lo:hi]), f"rank {rank}: band + halo differ"
        total += loss
        for k in grads:
            # the sum of `world` fp32 partial gradients against one fp32 chain: rounding of the largest partial, relative to the tensor's scale
            np.testing.assert_allclose(gr[k], grads[k].numpy(), rtol=2e-5, atol=1e-6 * (np.abs(grads[k].numpy()).max() + 1e-30))
    assert covered.all()
    full = float((torch.from_numpy(color) - target[:3]).abs().mean() + 0.1 * (torch.from_numpy(depth) - target[3:]).abs().mean())
    assert abs(total - full) <= 1e-5 * abs(full)
    for k in outs[0][4]:
        assert all(np.array_equal(outs[0][4][k], o[4][k]) for o in outs[1:])       # every rank holds the same all-reduced gradients


def test_balanced_bands_follow_the_load():
    from gs_icp_slam_amd.sharded import balanced_bands, band_code, equal_bands, tile_owner_mask
    assert equal_bands(680, 8) == [0, 3, 6, 8, 11, 14, 16, 19, 22] and equal_bands(480, 2) == [0, 8, 15]
    assert balanced_bands([1, 1, 1, 10, 1, 1, 1, 1], 3) == [0, 3, 4, 8]                 # the heavy row gets a band of its own
    b = balanced_bands([5.0] * 22, 8)
    assert b[0] == 0 and b[-1] == 22 and all(2 <= b[i + 1] - b[i] <= 3 for i in range(8))
    load = np.random.default_rng(1).integers(0, 50000, 22).tolist()
    b = balanced_bands(load, 8)
    sums = [sum(load[b[i]:b[i + 1]]) for i in range(8)]
    assert all(b[i + 1] > b[i] for i in range(8)) and max(sums) <= sum(load) / 8 + max(load)
    with pytest.raises(RuntimeError):
        balanced_bands([1, 2], 3)
    heavy_bottom = [1.0] * 21 + [100.0]                    # 680 rows: super-tile row 21 holds 8 pixel rows — not a band (BAND_HALO = 10) on its own
    assert balanced_bands(heavy_bottom, 4) == [0, 19, 20, 21, 22] and balanced_bands(heavy_bottom, 4, H=680) == [0, 18, 19, 20, 22] and balanced_bands(heavy_bottom, 4, H=704)[-2] == 21
    with pytest.raises(RuntimeError):
        balanced_bands([1.0, 1.0, 9.0], 3, H=72)
    m = tile_owner_mask(80, 96, *band_code(1, 3, 1, 2))
    assert m[32:].all() and not m[:32].any()
