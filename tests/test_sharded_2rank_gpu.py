"""GPU, TWO processes sharing the one device over gloo: the tile-sharded mapper iteration with the REAL HIP kernels on both ranks — rasteriser
on the rank's own super-tiles, tile movers + image all-gather, the fused loss on the rank's own 32x32 blocks, gradient exchange, fused Adam —
must walk the same optimiser trajectory as the single-process iteration (up to fp32 summation order in the gradient all-reduce), and the
summed loss shares must equal the single-process loss.  (tests/test_sharded_cpu.py checks the collective logic with an oracle stand-in;
tests/test_sharded_gpu.py the movers with emulated ranks and the captured iteration on a 1-rank RCCL group.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gs_icp_slam_amd import synth
from tests import util

pytestmark = pytest.mark.gpu
P, W, H, STEPS = 20000, 336, 208, 4
LRS = {"means3D": 4e-6, "shs": 2.5e-3, "opacities": 0.05, "scales": 5e-3, "rotations": 1e-3}


def _setup(dev):
    from diff_gaussian_rasterization import GaussianRasterizer
    cam = synth.make_camera(W, H, 150.0, 150.0, synth.DEFAULT_POSE_A)
    g = synth.s_map(P, seed=5)
    raw = {"means3D": torch.from_numpy(g["means3D"]), "scales": torch.log(torch.from_numpy(g["scales"])), "rotations": torch.from_numpy(g["rotations"]),
           "opacities": torch.logit(torch.from_numpy(g["opacities"]).clamp(1e-4, 1 - 1e-4)), "shs": torch.from_numpy(g["shs"])}
    params = {k: v.to(dev).contiguous().requires_grad_(True) for k, v in raw.items()}
    rs = util.make_settings(cam, [0.0, 0.0, 0.0])
    t2 = util.torch_inputs(synth.s_map(P, seed=5, perturb_seed=7))
    with torch.no_grad():
        gt_d, gt_c, _, _ = GaussianRasterizer(rs)(means3D=t2["means3D"], means2D=torch.zeros_like(t2["means3D"]), shs=t2["shs"],
                                                  opacities=t2["opacities"], scales=t2["scales"], rotations=t2["rotations"])
    return params, rs, gt_c.clone(), gt_d.clone()


def _iterate(params, rast, gt_c, gt_d, steps):
    from gs_icp_slam_amd.activations import activate
    from gs_icp_slam_amd.loss import mapper_loss_and_grads
    from gs_icp_slam_amd.optim import FusedAdam
    opt = FusedAdam([{"params": [params[k]], "lr": lr} for k, lr in LRS.items()], lr=0.0, eps=1e-15)
    losses = []
    for _ in range(steps):
        o, s_, q = activate(params["opacities"], params["scales"], params["rotations"])
        m2 = torch.zeros_like(params["means3D"], requires_grad=True)
        depth, color, radii, used = rast(means3D=params["means3D"], means2D=m2, shs=params["shs"], opacities=o, scales=s_, rotations=q)
        shard = rast.loss_shard() if hasattr(rast, "loss_shard") else (1, 0)
        parts, g_c, g_d = mapper_loss_and_grads(color, depth, gt_c, gt_d, lambda_dssim=0.2, tile_mod=shard[0], tile_rem=shard[1])
        if shard[0] > 1:
            rast.attach_loss_share(parts)
        torch.autograd.backward((color, depth), (g_c, g_d))
        if shard[0] > 1:
            parts = rast.summed_loss()
        losses.append(parts.detach().cpu().numpy().copy())
        opt.step()
        opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    return np.stack(losses)


def _worker(rank, world, port, q, bands=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gs_icp_slam_amd.sharded import ShardedGaussianRasterizer
    dev = torch.device("cuda", 0)
    params, rs, gt_c, gt_d = _setup(dev)
    rast = ShardedGaussianRasterizer(rs, bands=bands)
    losses = _iterate(params, rast, gt_c, gt_d, STEPS)
    q.put((rank, losses, {k: v.detach().cpu().numpy() for k, v in params.items()}, rast.holder.last_image_bytes))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bands", [None, "equal", [0, 2, 7]])
def test_two_ranks_on_one_gpu_walk_the_single_gpu_trajectory(hip_lib, bands):
    """bands=None: round-robin super-tiles + all-gather of the image (rounds 3-4).  bands="equal" / a list (round 5): contiguous bands of super-tile rows
    + a halo exchange of 2 x 10 pixel rows per rank — the same trajectory with 1/27 of the image bytes exchanged at this size."""
    from diff_gaussian_rasterization import GaussianRasterizer
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, bands)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=420) for _ in procs], key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    dev = torch.device("cuda", 0)
    params, rs, gt_c, gt_d = _setup(dev)
    ref_losses = _iterate(params, GaussianRasterizer(rs), gt_c, gt_d, STEPS)
    for rank, losses, pr, image_bytes in outs:
        if bands is None:
            assert image_bytes >= 4 * W * H * 4 // 2                     # this rank's half of the image (padded to whole super-tile slots)
        else:
            assert image_bytes == 2 * 4 * 10 * W * 4                     # 2 x BAND_HALO rows x W x {r, g, b, depth}
        np.testing.assert_allclose(losses, ref_losses, rtol=2e-5, atol=1e-7, err_msg=f"rank {rank}: summed loss shares differ from the single-GPU loss")
        for k, v in params.items():
            ref = v.detach().cpu().numpy()
            assert np.abs(pr[k] - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1.0) + 1e-7, f"rank {rank} {k}: {np.abs(pr[k] - ref).max():.3e}"
    for k in outs[0][2]:      # the replicated parameters stay replicated
        assert np.array_equal(outs[0][2][k], outs[1][2][k]), k
    assert ref_losses[-1, 0] < ref_losses[0, 0]
