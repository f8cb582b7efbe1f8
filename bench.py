#!/usr/bin/env python
"""bench.py — GS-ICP-SLAM hot path on MI355X.

One "step" = one SLAM frame's worth of the hot path on synthetic Replica-shaped input:
  tracker : pygicp.FastGICP  set_input_source + set_source_filter + align + get_source_correspondence
            on S-pair (8 280 points/frame, max_correspondence_distance 0.02)          [REF mp_Tracker.py:191-231]
  mapper  : one full optimisation iteration — activations, GaussianRasterizer forward, the mapping loss
            0.8 L1 + 0.2 (1-SSIM) + 0.1 L1(depth/10) (fused HIP kernel), backward, Adam step over the parameter groups
            (fused HIP kernel), zero_grad; activations fused; P = 300 000 surfels, 1200x680                 [REF mp_Mapper.py:219-248]
With N > 1 GPUs the mapper's tiles are sharded across ranks (strong scaling; gs_icp_slam_amd/sharded.py) and the
tracker runs as a replica on every rank (it does not shard — DESIGN.md).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, algorithmic bytes
over HIP-event kernel time) and `cpu_baseline` (the OpenMP GICP oracle timed on this host's cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--gaussians", type=int, default=300_000)
    ap.add_argument("--res", choices=["replica", "tum"], default="replica")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--serial", action="store_true", help="run tracker and mapper back-to-back on one thread (default: concurrently, as the\n                    reference runs them in two processes)")
    ap.add_argument("--no-graph", action="store_true", help="drive the mapper iteration eagerly from Python instead of replaying the captured\n                    HIP graph (N > 1 always runs eagerly)")
    ap.add_argument("--only", choices=["tracker", "mapper"], default=None, help="diagnostics: run only one half of the step (the JSON line is then\n                    NOT the contract metric)")
    ap.add_argument("--pyprofile", default=None, help="write a cProfile of the timed region to this file (diagnostics)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from gs_icp_slam_amd import _lib, synth
    from gs_icp_slam_amd.sharded import ShardedGaussianRasterizer
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    import pygicp

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # GSICP_BENCH_BACKEND=gloo lets the N > 1 code path be exercised on a box with fewer GPUs than ranks (ranks then share devices;
    # a functional check only — RCCL refuses two ranks on one device).  The driver's runs use the default: nccl (= RCCL), one GPU per rank.
    backend = os.environ.get("GSICP_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    cfg = synth.REPLICA if args.res == "replica" else synth.TUM
    W, H = cfg["W"], cfg["H"]
    P = args.gaussians
    # ---------------- mapper inputs (S-map) ----------------
    cam = synth.make_camera(W, H, cfg["fx"], cfg["fy"], synth.DEFAULT_POSE_A)
    g = synth.s_map(P, seed=2)
    # raw (pre-activation) parameters as GaussianModel keeps them [REF scene/gaussian_model.py:105-125]: log-scales, logit
    # opacities, un-normalised quaternions; the activations below are the reference's torch ops and are part of the step
    params = {"means3D": torch.from_numpy(g["means3D"]), "scales": torch.log(torch.from_numpy(g["scales"])),
              "rotations": torch.from_numpy(g["rotations"]),
              "opacities": torch.logit(torch.from_numpy(g["opacities"]).clamp(1e-4, 1 - 1e-4)), "shs": torch.from_numpy(g["shs"])}
    params = {k: v.to(dev).contiguous().requires_grad_(True) for k, v in params.items()}
    from gs_icp_slam_amd.loss import mapper_loss_and_grads
    from gs_icp_slam_amd.optim import FusedAdam
    lrs = {"means3D": 1.6e-6 * 2.5, "shs": 2.5e-3, "opacities": 0.05, "scales": 5e-3, "rotations": 1e-3}   # REF arguments/__init__.py:141-148
    use_graph = (world == 1) and not args.no_graph
    optimizer = FusedAdam([{"params": [params[k]], "lr": lr} for k, lr in lrs.items()], lr=0.0, eps=1e-15, capturable=use_graph)
    rs = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(3, device=dev), scale_modifier=1.0,
        viewmatrix=torch.from_numpy(cam["viewmatrix"]).to(dev), projmatrix=torch.from_numpy(cam["projmatrix"]).to(dev), sh_degree=0,
        campos=torch.from_numpy(cam["campos"]).to(dev), prefiltered=False, debug=False)
    rast = ShardedGaussianRasterizer(rs)
    with torch.no_grad():   # target images: render of a perturbed copy, so gradients are non-zero
        g2 = synth.s_map(P, seed=2, perturb_seed=3)
        t2 = {k: torch.from_numpy(g2[k]).to(dev) for k in ("means3D", "scales", "rotations", "opacities", "shs")}
        gt_depth, gt_color, _, _ = rast(means3D=t2["means3D"], means2D=torch.zeros_like(t2["means3D"]), shs=t2["shs"],
                                        opacities=t2["opacities"], scales=t2["scales"], rotations=t2["rotations"])
        gt_depth, gt_color = gt_depth.clone(), gt_color.clone()
        del t2

    # ---------------- tracker inputs (S-pair) ----------------
    sp = synth.s_pair(cfg, noise=(args.res == "tum"))
    pw = sp["points_a"].astype(np.float64) @ sp["pose_a"][:3, :3].T + sp["pose_a"][:3, 3]

    def filt(n, tr):
        f = np.zeros(n, np.int32)
        f[tr] = np.arange(1, len(tr) + 1)
        return f
    f_src = filt(len(sp["points_b"]), sp["trackable_b"])

    def setup_tracker(reg):
        reg.set_max_correspondence_distance(cfg["max_corr"])
        reg.set_max_knn_distance(99999.0)
        reg.set_input_target(pw)
        reg.set_target_filter(len(sp["trackable_a"]), filt(len(pw), sp["trackable_a"]))
        reg.calculate_target_covariance_with_filter()

    reg = pygicp.FastGICP()
    setup_tracker(reg)

    def tracker_step(r):
        r.set_input_source(sp["points_b"])
        r.set_source_filter(len(sp["trackable_b"]), f_src)
        T = r.align(sp["pose_a"])
        idx, d2 = r.get_source_correspondence()
        return T, idx, d2

    last = {}

    # The reference runs the tracker and the mapper as two concurrent processes on one GPU [REF gs_icp_slam.py:121-131].
    # Here: the tracker frame runs on a worker thread (its C calls release the GIL and use the tracker's own HIP stream)
    # while the main thread drives the mapper iteration on torch's stream; a step ends when both are done.
    import queue
    import threading
    jobs, done = queue.Queue(), queue.Queue()

    def tracker_worker():
        while True:
            if jobs.get() is None:
                return
            done.put(tracker_step(reg))
    worker = None
    if not args.serial:
        worker = threading.Thread(target=tracker_worker, daemon=True)
        worker.start()

    from gs_icp_slam_amd.activations import activate

    def activated():   # GaussianModel.get_opacity / get_scaling / get_rotation [REF scene/gaussian_model.py:105-125], one fused launch
        o, s_, q = activate(params["opacities"], params["scales"], params["rotations"])
        return dict(means3D=params["means3D"], shs=params["shs"], opacities=o, scales=s_, rotations=q)

    def mapper_iteration():
        """One iteration of Mapper.mapping [REF mp_Mapper.py:219-248]: render_3 -> loss -> backward -> Adam step -> zero_grad."""
        a = activated()
        means2D = torch.zeros_like(a["means3D"], requires_grad=True)
        depth, color, radii, used = rast(means3D=a["means3D"], means2D=means2D, shs=a["shs"], opacities=a["opacities"],
                                         scales=a["scales"], rotations=a["rotations"])
        # the fused loss hands dL/dimage and dL/ddepth straight to autograd (no loss node, no ones_like / multiply launches)
        parts, g_color, g_depth = mapper_loss_and_grads(color, depth, gt_color, gt_depth, lambda_dssim=0.2)
        torch.autograd.backward((color, depth), (g_color, g_depth))
        optimizer.step()
        optimizer.zero_grad(set_to_none=True)
        return parts[0], radii

    # Duplicate-list capacity for the sync-free forward: 1.5x the count of one probe forward (per rank: each rank bins its own tiles).
    def probe_capacity():
        cap = max(8 * P // world, 1 << 20)
        from gs_icp_slam_amd.rasterizer import GaussianRasterizer as _PlainRasterizer
        while True:   # plain rasteriser on this rank's tiles: no collective inside a loop whose trip count may differ between ranks
            probe = _PlainRasterizer(rs._replace(capacity=cap, tile_mod=world, tile_rem=rank))
            with torch.no_grad():
                a0 = activated()
                probe(means3D=a0["means3D"], means2D=torch.zeros_like(a0["means3D"]), shs=a0["shs"], opacities=a0["opacities"],
                      scales=a0["scales"], rotations=a0["rotations"])
            r = int(probe.num_rendered.item())
            if r <= cap:
                return int(1.5 * r) + 4096
            cap *= 2
    capacity = probe_capacity()
    rast = ShardedGaussianRasterizer(rs._replace(capacity=capacity))   # eager iterations also run without the forward's host sync

    mg = None
    if use_graph:
        # Single GPU: the whole iteration is one hipGraph launch (gs_icp_slam_amd/graph.py); the keyframe (camera + targets) is
        # re-selected before every replay, as the reference's mapper does [REF mp_Mapper.py:205-217].
        from gs_icp_slam_amd.graph import MapperIterationGraph
        mg = MapperIterationGraph(params, optimizer, H, W, cam["tanfovx"], cam["tanfovy"], sh_degree=0, capacity=capacity,
                                  lambda_dssim=0.2, warmup=2)
        mg.set_view(rs.viewmatrix, rs.projmatrix, rs.campos, gt_color, gt_depth)
        mg.capture()
        eager_iteration = mapper_iteration

        def mapper_iteration():   # noqa: F811
            mg.set_view(rs.viewmatrix, rs.projmatrix, rs.campos, gt_color, gt_depth)
            return mg.step(), mg.radii

    def step():
        if args.only == "tracker":
            T, idx, d2 = tracker_step(reg)
            last.update(T=T)
            return
        if args.only == "mapper":
            loss, radii = mapper_iteration()
            last.update(loss=loss, radii=radii)
            return
        if worker is not None:
            jobs.put(1)
            loss, radii = mapper_iteration()
            T, idx, d2 = done.get()
        else:
            T, idx, d2 = tracker_step(reg)
            loss, radii = mapper_iteration()
        last.update(T=T, loss=loss, radii=radii)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    _lib.profile_enable(True)
    _lib.profile_read()
    barrier()
    prof_py = None
    if args.pyprofile:
        import cProfile
        prof_py = cProfile.Profile()
        prof_py.enable()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if prof_py is not None:
        prof_py.disable()
        import pstats
        with open(args.pyprofile, "w") as fh:
            pstats.Stats(prof_py, stream=fh).sort_stats("cumulative").print_stats(45)
    prof = _lib.profile_read()
    n_prof = {k: args.steps for k in prof}
    if mg is None and int(rast.inner.num_rendered.item()) > capacity:
        raise RuntimeError("duplicate-list capacity overflowed during the timed region")
    if mg is not None:
        # kernels inside a replayed graph carry no HIP events: time the SAME kernels on the same inputs in eager iterations
        # right after the timed region (rocprofv3's kernel trace of this command sees both and agrees — profiles/README.md)
        if mg.overflowed():
            raise RuntimeError(f"duplicate-list capacity overflowed during the timed region: R = {int(mg.num_rendered.item())} > {mg.capacity}")
        n_e = max(5, min(args.steps, 20))
        eager_iteration()
        torch.cuda.synchronize()
        _lib.profile_read()
        for _ in range(n_e):
            eager_iteration()
        torch.cuda.synchronize()
        for k, v in _lib.profile_read().items():
            if v[1] > 0 and not k.startswith("gicp"):
                prof[k] = v
                n_prof[k] = n_e
    _lib.profile_enable(False)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---------------- roofline of the dominant kernel ----------------
    per_launch_us = {k: (1e3 * ms / n_prof[k]) for k, (ms, c) in prof.items() if c > 0}   # us per step (a stage may be several launches)
    raster_stages = ["preprocess", "tile_scan_lpt", "scatter", "tile_sort", "blend_forward", "blend_backward", "entry_grad_sum",
                     "preprocess_backward"]
    dominant = max((k for k in raster_stages if k in per_launch_us), key=lambda k: per_launch_us[k])
    # D (duplicates) and P_vis from one extra forward
    a_ = activated()
    means2D = torch.zeros_like(a_["means3D"], requires_grad=True)
    depth, color, radii, used = rast(means3D=a_["means3D"], means2D=means2D, shs=a_["shs"], opacities=a_["opacities"],
                                     scales=a_["scales"], rotations=a_["rotations"])
    D_local = int(rast.inner.num_rendered.item())
    P_vis = int((radii > 0).sum())
    T_tiles = ((W + 15) // 16) * ((H + 15) // 16)
    # Algorithmic bytes per launch: list word 4 B per (strip, entry) visit = 16 B per duplicate,
    # 48 B record per duplicate (an upper bound: only staged records are gathered), per-pixel state, gradient slots.
    alg_bytes = {"blend_forward": 64.0 * D_local + 24.0 * W * H / world + 8.0 * T_tiles / world,
                 "blend_backward": 64.0 * D_local + 40.0 * W * H / world + 48.0 * D_local,
                 "preprocess": 128.0 * P, "preprocess_backward": 184.0 * P + 48.0 * D_local, "tile_sort": 24.0 * D_local,
                 "scatter": 16.0 * D_local + 56.0 * P_vis, "tile_scan_lpt": 16.0 * T_tiles, "entry_grad_sum": 240.0 * D_local}
    ach = alg_bytes[dominant] / (per_launch_us[dominant] * 1e-6) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")   # FETCH_SIZE / WRITE_SIZE passes of this command (profiles/README.md)
    if os.path.exists(tpath) and world == 1 and args.res == "replica" and P == 300_000:
        tj = json.load(open(tpath)).get(dominant)
        if tj:
            traffic = int(tj["fetch_bytes"] + tj["write_bytes"])
    # the blend kernels gather 48-byte records that sit in the 256 MiB Infinity Cache: they are instruction-issue bound, not HBM bound.
    # When the SQ counter pass of this command is on disk (profiles/), state the VALU-issue floor next to the HBM fraction.
    note = "working set (~60 MB) sits in the 256 MiB Infinity Cache; the kernel is VALU-issue bound, the HBM fraction is a formality"
    kernel_of = {"blend_backward": "blend_backward_strip_kernel", "blend_forward": "blend_forward_strip_kernel"}
    sqpath = os.path.join(ROOT, "profiles", "r01_rocprofv3_pmc_sq.csv")
    if os.path.exists(sqpath) and dominant in kernel_of and world == 1 and args.res == "replica" and P == 300_000:
        import csv
        for row in csv.DictReader(open(sqpath)):
            if row["kernel"] == kernel_of[dominant] and float(row.get("SQ_INSTS_VALU", 0) or 0) > 0:
                valu = float(row["SQ_INSTS_VALU"])
                floor_us = valu * 4.0 / 1024.0 / 2.4e3      # wave64 VALU op = 4 cycles on a 16-lane SIMD; 1024 SIMDs; 2.4 GHz
                note += (f"; SQ_INSTS_VALU = {valu / 1e6:.1f} M wave-instructions x 4 cycles / 1024 SIMDs / 2.4 GHz = {floor_us:.0f} us "
                         f"issue floor vs kernel_us (profiles/r01_rocprofv3_pmc_sq.csv)")
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic, "kernel_us": round(per_launch_us[dominant], 2),
                "algorithmic_bytes": int(alg_bytes[dominant]), "note": note}

    # ---------------- CPU baseline: OpenMP GICP oracle (port), rank 0 only ----------------
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        import oracle
        oreg = oracle.OracleGICP()
        setup_tracker(oreg)
        tracker_step(oreg)  # warm-up (thread pool, first touch)
        # the problem is small (8 k points): more OpenMP threads than it can feed only add overhead, so pick the
        # fastest thread count on this host first and report THAT as the baseline
        ncpu = os.cpu_count() or 1
        best_thr, best_rate = ncpu, 0.0
        for thr in sorted({t for t in (4, 8, 16, 32, 64, ncpu) if t <= ncpu}):
            oreg.set_num_threads(thr)
            tracker_step(oreg)
            n, t0c = 0, time.perf_counter()
            while time.perf_counter() - t0c < 0.6:
                tracker_step(oreg)
                n += 1
            rate = n / (time.perf_counter() - t0c)
            if rate > best_rate:
                best_thr, best_rate = thr, rate
        oreg.set_num_threads(best_thr)
        n, t_cpu0 = 0, time.perf_counter()
        while time.perf_counter() - t_cpu0 < args.cpu_seconds:
            To, _, _ = tracker_step(oreg)
            n += 1
        t_cpu = time.perf_counter() - t_cpu0
        cpu = {"value": round(n / t_cpu, 2), "unit": "tracker frames/s (GICP align only; the reference has no CPU rasteriser)",
               "cores": best_thr, "host_cores": ncpu, "kind": "port",
               "sample": f"{n} x (set_input_source + align + get_source_correspondence) on S-pair {args.res}, {len(sp['points_b'])} points, "
                         f"{t_cpu:.1f} s wall, OpenMP kd-tree oracle at its fastest thread count",
               "pose_agrees_with_gpu": bool(np.allclose(To, last["T"], atol=1e-5))}

    if rank == 0:
        ms = 1e3 * dt / args.steps
        stage_us = {k: round(v, 2) for k, v in per_launch_us.items()}
        out = {
            "metric": ("SLAM hot-path FPS (GICP tracker frame + one full mapper iteration: render, loss, backward, Adam), Replica room0-shaped synthetic"
                       if args.only is None else f"DIAGNOSTIC: {args.only} half only"),
            "value": round(args.steps / dt, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2] shape: S-pair {args.res} tracker ({len(sp['points_b'])} pts, gate {cfg['max_corr']} m) + "
                                   f"S-map mapper iteration (P={P}, {W}x{H}, sh_degree 0)", "gaussians": P, "width": W, "height": H,
                       "duplicates_per_rank": D_local, "visible_gaussians": P_vis,
                       "tracker_mapper_overlap": not args.serial,
                       "mapper_iteration": "one hipGraph replay per iteration" if mg is not None else "eager launches from Python",
                       "parallelism": "single GPU" if world == 1 else f"mapper tiles sharded x{world} (RCCL all-reduce image + grads), tracker replicated"},
            "render_bwd_ms_per_iter": round(sum(per_launch_us.get(k, 0.0) for k in raster_stages) / 1e3, 4),
            "tracker_align_kernel_us": stage_us.get("gicp_align"),
            "stage_us_per_step": stage_us,
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if worker is not None:
        jobs.put(None)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
