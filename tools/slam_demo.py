#!/usr/bin/env python
"""The device-resident SLAM loop on synthetic frames, strung together the way the reference's two processes use the hot path
[REF mp_Tracker.py:113-330; mp_Mapper.py:140-250] — with ONE captured mapper graph for the whole run:

  frame 0   : front-end kernel -> world points -> tracker target (kNN covariances) -> first Gaussians into GaussianStore(stable=True)
              -> MapperIterationGraph captured ONCE over the store's full-capacity buffers (live count on the device)
  frame k   : front-end kernel -> set_input_source / set_source_trackable (device tensors) -> align -> correspondences -> overlap
              statistics; the reference's keyframe rules [REF mp_Tracker.py:233-249]:
                 tracking keyframe (overlap < keyframe_th, or last frame): not-yet-mapped points become Gaussians, the map goes back to the
                                   tracker on the device (set_target_from_gaussians)
                 mapping keyframe (every keyframe_freq frames after a tracking keyframe): all points of the frame become Gaussians
              keyframe ingestion = store.append (rows written in place, count bumped on the device): NO re-capture
  mapper    : `--iters` replays of the one graph per frame on a keyframe chosen like the reference does (new keyframe first, then random)
              [REF mp_Mapper.py:197-206], prune_large_and_transparent every `--prune-every` iterations [REF mp_Mapper.py:244-245]

Prints per-frame pose errors against the known trajectory, keyframe events, the loss trend and the number of graph captures (must be 1).
Not a benchmark of the frame maker (a CPU ray-caster run in worker processes before the loop starts); the loop itself is timed."""
import argparse
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _frame(job):
    from gs_icp_slam_amd import synth
    cfg, pose = job
    return synth.render_frame(cfg, pose)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("frames", nargs="?", type=int, default=56)
    ap.add_argument("--iters", type=int, default=6, help="mapper iterations (graph replays) per tracked frame")
    ap.add_argument("--prune-every", type=int, default=200)
    ap.add_argument("--keyframe-th", type=float, default=0.7)
    ap.add_argument("--keyframe-freq", type=int, default=10)
    ap.add_argument("--capacity", type=int, default=400_000, help="map capacity in Gaussians")
    ap.add_argument("--list-capacity", type=int, default=1 << 22, help="duplicate-list capacity of the sync-free rasteriser")
    ap.add_argument("--eval-every", type=int, default=0, help="map-quality checkpoint every this many mapper iterations: PSNR / SSIM of the map "
                    "rendered at the ESTIMATED poses of the frames tracked so far, computed the way the reference's end-of-run pass does "
                    "[REF mp_Mapper.py:335-420]")
    ap.add_argument("--eval-stride", type=int, default=8, help="checkpoints evaluate every this-many-th tracked frame (the final one: every frame)")
    ap.add_argument("--post-iters", type=int, default=0, help="keep mapping on random keyframes for this many iterations after the last frame (shows the plateau)")
    ap.add_argument("--scale-semantics", choices=["stddev", "variance"], default="stddev")
    ap.add_argument("--json", default=None, help="write the run summary and the quality curve here")
    ap.add_argument("--no-asserts", action="store_true")
    ap.add_argument("--save-map", default=None, help="write the trained map (ACTIVATED parameters as the rasteriser takes them) and the keyframe poses to this .npz "
                    "(tests/test_raster_gpu.py::test_trained_map_* compare the rasteriser with the oracle on it)")
    ap.add_argument("--dump", default=None, help="directory for side-by-side (ground truth | render | accumulated alpha) PNGs of a few frames at the final checkpoint")
    args = ap.parse_args()

    import torch
    import pygicp
    from gs_icp_slam_amd import synth
    from gs_icp_slam_amd.frontend import DepthFrontEnd, overlap_statistics, quaternion_multiply, rotation_to_quaternion_xyzw
    from gs_icp_slam_amd.gaussian_store import GaussianStore, rows_from_gicp
    from gs_icp_slam_amd.graph import MapperIterationGraph
    from gs_icp_slam_amd.optim import FusedAdam

    cfg = synth.REPLICA
    H, W = cfg["H"], cfg["W"]
    LRS = {"xyz": 1.6e-6 * 2.5, "f_dc": 2.5e-3, "f_rest": 1.25e-4, "opacity": 0.05, "scaling": 5e-3, "rotation": 1e-3}
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)

    poses = synth.trajectory(args.frames)
    t_gen = time.perf_counter()
    from concurrent.futures import ProcessPoolExecutor
    with ProcessPoolExecutor(max_workers=max(1, min(16, (os.cpu_count() or 2) // 2))) as ex:
        frames = list(ex.map(_frame, [(cfg, p) for p in poses]))
    t_gen = time.perf_counter() - t_gen

    def pose_err(T, gt):
        dR = np.asarray(T, np.float64)[:3, :3] @ gt[:3, :3].T
        return np.degrees(np.linalg.norm(dR - np.eye(3)) / math.sqrt(2.0)), 1e3 * np.linalg.norm(np.asarray(T, np.float64)[:3, 3] - gt[:3, 3])

    def upload(k):
        rgb, d16 = frames[k]
        return torch.from_numpy(d16.view(np.int16)).to(dev), torch.from_numpy(rgb).to(dev)

    class Keyframe:
        def __init__(self, pose, rgb_dev, d16_dev):
            cam = synth.make_camera(W, H, cfg["fx"], cfg["fy"], pose)
            self.pose = np.asarray(pose, np.float64).copy()
            self.view = torch.from_numpy(cam["viewmatrix"]).to(dev)
            self.proj = torch.from_numpy(cam["projmatrix"]).to(dev)
            self.campos = torch.from_numpy(cam["campos"]).to(dev)
            self.tan = (cam["tanfovx"], cam["tanfovy"])
            self.gt_img = (rgb_dev.permute(2, 0, 1).float() / 255.0).contiguous()
            self.gt_dep = (d16_dev.view(torch.int16).to(torch.int32).bitwise_and(0xFFFF).float() / np.float32(cfg["depth_scale"]))[None].contiguous()

    t_loop = time.perf_counter()
    fe = DepthFrontEnd(H, W, cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], cfg["stride"], cfg["depth_scale"], cfg["depth_trunc"])
    reg = pygicp.FastGICP()
    reg.set_max_correspondence_distance(cfg["max_corr"])
    reg.set_max_knn_distance(99999.0)
    if args.scale_semantics != "stddev":
        reg.set_scale_semantics(args.scale_semantics)

    # ------------------------------------------------------------------------------------------ frame 0
    d16_dev, rgb_dev = upload(0)
    pc = fe.make_pointcloud(d16_dev, rgb_dev)
    pw = DepthFrontEnd.to_world(pc.points, poses[0]).contiguous()
    reg.set_input_target(pw)
    trk = pc.trackable_idx.cpu().numpy()
    filt = np.zeros(pw.shape[0], np.int32)
    filt[trk] = np.arange(1, len(trk) + 1)
    reg.set_target_filter(len(trk), filt)
    reg.calculate_target_covariance_with_filter()
    rots = torch.from_numpy(np.reshape(np.asarray(reg.get_target_rotationsq(), np.float32), (-1, 4)).copy()).to(dev)
    scales = torch.from_numpy(np.reshape(np.asarray(reg.get_target_scales(), np.float32), (-1, 3)).copy()).to(dev)
    store = GaussianStore(args.capacity, n_rest=0, stable=True)
    rows, tmask = rows_from_gicp(pw, pc.colors, rots, scales.clamp_min(1e-4), pc.z_values, pc.trackable_idx.long())
    store.append(rows, tmask)
    opt = store.attach(FusedAdam, LRS, lr=0.0, eps=1e-15, capturable=True)
    keyframes = [Keyframe(poses[0], rgb_dev, d16_dev)]
    new_keyframes = [0]
    params = {"means3D": store.params["xyz"], "shs": store.params["f_dc"], "opacities": store.params["opacity"],
              "scales": store.params["scaling"], "rotations": store.params["rotation"]}
    mg = MapperIterationGraph(params, opt, H, W, keyframes[0].tan[0], keyframes[0].tan[1], sh_degree=0, capacity=args.list_capacity, warmup=1,
                              live_count=store.live_count)
    captures = 0
    kf = keyframes[0]
    mg.set_view(kf.view, kf.proj, kf.campos, kf.gt_img, kf.gt_dep)
    mg.capture()
    captures += 1
    graph_obj = mg.graph

    train_iter, prunes, losses = 0, 0, []

    def map_some(n):
        nonlocal train_iter, prunes
        for _ in range(n):
            idx = new_keyframes.pop(0) if new_keyframes else int(rng.integers(len(keyframes)))
            kf = keyframes[idx]
            mg.set_view(kf.view, kf.proj, kf.campos, kf.gt_img, kf.gt_dep)
            loss = mg.step()
            train_iter += 1
            if train_iter % 25 == 0:
                losses.append(float(loss))
            if train_iter % args.prune_every == 0:   # prune_large_and_transparent(0.005, 2.5) [REF mp_Mapper.py:244-245; scene/gaussian_model.py:580-592]
                with torch.no_grad():
                    remove = (torch.sigmoid(store.live("opacity")) < 0.005).squeeze(-1) | (torch.exp(store.live("scaling")).max(dim=1).values > 0.1 * 2.5)
                before = store.n
                store.prune(remove)
                prunes += 1
                print(f"    prune at iteration {train_iter}: {before} -> {store.n} Gaussians (same buffers, same graph)")

    # ---- map quality the way the reference's end-of-run pass measures it [REF mp_Mapper.py:335-420]: render the map at the ESTIMATED pose of a
    # frame, clamp to [0,1], mask both images with gt depth > 0, PSNR = -10 log10(mean squared error), SSIM with the reference's window
    from gs_icp_slam_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from gs_icp_slam_amd.loss import mapper_loss_and_grads
    est_poses = {0: poses[0].copy()}
    curve = []

    def evaluate(frame_ids, dump_ids=()):
        ps, ss, l1d = [], [], []
        with torch.no_grad():
            a = dict(means3D=store.live("xyz"), shs=store.live("f_dc"), opacities=torch.sigmoid(store.live("opacity")),
                     scales=torch.exp(store.live("scaling")), rotations=torch.nn.functional.normalize(store.live("rotation")))
            for k in frame_ids:
                cam = synth.make_camera(W, H, cfg["fx"], cfg["fy"], est_poses[k])
                rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                                                   bg=torch.zeros(3, device=dev), scale_modifier=1.0,
                                                   viewmatrix=torch.from_numpy(cam["viewmatrix"]).to(dev),
                                                   projmatrix=torch.from_numpy(cam["projmatrix"]).to(dev), sh_degree=0,
                                                   campos=torch.from_numpy(cam["campos"]).to(dev), prefiltered=False, debug=False)
                depth, color, _r, _u = GaussianRasterizer(rs)(means3D=a["means3D"], means2D=torch.zeros_like(a["means3D"]), shs=a["shs"],
                                                             opacities=a["opacities"], scales=a["scales"], rotations=a["rotations"])
                d16_k, rgb_k = upload(k)
                gt = (rgb_k.permute(2, 0, 1).float() / 255.0)
                gtd = (d16_k.view(torch.int16).to(torch.int32).bitwise_and(0xFFFF).float() / np.float32(cfg["depth_scale"]))[None]
                mask = gtd > 0
                ours, gt = torch.clamp(color, 0.0, 1.0) * mask, gt * mask
                ps.append(float(-10.0 * torch.log10(torch.mean((gt - ours) ** 2))))
                parts, _gc, _gd = mapper_loss_and_grads(ours.contiguous(), depth.contiguous(), gt.contiguous(), gtd.contiguous())
                ss.append(float(parts[2]))
                l1d.append(float(torch.abs(depth - gtd)[mask].mean()))
                if dump_ids and k in dump_ids:
                    from PIL import Image
                    _d2, acc, _r2, _u2 = GaussianRasterizer(rs)(means3D=a["means3D"], means2D=torch.zeros_like(a["means3D"]),
                                                               colors_precomp=torch.ones_like(a["means3D"]), opacities=a["opacities"],
                                                               scales=a["scales"], rotations=a["rotations"])
                    row = torch.cat([gt, ours, acc.clamp(0, 1)], dim=2)[:, ::2, ::2]
                    os.makedirs(args.dump, exist_ok=True)
                    Image.fromarray((row.permute(1, 2, 0) * 255).byte().cpu().numpy(), "RGB").save(os.path.join(args.dump, f"frame{k:04d}_it{train_iter}.png"))
        return dict(psnr=float(np.mean(ps)), psnr_min=float(np.min(ps)), ssim=float(np.mean(ss)), depth_l1_m=float(np.mean(l1d)), frames=len(ps))

    def checkpoint(final=False):
        ids = sorted(est_poses)
        if not final:
            ids = ids[:: max(1, args.eval_stride)]
        q = evaluate(ids, dump_ids=(ids[0], ids[len(ids) // 2], ids[-1]) if (final and args.dump) else ())
        if final:
            with torch.no_grad():
                op, sc = torch.sigmoid(store.live("opacity")).reshape(-1), torch.exp(store.live("scaling"))
                qs = torch.tensor([0.05, 0.5, 0.95], device=dev)
                q["opacity_quantiles_5_50_95"] = [float(v) for v in torch.quantile(op[:: max(1, op.numel() // 200000)], qs)]
                q["max_scale_m_quantiles_5_50_95"] = [float(v) for v in torch.quantile(sc.max(dim=1).values[:: max(1, op.numel() // 200000)], qs)]
                q["min_scale_m_quantiles_5_50_95"] = [float(v) for v in torch.quantile(sc.min(dim=1).values[:: max(1, op.numel() // 200000)], qs)]
        q.update(iteration=train_iter, frames_tracked=len(est_poses), gaussians=store.n, keyframes=len(keyframes), final=final)
        curve.append(q)
        print(f"    quality at iteration {train_iter} ({len(est_poses)} frames tracked, {store.n} Gaussians): PSNR {q['psnr']:.2f} dB "
              f"(worst frame {q['psnr_min']:.2f}), SSIM {q['ssim']:.3f}, depth L1 {1e3 * q['depth_l1_m']:.1f} mm over {q['frames']} frames")

    _map_some_plain = map_some

    def map_some(n):   # noqa: F811 — the same loop, cut at the checkpoints
        if args.eval_every <= 0:
            return _map_some_plain(n)
        while n > 0:
            m = min(n, args.eval_every - train_iter % args.eval_every)
            _map_some_plain(m)
            n -= m
            if train_iter % args.eval_every == 0:
                checkpoint()

    if args.eval_every > 0:
        checkpoint()
    map_some(20)
    print(f"frame 0: {pw.shape[0]} points -> {store.n} Gaussians; first losses {losses[:1]}")

    # ------------------------------------------------------------------------------------------ tracking
    pose_est = poses[0].copy()
    worst = (0.0, 0.0)
    n_track_kf = n_map_kf = 0
    since_tracking_kf = 0
    for k in range(1, args.frames):
        d16_dev, rgb_dev = upload(k)
        pc = fe.make_pointcloud(d16_dev, rgb_dev)
        reg.set_input_source(pc.points)
        reg.set_source_trackable(pc.trackable_idx)
        T = reg.align(pose_est)
        idx, d2 = reg.get_source_correspondence()
        ang, mm = pose_err(T, poses[k])
        worst = (max(worst[0], ang), max(worst[1], mm))
        ratio, new_idx = overlap_statistics(torch.from_numpy(d2), 5e-4, 5e-5)
        pose_est = np.asarray(T, np.float64)
        est_poses[k] = pose_est.copy()
        tracking_kf = k >= args.frames - 1 or ratio < args.keyframe_th
        since_tracking_kf = 0 if tracking_kf else since_tracking_kf + 1
        mapping_kf = (not tracking_kf) and since_tracking_kf % args.keyframe_freq == 0
        line = f"frame {k}: {reg.iterations} LM iterations, pose error {ang:.4f} deg / {mm:.3f} mm, overlap {ratio:.2f}"
        if tracking_kf or mapping_kf:
            q_cam = rotation_to_quaternion_xyzw(pose_est[:3, :3]).float().to(dev)
            rots_w = quaternion_multiply(q_cam, reg.get_source_rotationsq_tensor())
            scales_s = reg.get_source_scales_tensor().clamp_min(1e-4)
            pw_k = DepthFrontEnd.to_world(pc.points, pose_est)
            if tracking_kf:   # only what the map does not explain yet [REF mp_Tracker.py:266-274]
                sel = pc.trackable_idx.long()[new_idx.to(dev)]
                rows, tmask = rows_from_gicp(pw_k[sel].contiguous(), pc.colors[sel], rots_w[sel].contiguous(), scales_s[sel].contiguous(),
                                             pc.z_values[sel], torch.arange(sel.numel(), device=dev))
            else:             # every point of the frame, none trackable [REF mp_Tracker.py:311-316; mp_Mapper.py:183-187]
                rows, tmask = rows_from_gicp(pw_k.contiguous(), pc.colors, rots_w.contiguous(), scales_s.contiguous(), pc.z_values, None)
            before = store.n
            if rows["xyz"].shape[0] > 0:
                store.append(rows, tmask)          # in place + device-side count: the captured graph keeps replaying
            keyframes.append(Keyframe(pose_est, rgb_dev, d16_dev))
            new_keyframes.append(len(keyframes) - 1)
            line += f" | {'tracking' if tracking_kf else 'mapping'} keyframe: +{store.n - before} Gaussians ({store.n})"
            if tracking_kf:
                n_track_kf += 1
                with torch.no_grad():              # the map goes back to the tracker without leaving the device [REF mp_Mapper.py:168-172]
                    n_t = reg.set_target_from_gaussians(store.live("xyz"), torch.nn.functional.normalize(store.live("rotation")),
                                                        torch.exp(store.live("scaling")), torch.sigmoid(store.live("opacity")),
                                                        store.trackable_mask, 0.05)
                line += f", new target {n_t} points"
            else:
                n_map_kf += 1
        map_some(args.iters)
        lost = mg.ensure_capacity()        # duplicate lists outgrown (steps were skipped on the device): enlarge, re-capture, repeat the lost steps
        if lost:
            line += f" | list capacity grown to {mg.capacity}, {lost} skipped iteration(s) repeated"
            map_some(lost)
        print(line)
    torch.cuda.synchronize()
    t_loop = time.perf_counter() - t_loop
    if args.post_iters > 0:
        map_some(args.post_iters)
        torch.cuda.synchronize()
    if args.eval_every > 0:
        checkpoint(final=True)
    assert not mg.overflowed(), "duplicate-list capacity too small"
    recaptures = captures - 1 + (0 if mg.graph is graph_obj else 1)
    print(f"losses every 25 iterations: {[round(x, 4) for x in losses]}")
    print(f"{args.frames} frames ({args.frames - 1} tracked), {n_track_kf} tracking + {n_map_kf} mapping keyframes, {prunes} prune(s), {train_iter} mapper iterations, "
          f"{store.n} Gaussians, graph captures {captures}, re-captures {recaptures}, skipped optimiser steps {mg.skipped_steps()}")
    print(f"worst pose error {worst[0]:.4f} deg / {worst[1]:.3f} mm; loop wall {t_loop:.2f} s = {(args.frames - 1) / t_loop:.1f} frames/s "
          f"with {args.iters} mapper iterations per frame (frame maker before the loop: {t_gen:.1f} s)")
    if args.save_map:
        with torch.no_grad():
            np.savez(args.save_map, means3D=store.live("xyz").cpu().numpy(), shs=store.live("f_dc").cpu().numpy(),
                     opacities=torch.sigmoid(store.live("opacity")).cpu().numpy(), scales=torch.exp(store.live("scaling")).cpu().numpy(),
                     rotations=torch.nn.functional.normalize(store.live("rotation")).cpu().numpy(),
                     keyframe_poses=np.stack([k_.pose for k_ in keyframes]), mapper_iterations=train_iter, frames=args.frames)
    if args.json:
        import json
        with open(args.json, "w") as fh:
            json.dump(dict(frames=args.frames, iters_per_frame=args.iters, post_iters=args.post_iters, prune_every=args.prune_every,
                           scale_semantics=args.scale_semantics, tracking_keyframes=n_track_kf, mapping_keyframes=n_map_kf, prunes=prunes,
                           mapper_iterations=train_iter, gaussians=store.n, graph_captures=captures, recaptures=recaptures,
                           skipped_steps=mg.skipped_steps(), worst_pose_error_deg_mm=[worst[0], worst[1]], loop_wall_s=t_loop,
                           losses_every_25=losses, quality_curve=curve, data="synthetic analytic room, Replica-shaped 1200x680, noise-free",
                           note="PSNR / SSIM as [REF mp_Mapper.py:335-420] computes them (estimated poses, gt-depth mask); mapper = one captured "
                                "hipGraph per iteration (render_3 + l1/ssim/depth loss + Adam with the reference's learning rates)"), fh, indent=1)
    if args.no_asserts:
        return
    assert recaptures == 0, "the mapper graph had to be re-captured"
    assert worst[0] < 0.05 and worst[1] < 1.5, "tracking drifted"
    if len(losses) >= 4:
        assert min(losses[-2:]) < losses[0], "mapping loss did not fall"
    print("slam demo OK")


if __name__ == "__main__":
    main()
