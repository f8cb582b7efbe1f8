"""The "next" rows of SURVEY.md §8(f) INSIDE the reference's own two-process system.

INTEGRATION.md §6-8 describes the fused mapper iteration, the capacity-based map store, the device-resident hand-off and the front-end kernel as
few-line edits of `mp_Mapper.py`, `mp_Tracker.py`, `scene/gaussian_model.py` and `scene/shared_objs.py`.  This module is what those few lines call;
`oracle/make_refpy.py --fused` applies the edits as an AST transform at build time (into the git-ignored `oracle/_ref/refpy_fused`; no reference
source enters this repository) and `tools/run_reference_slam.py --fused` runs the result.  The edits, all of them:

  scene/gaussian_model.py   + `patch_gaussian_model(GaussianModel)` after the class                                              (rank 4, rank 1)
        the map lives in a `GaussianStore(stable=True)`: create_from_pcd2_tensor / add_from_pcd2_tensor append rows in place, prune_large_and_transparent
        is one stream compaction, training_setup builds `FusedAdam(capturable=True)` over the store's buffers; `_xyz`, `_opacity`, ... stay readable
        (live [:n] views), so render_3 (end-of-run evaluation), save_ply and get_trackable_gaussians_tensor keep working                [REF :134-231, 409-492, 580-592]
  mp_Mapper.py              the statements between `self.training=True` and `self.training = False` (render_3, mask, l1 / ssim / depth loss,
        backward, prune every 200, optimizer.step, zero_grad) -> `fused_mapping_iteration(self, viewpoint_cam, gt_image, gt_depth_image)`:
        set_view + ONE hipGraph replay (one capture per resolution for the whole run)                                                (rank 1)  [REF :219-262]
  scene/shared_objs.py      + `patch_shared_targets(SharedTargetPoints)`: the tracker's new target crosses the process boundary as DEVICE tensors
        (HIP-IPC, like SharedGaussians in the other direction) instead of pinned-less host copies                                   (rank 2)  [REF :99-126]
  mp_Tracker*.py            `get_values_np()` -> `get_values_tensor()` at the tracking keyframe (pygicp takes the device tensors), and
        + `patch_tracker(Tracker)`: downsample_and_make_pointcloud2 runs the front-end kernel (same return values, bit for bit)      (ranks 2, 3)  [REF :282-288, 415-431]

Everything here calls the product (store, graph, optimiser, tracker mirror); nothing falls back to the reference's torch chain when the HIP library is missing.
"""
import math
import os

import torch

from . import _lib
from .gaussian_store import GaussianStore, rows_from_gicp
from .optim import FusedAdam


# ------------------------------------------------------------------------------------------------------------------ scene/gaussian_model.py
def _refresh_views(gm):
    st = gm._store
    gm._xyz, gm._features_dc, gm._features_rest = st.live("xyz"), st.live("f_dc"), st.live("f_rest")
    gm._opacity, gm._scaling, gm._rotation = st.live("opacity"), st.live("scaling"), st.live("rotation")
    gm.max_radii2D = st.view("aux", "max_radii2D")
    gm.xyz_gradient_accum, gm.denom = st.view("aux", "xyz_gradient_accum"), st.view("aux", "denom")
    gm.trackable_mask = st.trackable_mask


def patch_gaussian_model(cls):
    """Store-backed replacements for the GaussianModel methods the mapping process calls [REF mp_Mapper.py:131-136, 163-187, 244-245]."""
    if getattr(cls, "_gsicp_fused", False):
        return cls
    plain_update_lr = cls.update_learning_rate

    def create_from_pcd2_tensor(self, points, colors, rots_, scales_, z_vals_, trackable_idxs):
        # rows of the capacity-backed map (376 B each: parameters, both Adam moments and statistics in two buffer sets: 2 M rows = 0.75 GB of 288 GB;
        # the per-row kernels run their grids over the capacity, so it is not made larger than a run needs).  The reference's map grows without
        # bound [REF scene/gaussian_model.py:474-492]; whole-room Replica maps reach 1-2 M Gaussians: when the capacity is exhausted the store moves
        # into buffers of twice the size (GaussianStore.grow; add_from_pcd2_tensor below) and the mapper iteration is captured again.
        capacity = int(os.environ.get("GSICP_FUSED_CAPACITY", "2000000"))
        n_rest = (self.max_sh_degree + 1) ** 2 - 1
        self._store = GaussianStore(capacity, n_rest=n_rest, device=points.device, stable=True)
        rows, mask = rows_from_gicp(points.float(), colors.float(), rots_.float(), scales_.float(), z_vals_.float(), trackable_idxs, self.max_sh_degree)
        self._store.append(rows, mask)
        _refresh_views(self)
        self.keyframe_idx = torch.ones((self._store.n, 1), dtype=torch.bool, device=points.device)

    def add_from_pcd2_tensor(self, points, colors, rots_, scales_, z_vals_, trackable_idxs):
        rows, mask = rows_from_gicp(points.float(), colors.float(), rots_.float(), scales_.float(), z_vals_.float(),
                                    trackable_idxs if len(trackable_idxs) != 0 else None, self.max_sh_degree)
        if self._store.n + rows["xyz"].shape[0] > self._store.capacity:
            # The reference's map grows without bound [REF scene/gaussian_model.py:474-492]: so does this one (round 5 dropped the keyframe).
            # The store moves into buffers of twice the capacity; every address changes, so the captured iterations are released and captured again on
            # the next mapping iteration (one capture ~ three iterations of time), and the row-freeze mask (policy `freeze`) is bound to the new buffer.
            new_cap = max(2 * self._store.capacity, self._store.n + int(rows["xyz"].shape[0]))
            print(f"GSICP: map capacity {self._store.capacity} rows exhausted ({self._store.n} live + {rows['xyz'].shape[0]} new): growing to {new_cap} rows "
                  f"(one re-capture of the mapper iteration; start with a larger GSICP_FUSED_CAPACITY to avoid it)", flush=True)
            for mg in self.__dict__.get("_gsicp_graphs", {}).values():
                mg.release()
            self.__dict__.get("_gsicp_graphs", {}).clear()
            self._store.grow(new_cap)
            pol = fused_policy()
            if pol["freeze_groups"] and self.optimizer is not None:
                self.optimizer.set_row_freeze(self._store._sets[0][("aux", "trackable_mask")], pol["freeze_groups"])
        self._store.append(rows, mask)          # rows written in place, live count bumped on the device: the captured iteration keeps replaying
        _refresh_views(self)

    def training_setup(self, training_args):
        from utils.general_utils import get_expon_lr_func     # the reference's own scheduler [REF scene/gaussian_model.py:233-236]
        self.percent_dense = training_args.percent_dense
        lrs = {"xyz": training_args.position_lr_init * self.spatial_lr_scale, "f_dc": training_args.feature_lr, "f_rest": training_args.feature_lr / 20.0,
               "opacity": training_args.opacity_lr, "scaling": training_args.scaling_lr, "rotation": training_args.rotation_lr}
        self.optimizer = self._store.attach(FusedAdam, lrs, lr=0.0, eps=1e-15, capturable=True)
        pol = fused_policy()
        print(f"GSICP fused mapper policy: {POLICY_NOTES[pol['name']]}" + (f", {pol['iters_per_frame']} steps per frame" if pol["iters_per_frame"] > 0 else "")
              + "  [GSICP_FUSED_POLICY]", flush=True)
        if pol["freeze_groups"]:      # the `freeze` policy (opt-in): the Gaussians the tracker aligns against keep their geometry (see fused_policy)
            self.optimizer.set_row_freeze(self._store._sets[0][("aux", "trackable_mask")], pol["freeze_groups"])
        self.xyz_scheduler_args = get_expon_lr_func(lr_init=training_args.position_lr_init * self.spatial_lr_scale,
                                                    lr_final=training_args.position_lr_final * self.spatial_lr_scale,
                                                    lr_delay_mult=training_args.position_lr_delay_mult, max_steps=training_args.position_lr_max_steps)

    def update_learning_rate(self, iteration):
        out = plain_update_lr(self, iteration)
        self.optimizer.sync_lr()                # the capturable optimiser reads its learning rates from the device
        return out

    def prune_large_and_transparent(self, min_opacity, extent):
        with torch.no_grad():
            remove = (torch.sigmoid(self._store.live("opacity")) < min_opacity).squeeze(-1)
            if extent is not None:
                remove = torch.logical_or(remove, torch.exp(self._store.live("scaling")).max(dim=1).values > 0.1 * extent)
            self._store.prune(remove)           # one order-preserving compaction of all 23 arrays; addresses unchanged
        _refresh_views(self)

    def get_trackable_gaussians_tensor(self, opacity_th):
        with torch.no_grad():
            keep = torch.logical_and((self.get_opacity > opacity_th).squeeze(-1), self.trackable_mask)
            out = self.get_xyz[keep], self.get_rotation[keep], self.get_scaling[keep]
        return out if os.environ.get("GSICP_FUSED_DEVICE_TARGETS", "1") == "1" else tuple(t.cpu() for t in out)

    for fn in (create_from_pcd2_tensor, add_from_pcd2_tensor, training_setup, update_learning_rate, prune_large_and_transparent,
               get_trackable_gaussians_tensor):
        setattr(cls, fn.__name__, fn)
    cls._gsicp_fused = True
    return cls


# ------------------------------------------------------------------------------------------------------------------ mp_Mapper.py
_STAMPS = []
_GPU_MS = []          # device time of the iterations (hipEvent pairs around set_view + the graph replay), whatever the pacing does to the cadence


def _stamp(mapper):
    """In-system mapper cadence (GSICP_ANNOUNCE, set by tools/run_reference_slam.py): entry times of the iterations; at process exit the interval
    statistics are printed — the counterpart of what the drop-in call trace yields for the untouched loop (one iteration = the time between two
    rasteriser forward calls)."""
    if not os.environ.get("GSICP_ANNOUNCE"):
        return
    import time
    if not _STAMPS:
        import atexit

        def report():
            if len(_STAMPS) > 2:
                d = sorted(b - a for a, b in zip(_STAMPS[:-1], _STAMPS[1:]))
                g = sorted(_GPU_MS) or [float("nan")]
                print(f"GSICP_FUSED_MAPPER iterations {len(_STAMPS)} median_ms {1e3 * d[len(d) // 2]:.4f} mean_ms {1e3 * sum(d) / len(d):.4f} "
                      f"p90_ms {1e3 * d[int(0.9 * (len(d) - 1))]:.4f} captures {mapper.__dict__.get('_gsicp_captures', 0)} gaussians {mapper.gaussians._store.n} "
                      f"gpu_median_ms {g[len(g) // 2]:.4f} gpu_p90_ms {g[int(0.9 * (len(g) - 1))]:.4f} paced_waits {mapper.gaussians.__dict__.get('_gsicp_paced', 0)} "
                      f"iters_per_frame {fused_policy()['iters_per_frame']} policy {fused_policy()['name']}", flush=True)
        atexit.register(report)
    _STAMPS.append(time.perf_counter())


@_lib.traced("refglue.fused_mapping_iteration")
def fused_mapping_iteration(mapper, viewpoint_cam, gt_image, gt_depth_image):
    """What stands where [REF mp_Mapper.py:219-262] stood: one keyframe selection launch + one hipGraph replay (activations inside the rasteriser,
    forward, fused L1 + SSIM + depth loss, backward, capturable Adam), the prune every 200 iterations, and the list-capacity check."""
    from .graph import MapperIterationGraph
    gm = mapper.gaussians
    store = gm._store
    _stamp(mapper)
    H, W = int(gt_image.shape[-2]), int(gt_image.shape[-1])
    graphs = gm.__dict__.setdefault("_gsicp_graphs", {})
    mg = graphs.get((H, W))
    if mg is None:
        params = {"means3D": store.params["xyz"], "shs": store.params["f_dc"], "opacities": store.params["opacity"],
                  "scales": store.params["scaling"], "rotations": store.params["rotation"]}
        if store.n_rest != 0:
            raise RuntimeError("fused mapping iteration: sh_degree > 0 needs the f_dc / f_rest concatenation inside the graph (not built: the reference runs sh_degree 0)")
        mg = MapperIterationGraph(params, gm.optimizer, H, W, math.tan(float(viewpoint_cam.FoVx[0]) * 0.5), math.tan(float(viewpoint_cam.FoVy[0]) * 0.5),
                                  sh_degree=gm.active_sh_degree, capacity=int(os.environ.get("GSICP_FUSED_LIST_CAPACITY", str(1 << 23))),
                                  bg=mapper.background, lambda_dssim=mapper.lambda_dssim, warmup=1, live_count=store.live_count)
        mg.set_view(viewpoint_cam.world_view_transform, viewpoint_cam.full_proj_transform, viewpoint_cam.camera_center, gt_image.contiguous(),
                    gt_depth_image.contiguous())
        mg.capture()                            # applies no optimiser update (the warm-up is rolled back)
        graphs[(H, W)] = mg
        mapper.__dict__["_gsicp_captures"] = mapper.__dict__.get("_gsicp_captures", 0) + 1
    if mapper.train_iter % 200 == 0:
        # [REF mp_Mapper.py:243-248] prunes between loss.backward() and optimizer.step(); prune_points re-creates every Parameter, the fresh ones carry
        # no .grad, so torch's Adam applies NOTHING on these iterations (1 in 200): the iteration's only lasting effect is the prune.  Same here.
        gm.prune_large_and_transparent(0.005, mapper.prune_th)
        return mg.loss_parts[0]
    _pace(mapper, gm)
    # Bounded run-ahead: a graph launch returns at once, so this loop could queue thousands of iterations ahead of the GPU — and the TRACKER process's
    # small kernels would wait behind them (measured without the bound: tracker 17 ms per frame, System FPS 66 instead of 167).  The reference's own
    # iteration is throttled by its synchronous forward; here at most `GSICP_FUSED_INFLIGHT` (2) replays are in flight.
    ring = gm.__dict__.setdefault("_gsicp_ring", [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                                                  for _ in range(max(1, int(os.environ.get("GSICP_FUSED_INFLIGHT", "2"))))])
    done = gm.__dict__.get("_gsicp_steps", 0)
    begin, end = ring[done % len(ring)]
    if done >= len(ring):
        end.synchronize()
        if len(_GPU_MS) < 100000:
            _GPU_MS.append(begin.elapsed_time(end))
    begin.record()
    mg.set_view(viewpoint_cam.world_view_transform, viewpoint_cam.full_proj_transform, viewpoint_cam.camera_center, gt_image.contiguous(),
                gt_depth_image.contiguous())
    loss = mg.step()
    end.record()
    gm.__dict__["_gsicp_steps"] = done + 1
    if done % 10 == 9 and mg.ensure_capacity():
        # duplicate lists outgrown: at most ten replays rendered nothing and had their optimiser steps skipped ON THE DEVICE (nothing drifted); the lists
        # are enlarged and the iteration re-captured.  The skipped steps are NOT repeated on the current view: the loop simply goes on.
        mapper.__dict__["_gsicp_captures"] = mapper.__dict__.get("_gsicp_captures", 0) + 1
    return loss


# ---- How the in-system fused mapper shares the map with the tracker: the POLICY (GSICP_FUSED_POLICY), measured in profiles/r05_fused_pacing_sweep*.json.
# The reference's mapper free-runs [REF mp_Mapper.py:150-262] and the tracker re-targets on the map's trackable Gaussians — position, rotation
# and scale as the optimiser has left them [REF mp_Tracker.py:282-288; scene/gaussian_model.py:207-215] — at every tracking keyframe.  How far
# those have been optimised by then depends on the mapper's speed: the untouched loop fits 0.3-2 Adam steps into a tracked frame on this box,
# the fused one 7-50.  On NOISY depth that is not neutral: 300 noisy frames, untouched ATE 0.8-1.0 cm; fused free-running 6.8-9.6 cm; fused with
# only the POSITIONS of the trackable Gaussians frozen 9.6 cm; with position + scale + rotation frozen 0.56-0.62 cm — the damage goes through the
# covariances the tracker receives (thousands of steps on noisy depth stretch the Gaussians into needles, the GICP planes follow the needles), not
# through the iteration count as such.  Policies:
#   free (DEFAULT since round 6: the reference's optimiser, every Gaussian trains [REF mp_Mapper.py:243-248; scene/gaussian_model.py:222-231])
#                     free-run, nothing frozen, nothing paced: result parity with the reference's statements (tests/test_reference_slam_gpu.py pins the
#                     arithmetic).  Noise-free synthetic data: ATE 0.005 cm.  On noisy synthetic depth it loses accuracy (see above); until that is
#                     re-judged on REAL TUM / Replica data (none on either box) the deviation below stays an explicit opt-in.
#   freeze            OPT-IN, DEVIATES from the reference's optimiser: free-run; the TRACKABLE Gaussians [REF scene/gaussian_model.py:143-180 trackable_mask] keep the position, scale and rotation
#                     GICP gave them (FusedAdam.set_row_freeze: those rows of xyz / scaling / rotation are no parameters; colour and opacity of
#                     every Gaussian and the geometry of the non-trackable ones train as in the reference).  Meets the bar — ATE <= untouched + 0.1 cm,
#                     PSNR >= untouched — with the MOST iterations: 300 noisy frames 0.56-0.62 cm / 27.7-28.2 dB at 2 050-2 110 iterations
#                     (untouched 0.77-1.04 cm / 22.8-23.3 dB), 16 363 iterations at the 30-FPS cap 0.67 cm / 28.3 dB (round 4, free-running: 9.3 cm),
#                     TUM-shaped 0.31-0.37 cm / 27.3 dB (untouched 0.31 cm / 17.3 dB), noise-free 0.005 cm / 34.1 dB.
#   budget            DIAGNOSTIC: the reference's optimiser untouched, at most GSICP_FUSED_ITERS_PER_FRAME (2) steps per tracked frame, the GPU time it
#                     frees left to the tracker: 0.74-0.84 cm / 25.1-25.7 dB over 300 frames (614 steps) — but it only postpones the damage: 4 per frame
#                     lose the track over 300 frames (7.8 cm), 2 per frame over 600 (14.5 cm at 1 217 steps, where `freeze` gives 0.61 cm at 4 061).
# GSICP_FUSED_FREEZE_GROUPS overrides which parameter groups the freeze covers (e.g. "xyz": the experiment that showed positions are not the cause).
DEFAULT_POLICY = "free"
DEFAULT_ITERS_PER_FRAME = 2.0
POLICY_NOTES = {"free": "free (the reference's optimiser: every Gaussian trains)",
                "freeze": "freeze (DEVIATES from the reference's optimiser: position / scale / rotation of trackable Gaussians frozen)",
                "budget": "budget (the reference's optimiser, paced to a number of steps per tracked frame)"}


def fused_policy():
    name = os.environ.get("GSICP_FUSED_POLICY", DEFAULT_POLICY)
    if name not in ("freeze", "budget", "free"):
        raise RuntimeError(f"GSICP_FUSED_POLICY={name!r}: expected freeze, budget or free")
    groups = ()
    if name == "freeze":
        groups = tuple(g for g in os.environ.get("GSICP_FUSED_FREEZE_GROUPS", "xyz,scaling,rotation").split(",") if g)
    k = float(os.environ.get("GSICP_FUSED_ITERS_PER_FRAME", str(DEFAULT_ITERS_PER_FRAME if name == "budget" else 0.0)))
    return dict(name=name, freeze_groups=groups, iters_per_frame=k)


def _pace(mapper, gm):
    """Pacing policy of the fused mapper: hold the loop to `k` optimiser steps per frame the tracker has consumed (the shared frame counter
    [REF gs_icp_slam.py:95; mp_Tracker.py:117]).  Every call of fused_mapping_iteration still performs exactly ONE iteration — the loop's own
    bookkeeping (new keyframes trained once first, train_iter, the prune cadence) is untouched; pacing only delays it.  The wait ends at once when
    the tracker raises a keyframe flag or the end of the dataset (it blocks on the mapper there [REF mp_Tracker.py:285-286]) and is bounded
    (`GSICP_FUSED_MAX_WAIT_MS`) so a stalled frame counter cannot hang the mapper.  `GSICP_FUSED_MIN_PERIOD_MS` (a wall-clock period) is kept
    as a second, diagnostic knob."""
    import time
    period = float(os.environ.get("GSICP_FUSED_MIN_PERIOD_MS", "0"))
    if period > 0:
        t_prev = gm.__dict__.get("_gsicp_t_prev")
        if t_prev is not None:
            wait = t_prev + 1e-3 * period - time.perf_counter()
            if wait > 0:
                time.sleep(wait)
        gm.__dict__["_gsicp_t_prev"] = time.perf_counter()
    k = fused_policy()["iters_per_frame"]
    frames = getattr(mapper, "iter_shared", None)
    flags = [getattr(mapper, n, None) for n in ("end_of_dataset", "is_tracking_keyframe_shared", "is_mapping_keyframe_shared")]
    if k <= 0 or frames is None or any(f is None for f in flags):
        return
    done = gm.__dict__.get("_gsicp_steps", 0)
    burst = float(os.environ.get("GSICP_FUSED_BURST", "8"))          # the first keyframe's iterations, before any frame has been counted
    deadline = time.perf_counter() + 1e-3 * float(os.environ.get("GSICP_FUSED_MAX_WAIT_MS", "250"))
    waited = False
    while done >= burst + k * (int(frames[0]) + 1) and not any(int(f[0]) for f in flags) and time.perf_counter() < deadline:
        time.sleep(5e-5)
        waited = True
    if waited:
        gm.__dict__["_gsicp_paced"] = gm.__dict__.get("_gsicp_paced", 0) + 1


# ------------------------------------------------------------------------------------------------------------------ scene/shared_objs.py
def patch_shared_targets(cls):
    """SharedTargetPoints [REF scene/shared_objs.py:99-126] with DEVICE buffers: the mapper writes the trackable Gaussians with a device-to-device
    copy, the tracker process reads them through HIP-IPC (how SharedGaussians already crosses in the other direction).  The writer synchronises
    before it raises `target_gaussians_ready` [REF mp_Mapper.py:170-171]: the flag is host shared memory, the copies are asynchronous."""
    if getattr(cls, "_gsicp_fused", False) or os.environ.get("GSICP_FUSED_DEVICE_TARGETS", "1") != "1":
        return cls
    plain_init, plain_input = cls.__init__, cls.input_values

    def __init__(self, num_points):
        plain_init(self, min(int(num_points), int(os.environ.get("GSICP_FUSED_TARGET_CAPACITY", "3000000"))))
        self.xyz, self.rots, self.scales = self.xyz.cuda(), self.rots.cuda(), self.scales.cuda()

    def input_values(self, new_xyz, new_rots, new_scales):
        plain_input(self, new_xyz, new_rots, new_scales)
        torch.cuda.synchronize()

    cls.__init__, cls.input_values = __init__, input_values
    cls._gsicp_fused = True
    return cls


# ------------------------------------------------------------------------------------------------------------------ mp_Tracker.py
def patch_tracker(cls):
    """Tracker.downsample_and_make_pointcloud2 [REF mp_Tracker.py:415-431] through the front-end kernel: depth and colour go to the device, ONE launch
    picks / converts / compacts / back-projects, the four arrays come back as numpy with the values the reference's torch-CPU chain yields bit for bit
    (tests/test_frontend.py, tests/test_hostcode_pinned.py) — the rest of Tracker.tracking keeps its numpy interface."""
    if getattr(cls, "_gsicp_fused", False):
        return cls
    import numpy as np
    plain_init = cls.__init__

    def __init__(self, slam, *args, **kw):
        # the shared frame counter [REF gs_icp_slam.py:95]: mp_Tracker.py keeps and advances it [REF mp_Tracker.py:35, 117], mp_Tracker_unlimit.py does
        # neither — the fused mapper's pacing policy reads it (refglue._pace), so the patched front-end advances it in both variants
        plain_init(self, slam, *args, **kw)
        if not hasattr(self, "iter_shared") and hasattr(slam, "iter_shared"):
            self.iter_shared = slam.iter_shared

    def downsample_and_make_pointcloud2(self, depth_img, rgb_img):
        from .frontend import DepthFrontEnd
        if getattr(self, "iter_shared", None) is not None:
            self.iter_shared[0] = int(getattr(self, "iteration_images", 0))
        fe = self.__dict__.get("_gsicp_frontend")
        if fe is None:
            fe = DepthFrontEnd(self.H, self.W, self.fx, self.fy, self.cx, self.cy, self.downsample_rate, self.depth_scale, self.depth_trunc)
            self.__dict__["_gsicp_frontend"] = fe
        d = np.ascontiguousarray(depth_img)
        if d.dtype == np.uint16:
            d_dev = torch.from_numpy(d.view(np.int16)).cuda()
        else:
            d_dev = torch.from_numpy(d.astype(np.float32)).cuda()
        pc = fe.make_pointcloud(d_dev, torch.from_numpy(np.ascontiguousarray(rgb_img)).cuda())
        return pc.points.cpu().numpy(), pc.colors.cpu().numpy(), pc.z_values.cpu().numpy(), pc.trackable_idx.cpu().numpy().astype(np.int64)

    cls.__init__ = __init__
    cls.downsample_and_make_pointcloud2 = downsample_and_make_pointcloud2
    cls._gsicp_fused = True
    return cls
