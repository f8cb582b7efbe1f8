"""One mapper iteration as a replayable HIP graph.

The reference's mapping loop [REF mp_Mapper.py:219-248] launches ~60 small kernels per iteration from Python (activations,
rasteriser forward with a host round trip, loss chain, backward, Adam); at P = 300 k the GPU work is ~0.55 ms and the
Python/launch path around it costs as much again.  With the sync-free forward (`GaussianRasterizationSettings.capacity`),
the fused loss and the capturable fused Adam, nothing in the iteration needs the host, so the whole iteration is captured
once (`torch.cuda.CUDAGraph`, i.e. hipGraph) and replayed with one launch per iteration.

What changes from iteration to iteration in the reference — the keyframe: camera matrices and the two target images —
is selected by `set_view()` before `step()`: the camera goes into static device buffers, the two images are read IN PLACE through a device
slot pair holding their addresses (round 5; rounds 2-4 copied 13 MB per iteration into static image buffers).  Everything with a fixed address (parameters,
Adam state, learning rates, step count) is updated in place by the replay.  The graph is valid while the parameter tensors
are the ones captured.  Over a `GaussianStore(stable=True)` (full-capacity buffers, live count on the device: pass `live_count=`) that is the
WHOLE RUN — keyframe growth and pruning change no pointer and no launch grid, so nothing is ever re-captured.  Over reference-style storage
(new parameter tensors after every append / prune [REF scene/gaussian_model.py:409-492]) build a new MapperIterationGraph after each change —
capture costs about three eager iterations of time but applies NO optimiser update (warm-up is rolled back).

Several GPUs: pass `rasterizer_factory=lambda rs: ShardedGaussianRasterizer(rs, vis_capacity=R)` (sharded.py).  Its two RCCL collectives
have static sizes and are captured with everything else; the overflow guard then is the all-reduced flag, identical on every rank.
"""
import ctypes
import os
import time

import torch

from . import _lib
from .activations import activate
from .loss import mapper_loss_and_grads
from .optim import FusedAdam
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer


def process_group_live():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def drain_process_group_watchdog(dev, seconds=None):
    """Make a stream capture safe next to a live ProcessGroupNCCL.  The group's watchdog thread polls every outstanding
    collective with hipEventQuery about every 100 ms until it has seen it finished; a capture in HIP's default GLOBAL mode forbids that call from
    ANY thread, the query throws on the watchdog thread and the process is terminated (`operation not permitted when stream is capturing`).
    Two measures, either of which is sufficient: (1) every capture of this package is begun in THREAD_LOCAL mode while a group is initialised
    (`capture_mode()`), which restricts the rule to the capturing thread; (2) before capturing, the device is synchronised (all eager
    collectives have finished) and the watchdog is given `seconds` (default 0.5 = several polling periods; GSICP_CAPTURE_DRAIN_S) to retire
    them, so that nothing is left for it to poll during the capture.  No collective is issued here: a graph may be captured by one rank alone."""
    if not process_group_live():
        return
    torch.cuda.synchronize(dev)
    if seconds is None:
        seconds = float(os.environ.get("GSICP_CAPTURE_DRAIN_S", "0.5"))
    time.sleep(max(0.0, seconds))
    torch.cuda.synchronize(dev)


def capture_mode():
    """capture_error_mode for torch.cuda.graph: thread_local while a process group is initialised (see drain_process_group_watchdog), else the default."""
    return "thread_local" if process_group_live() else "global"


def default_activations(p, live_rows=None):
    """GaussianModel's property getters [REF scene/gaussian_model.py:105-125] on the raw parameter tensors (one fused launch)."""
    opacities, scales, rotations = activate(p["opacities"], p["scales"], p["rotations"], live_rows)
    return dict(means3D=p["means3D"], shs=p["shs"], opacities=opacities, scales=scales, rotations=rotations)


def torch_activations(p, live_rows=None):
    """The same with the reference's torch ops (sigmoid / exp / normalize)."""
    return dict(means3D=p["means3D"], shs=p["shs"], opacities=torch.sigmoid(p["opacities"]), scales=torch.exp(p["scales"]),
                rotations=torch.nn.functional.normalize(p["rotations"]))


class MapperIterationGraph:
    """params: dict of raw leaf tensors (means3D, shs, opacities, scales, rotations) with requires_grad;
    optimizer: FusedAdam(capturable=True) over them;  capacity: upper bound for the number of (Gaussian, tile) duplicates
    (e.g. 1.5x the count of an eager forward; `overflowed()` tells when it was too small, `ensure_capacity()` enlarges it and re-captures)."""

    def __init__(self, params, optimizer, image_height, image_width, tanfovx, tanfovy, sh_degree, capacity, bg=None, lambda_dssim=0.2,
                 depth_weight=0.1, d_max=10.0, activations=None, rasterizer_factory=None, warmup=2, live_count=None,
                 depth_mode=0, grad_hook=None):
        # grad_hook(params): optional, runs between the backward and the optimiser step INSIDE the captured iteration (static tensors only) —
        # e.g. refglue's experiment knob that zeroes the geometry gradients of Gaussians the tracker aligns against.
        self._grad_hook = grad_hook
        # activations=None (default): the rasteriser takes the RAW parameters and applies sigmoid / exp / normalize and their chain rule inside
        # its preprocess kernels (GaussianRasterizationSettings.raw_params) — two launches and 64 B per Gaussian of traffic less per iteration;
        # pass default_activations / torch_activations to run them as separate operators instead.
        fused = activations is None
        if not isinstance(optimizer, FusedAdam) or not optimizer.capturable:
            raise RuntimeError("MapperIterationGraph needs FusedAdam(capturable=True): a host-side step count cannot be replayed")
        if capacity <= 0:
            raise RuntimeError("MapperIterationGraph needs a positive duplicate-list capacity")
        dev = params["means3D"].device
        self.params, self.optimizer, self.activations = params, optimizer, activations
        self.capacity = int(capacity)
        self.lambda_dssim, self.depth_weight, self.d_max = float(lambda_dssim), float(depth_weight), float(d_max)
        H, W = int(image_height), int(image_width)
        f32 = dict(dtype=torch.float32, device=dev)
        # static inputs of the captured iteration
        self.viewmatrix = torch.eye(4, **f32)
        self.projmatrix = torch.eye(4, **f32)
        self.campos = torch.zeros(3, **f32)
        # ground truth of the selected keyframe: NOT copied (round 5).  The captured loss kernels read the two image pointers from `gt_slots`
        # (device int64[2]); set_view() writes them together with the camera in one 64-thread launch and keeps the tensors alive.  Inputs that
        # cannot be read in place (host tensors, other dtypes, strided views) go through the two staging buffers, allocated on first need.
        self._H, self._W = H, W
        self.gt_slots = torch.zeros(2, dtype=torch.int64, device=dev)
        self._gt_refs = None
        self._gt_ring, self._gt_ring_depth = [], max(1, int(os.environ.get("GSICP_VIEW_REFS", "8")))
        self._gt_stage = None
        self.bg = torch.zeros(3, **f32) if bg is None else bg.to(**f32)
        rs = GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=float(tanfovx), tanfovy=float(tanfovy), bg=self.bg, scale_modifier=1.0,
            viewmatrix=self.viewmatrix, projmatrix=self.projmatrix, sh_degree=int(sh_degree), campos=self.campos, prefiltered=False,
            debug=False, capacity=self.capacity, live_count=live_count, depth_mode=int(depth_mode), raw_params=fused)
        self._fused_activations = fused
        self._rs, self._rasterizer_factory = rs, rasterizer_factory
        self._skipped_seen = 0
        self.regrowths = 0          # times ensure_capacity() had to enlarge the lists and re-capture
        # live_count (int32[1] device tensor): `params` are the FULL-CAPACITY buffers of a GaussianStore(stable=True) and only the first
        # live_count[0] rows are Gaussians.  Growth and pruning then change that number and rows in place — no pointer, shape or launch grid
        # of the captured iteration changes, so ONE capture serves the whole run [REF mp_Mapper.py:161-195, 244-245 append / prune].
        self.live_count = live_count
        # SPARSE gradients (round 6): this object owns every reader of the parameter gradients (the optimiser step inside the captured iteration), so the
        # backward need not write — nor FusedAdam read — the zero rows of culled Gaussians: the rasteriser gets `sparse_grads`, the optimiser the forward's
        # radii as its row mask.  Needs the in-kernel activations (raw parameters: no separate activation backward reading those rows) and a rasteriser
        # whose own gradient handling touches visible rows only (plain; tile-sharded with the static exchange — NOT the dense keyframe-parallel
        # all-reduce).  GSICP_SPARSE_GRADS=0 switches it off (A/B).  A grad_hook sees undefined values in culled rows: mask by `radii > 0`.
        self._sparse = bool(fused and os.environ.get("GSICP_SPARSE_GRADS", "1") != "0")
        if self._sparse:
            rs = rs._replace(sparse_grads=True)
        self.rasterizer = rasterizer_factory(rs) if rasterizer_factory is not None else GaussianRasterizer(rs)
        if self._sparse and rasterizer_factory is not None and not getattr(self.rasterizer, "sparse_grads_ok", lambda: False)():
            self._sparse = False           # this rasteriser reads every gradient row itself (e.g. the dense keyframe-parallel all-reduce): build it again, dense
            rs = rs._replace(sparse_grads=False)
            self.rasterizer = rasterizer_factory(rs)
        self._rs = rs
        self._warmup = int(warmup)
        # device-side overflow guard: a replay whose duplicate count exceeds the capacity renders nothing; the Adam kernels
        # read the count and skip that step entirely (no stale-momentum drift, step count not advanced), counting it in a sticky counter
        inner = self.rasterizer.inner if hasattr(self.rasterizer, "inner") else self.rasterizer
        if getattr(inner, "num_rendered", None) is None:
            inner.num_rendered = torch.zeros(1, dtype=torch.int32, device=dev)
        self._guard_count, self._guard_limit = inner.num_rendered, self.capacity   # bound to the optimiser in capture(): several graphs (e.g.
        # one per training_stage resolution) may share one optimiser, each with its own guard
        shared_guard = self.rasterizer.overflow_guard() if hasattr(self.rasterizer, "overflow_guard") else None
        if shared_guard is not None:     # tile-sharded across GPUs (sharded.py, static exchange): 1 when ANY rank overflowed, so all ranks skip alike
            self._guard_count, self._guard_limit = shared_guard
        # STEP BUMP inside the loss (round 6): the guard is this rank's duplicate count, final after the forward — the loss kernel's finishing thread
        # advances the step counter and the Adam launch runs `step_already_bumped` (no one-thread bump launch: 4.1 us per iteration).  The shared guard
        # of the multi-GPU exchange is only known after the backward: there the bump launch stays.  GSICP_STEP_BUMP_IN_LOSS=0 switches it off (A/B).
        self._bump_in_loss = shared_guard is None and os.environ.get("GSICP_STEP_BUMP_IN_LOSS", "1") != "0"
        # PRE-ZEROED forward (round 6): the captured forward replays with the same scratch buffers, so the counter region it would clear with a launch of its own is
        # cleared by the keyframe-selection launch that precedes every replay anyway (gsicp_mapper_select_view_zero; 4.9 us of launch floor per iteration, 15 -> 14
        # kernel nodes).  The region is read from the library right after the capture; step() re-issues the selection when no set_view() came since the last replay.
        # GSICP_PREZERO=0 switches it off (A/B).
        self._prezero = os.environ.get("GSICP_PREZERO", "1") != "0"
        self._zero_region = None        # (pointer, 4-byte words) of the captured forward's counter region
        self._view_fresh = False        # a selection launch has been issued since the last replay
        # The guard and the live-row count are bound to the optimiser only WHILE this graph's launches are issued (capture(), warm-up included)
        # and what was bound before is restored afterwards: an eager step() on the same optimiser later is gated by whatever ITS
        # caller bound (GaussianStore.attach binds the store's live count), not by the count the last replay happened to leave behind.
        if optimizer.skipped_steps is None or optimizer.skipped_steps.device != self._guard_count.device:
            optimizer.skipped_steps = torch.zeros(1, dtype=torch.int32, device=self._guard_count.device)
        # screen-space gradient holder [REF gaussian_renderer/__init__.py:227]: the reference makes a fresh zero tensor per call;
        # its VALUE is never read by the rasteriser, so one static tensor serves every replay
        self._means2D = torch.zeros_like(params["means3D"], requires_grad=True)
        self.graph = None
        # static outputs
        self.loss_parts = None      # tensor([loss, L1, SSIM mean, depth L1]) of the last replay
        self.radii = None
        self.is_used = None
        self.screenspace_grad = None    # (P,3); with sparse gradients (the default here) only the rows with radii > 0 are defined — mask as the
        #                                 reference does with its visibility_filter [REF gaussian_renderer/__init__.py:307-308]
        self.num_rendered = None    # int32[1]: true duplicate count of the last replay

    # ------------------------------------------------------------------------------------------------------------
    def set_view(self, viewmatrix, projmatrix, campos, gt_image, gt_depth, copy=False):
        """Select the keyframe of the next step(): ONE 64-thread launch writes the camera into the graph's static inputs and the addresses of the two
        ground-truth images into the slot pair the captured loss kernels read.  The images are used IN PLACE: they must not be modified until the
        replay that reads them has finished.  This object keeps references to the tensors of the last `GSICP_VIEW_REFS` (8) selections — more than any
        caller here keeps replays in flight (bench.py / refglue: 2) — and marks them as in use on the replay's stream, so freeing them early is safe;
        an in-place EDIT of a selected image while its replay is queued is not detectable: pass `copy=True` (the images go through this object's own
        staging buffers: two copies per selection, rounds 2-4's behaviour) when the caller cannot guarantee that."""
        dev = self.gt_slots.device
        HW = self._H * self._W
        f32 = lambda t, n: t.is_cuda and t.device == dev and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n   # noqa: E731
        cam_ok = f32(viewmatrix, 16) and f32(projmatrix, 16) and f32(campos, 3)
        if not cam_ok:
            viewmatrix = viewmatrix.to(device=dev, dtype=torch.float32).contiguous()
            projmatrix = projmatrix.to(device=dev, dtype=torch.float32).contiguous()
            campos = campos.to(device=dev, dtype=torch.float32).reshape(3).contiguous()
        if copy or not (f32(gt_image, 3 * HW) and f32(gt_depth, HW) and gt_image.data_ptr() % 16 == 0 and gt_depth.data_ptr() % 16 == 0):
            if self._gt_stage is None:
                self._gt_stage = (torch.zeros((3, self._H, self._W), dtype=torch.float32, device=dev),
                                  torch.zeros((1, self._H, self._W), dtype=torch.float32, device=dev))
            self._gt_stage[0].copy_(gt_image.reshape(self._gt_stage[0].shape), non_blocking=True)
            self._gt_stage[1].copy_(gt_depth.reshape(self._gt_stage[1].shape), non_blocking=True)
            gt_image, gt_depth = self._gt_stage
        self._gt_refs = (viewmatrix, projmatrix, campos, gt_image, gt_depth)
        self._select()
        cur = torch.cuda.current_stream(dev)
        for t in self._gt_refs:
            t.record_stream(cur)            # the caching allocator must not hand this memory to another stream while the replay may still read it
        self._gt_ring.append(self._gt_refs)
        if len(self._gt_ring) > self._gt_ring_depth:
            self._gt_ring.pop(0)

    def _select(self):
        """The keyframe-selection launch for the tensors of the last set_view(); with a captured, pre-zeroed forward it also clears that forward's counter region."""
        viewmatrix, projmatrix, campos, gt_image, gt_depth = self._gt_refs
        dev = self.gt_slots.device
        lib = _lib.load()
        p = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            if self._zero_region is not None:
                _lib.check(lib.gsicp_mapper_select_view_zero(p(viewmatrix), p(projmatrix), p(campos), p(gt_image), p(gt_depth), p(self.viewmatrix), p(self.projmatrix),
                                                             p(self.campos), p(self.gt_slots), ctypes.c_void_p(self._zero_region[0]), self._zero_region[1], stream),
                           "gsicp_mapper_select_view_zero")
            else:
                _lib.check(lib.gsicp_mapper_select_view(p(viewmatrix), p(projmatrix), p(campos), p(gt_image), p(gt_depth), p(self.viewmatrix),
                                                        p(self.projmatrix), p(self.campos), p(self.gt_slots), stream), "gsicp_mapper_select_view")
        self._view_fresh = True

    def _iteration(self):
        if self._fused_activations:
            a = dict(means3D=self.params["means3D"], shs=self.params["shs"], opacities=self.params["opacities"], scales=self.params["scales"],
                     rotations=self.params["rotations"])
        else:
            a = self.activations(self.params, self.live_count) if self.live_count is not None else self.activations(self.params)
        depth, color, radii, used = self.rasterizer(means3D=a["means3D"], means2D=self._means2D, shs=a["shs"], opacities=a["opacities"],
                                                    scales=a["scales"], rotations=a["rotations"])
        # the loss kernels produce dL/dimage and dL/ddepth directly: no autograd node, no ones_like / multiply launches.  Tile-sharded over several
        # GPUs the loss is sharded with the tiles: this rank's 32x32 blocks only; its share of the four values travels inside the gradient exchange
        shard = self.rasterizer.loss_shard() if hasattr(self.rasterizer, "loss_shard") else (1, 0)
        step_bump = None
        if self._bump_in_loss:
            step_t = self.optimizer.shared_step_tensor()      # None before the optimiser's first step (the first warm-up iteration): the bump launch then
            if step_t is not None:
                step_bump = (step_t, self._guard_count, self._guard_limit, self.optimizer.skipped_steps)
        parts, g_color, g_depth = mapper_loss_and_grads(color, depth, None, None, lambda_dssim=self.lambda_dssim, depth_weight=self.depth_weight,
                                                        d_max=self.d_max, tile_mod=shard[0], tile_rem=shard[1], gt_slots=self.gt_slots, step_bump=step_bump)
        if shard[0] > 1:
            self.rasterizer.attach_loss_share(parts)
        torch.autograd.backward((color, depth), (g_color, g_depth))
        if shard[0] > 1:
            parts = self.rasterizer.summed_loss()
        self.screenspace_grad = self._means2D.grad     # (P,3) viewspace gradient of this iteration (densification statistics)
        self._means2D.grad = None
        if self._grad_hook is not None:
            self._grad_hook(self.params)
        if self._sparse:
            self.optimizer.set_grad_row_mask(radii)      # culled rows: g = 0, not read (the backward did not write them)
        try:
            self.optimizer.step(step_already_bumped=step_bump is not None)
        finally:
            self.optimizer.set_grad_row_mask(None)       # an eager step() on this optimiser by somebody else reads every gradient element again
        self.optimizer.zero_grad(set_to_none=True)
        return parts, radii, used

    def capture(self):
        """Warm up (allocator pools, device lr array, Adam state) and capture.  The warm-up iterations run on the live parameters
        but are NOT optimiser steps of the caller's schedule: parameters, both moments and the step counters are snapshotted before and
        restored after, so capturing (and re-capturing after the map changed) applies zero updates — the reference performs exactly one
        update per loop iteration [REF mp_Mapper.py:219-248]."""
        with self.optimizer.scoped_bindings(guard=(self._guard_count, self._guard_limit), live_rows=self.live_count):
            return self._capture()   # the launches captured inside read THIS graph's count / limit / live rows

    def _capture(self):
        if self._gt_refs is None:
            raise RuntimeError("MapperIterationGraph.capture(): call set_view() first (the captured loss kernels read the keyframe's images through it)")
        dev = self.params["means3D"].device
        self.optimizer.zero_grad(set_to_none=True)
        snap_p = {k: v.detach().clone() for k, v in self.params.items()}
        snap_s = {}
        for p in self.params.values():
            st = self.optimizer.state.get(p, {})
            snap_s[p] = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()}
        skipped0 = None if self.optimizer.skipped_steps is None else self.optimizer.skipped_steps.clone()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):          # eager warm-up on a side stream
            for _ in range(max(self._warmup, 1)):
                self._iteration()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        with torch.no_grad():
            steps_done = set()
            for k, v in self.params.items():
                v.copy_(snap_p[k])
                st = self.optimizer.state.get(v, {})
                for name, val in st.items():
                    if not torch.is_tensor(val):
                        continue
                    if name == "step":
                        if val.data_ptr() in steps_done:
                            continue
                        steps_done.add(val.data_ptr())
                    old = snap_s[v].get(name)
                    if torch.is_tensor(old):
                        val.copy_(old)
                    else:                         # state was created by the warm-up: its pre-warm-up value is zero
                        val.zero_()
            if skipped0 is not None:
                self.optimizer.skipped_steps.copy_(skipped0)
        torch.cuda.synchronize(dev)
        drain_process_group_watchdog(dev)
        self.graph = torch.cuda.CUDAGraph()
        self._zero_region = None
        inner = self.rasterizer.inner if hasattr(self.rasterizer, "inner") else self.rasterizer
        plain_settings = inner.raster_settings
        try:
            if self._prezero:      # the CAPTURED forward only: the warm-up iterations above cleared their own (per-call) scratch
                inner.raster_settings = plain_settings._replace(prezeroed=True)
            with torch.cuda.graph(self.graph, capture_error_mode=capture_mode()):
                parts, radii, used = self._iteration()
            if self._prezero:      # this thread's last forward call IS the captured one: its counter region keeps its address for the life of the graph
                ptr, words = ctypes.c_void_p(), ctypes.c_size_t()
                _lib.check(_lib.load().gsicp_raster_last_zero_region(ctypes.byref(ptr), ctypes.byref(words)), "gsicp_raster_last_zero_region")
                self._zero_region = (int(ptr.value), int(words.value))
        except Exception:
            self.graph = None       # a failed capture leaves no half-built graph behind: the caller may fall back to eager iterations
            self._zero_region = None
            self.optimizer.zero_grad(set_to_none=True)
            self._means2D.grad = None
            raise
        finally:
            inner.raster_settings = plain_settings
        self._view_fresh = False     # the first replay needs a selection launch that also clears the region
        self.loss_parts, self.radii, self.is_used = parts, radii, used
        self.num_rendered = self.rasterizer.num_rendered if hasattr(self.rasterizer, "num_rendered") else None
        if self.num_rendered is None and hasattr(self.rasterizer, "inner"):
            self.num_rendered = self.rasterizer.inner.num_rendered
        return self

    def step(self):
        """Replay one iteration (one graph launch).  Returns the static loss tensor (0-dim view; read it when needed)."""
        if self.graph is None:
            self.capture()
        if self._zero_region is not None and not self._view_fresh:
            self._select()          # no set_view() since the last replay: the same keyframe again — and the counter region cleared
        self.graph.replay()
        self._view_fresh = False
        return self.loss_parts[0]

    def release(self):
        """Drop the captured hipGraph (a later step() captures again).  Call it BEFORE `torch.distributed.destroy_process_group()` when the
        iteration holds RCCL collectives: a graph must not outlive the communicator whose kernels it replays."""
        if self.graph is not None:
            torch.cuda.synchronize(self.params["means3D"].device)
            self.graph.reset()
            self.graph = None
            self._zero_region = None

    def overflowed(self):
        """True when the last replay produced more duplicates than the capacity (it then rendered nothing and its optimiser step was
        skipped on the device).  Synchronises."""
        if self._guard_count is not self.num_rendered and int(self._guard_count.item()) > self._guard_limit:   # sharded: some rank overflowed
            return True
        return self.num_rendered is not None and int(self.num_rendered.item()) > self.capacity

    def ensure_capacity(self, growth=1.5):
        """Capacity auto-grow.  Call between replays whenever a host synchronisation is acceptable (e.g. once per
        keyframe): if replays since the last call overflowed the duplicate lists — they rendered nothing and their optimiser steps were skipped
        on the device, nothing drifted — the lists are enlarged to max(growth x capacity, 1.25 x the count that overflowed) and the iteration
        is captured again over the same parameters and optimiser state.  Returns the number of optimiser steps lost since the last call, so
        that the caller can repeat them.  With a tile-sharded rasteriser every rank sees the same all-reduced flag and grows alike."""
        now = self.skipped_steps()
        lost = now - self._skipped_seen
        self._skipped_seen = now
        grow = lost > 0 or self.overflowed()
        need = int(self.num_rendered.item()) if self.num_rendered is not None else 0
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and hasattr(self.rasterizer, "overflow_guard"):
            # every rank must take the same decision AND the same new capacity: the re-capture below warms up with collectives (ADVICE r3:
            # a rank-local guard could make ranks diverge and deadlock there), and static exchange sizes derive from the capacity
            t = torch.tensor([int(grow), need, lost], dtype=torch.int64, device=self.params["means3D"].device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            grow, need, lost = bool(t[0].item()), int(t[1].item()), int(t[2].item())
        if not grow:
            return 0
        self.capacity = max(int(self.capacity * growth) + 1, int(1.25 * need) + 4096)
        self._rs = self._rs._replace(capacity=self.capacity)   # (carries sparse_grads)
        dev = self.params["means3D"].device
        self.rasterizer = self._rasterizer_factory(self._rs) if self._rasterizer_factory is not None else GaussianRasterizer(self._rs)
        inner = self.rasterizer.inner if hasattr(self.rasterizer, "inner") else self.rasterizer
        if getattr(inner, "num_rendered", None) is None:
            inner.num_rendered = torch.zeros(1, dtype=torch.int32, device=dev)
        self._guard_count, self._guard_limit = inner.num_rendered, self.capacity
        shared_guard = self.rasterizer.overflow_guard() if hasattr(self.rasterizer, "overflow_guard") else None
        if shared_guard is not None:
            self._guard_count, self._guard_limit = shared_guard
        self.graph = None
        self.capture()
        self.regrowths += 1
        return max(lost, 1)

    def skipped_steps(self):
        """Number of replays whose optimiser step the device-side overflow guard skipped so far (sticky counter; synchronises)."""
        t = self.optimizer.skipped_steps
        return 0 if t is None else int(t.item())
