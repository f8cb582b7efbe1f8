"""Multi-tensor Adam in one HIP launch: a drop-in for the `torch.optim.Adam(l, lr=0.0, eps=1e-15)`
that GaussianModel builds [REF scene/gaussian_model.py:222-231] and the mapper steps [REF mp_Mapper.py:247].

State layout is torch's (`state[p] = {"step", "exp_avg", "exp_avg_sq"}`), because GaussianModel edits the optimiser state
in place when it appends or prunes Gaussians [REF scene/gaussian_model.py:409-492].  amsgrad / weight decay / maximize are
not supported (the reference does not use them).

``capturable=True`` (same meaning as torch.optim.Adam's flag) keeps the step count and the learning rates in device
memory, so that ``step()`` can be captured in a HIP graph and replayed (gs_icp_slam_amd/graph.py).  ``state[p]["step"]``
is then an int32 device tensor shared by the parameters that were first stepped together; after changing a group's
``lr`` call ``sync_lr()`` (outside capture) to push the new values to the device.
"""
import ctypes

import torch

from . import _lib

_MAX = 8


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.capturable = bool(capturable)
        self._lr_dev = {}       # (group index, position, betas, eps) bucket key -> [device double tensor, host lr values it holds]
        self._guard = None      # (count tensor uint32/int32[1], limit): skip the step when count > limit (capturable path)
        self.skipped_steps = None   # device int32[1], sticky: steps skipped by the guard since construction (read it whenever convenient)
        self._live_rows = None      # device int32[1]: parameters are capacity-backed, only this many leading rows are updated
        self._row_freeze = None     # (device int32[rows] mask, names of the param groups it applies to): rows with a non-zero word are left alone
        self._grad_rows = None      # device int32[rows] (the rasteriser's radii): rows with a word <= 0 take g = 0 and their gradient is not read
        self._prebumped = False     # the NEXT capturable step() finds its step counter already advanced (mapper_loss_and_grads(step_bump=...))

    def set_grad_row_mask(self, radii):
        """Capturable path, SPARSE gradients (round 6): `radii` = the int32 (P,) output of the rasteriser forward whose backward produced the gradients of
        the next step().  Rows with radii <= 0 are culled Gaussians: their gradient is zero by definition, the backward (GaussianRasterizationSettings.
        sparse_grads) did not write it and this step does not read it — the update is torch.optim.Adam's on g = 0.  Applies to every tensor with P rows;
        `None` removes it (every gradient element is read again).  Read at every step, so call it with the forward's radii before each step()."""
        if radii is not None and (not radii.is_cuda or radii.dtype != torch.int32 or not radii.is_contiguous()):
            raise RuntimeError("FusedAdam.set_grad_row_mask: expected the contiguous int32 device tensor `radii` of the rasteriser forward")
        self._grad_rows = radii

    def shared_step_tensor(self):
        """The ONE device step counter every parameter of this optimiser shares (capturable path, after the first step), or None when the parameters
        do not all share one — what mapper_loss_and_grads(step_bump=...) advances ahead of a step(step_already_bumped=True)."""
        steps = {}
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p, {})
                if not torch.is_tensor(st.get("step")):
                    return None
                steps[st["step"].data_ptr()] = st["step"]
        return next(iter(steps.values())) if len(steps) == 1 else None

    def set_row_freeze(self, mask, group_names):
        """Capturable path: the param groups named in `group_names` (their "name" key, as GaussianModel sets it [REF scene/gaussian_model.py:222-229])
        leave the rows whose `mask` word (DEVICE int32, one per row, read at every step) is non-zero untouched — parameter and both moments, as if
        those rows were no parameters.  `mask=None` removes it.  (gsicp_adam_step_masked; used by refglue's `freeze` policy.)"""
        if mask is None:
            self._row_freeze = None
            return
        if not mask.is_cuda or mask.dtype != torch.int32 or not mask.is_contiguous():
            raise RuntimeError("FusedAdam.set_row_freeze: expected a contiguous int32 device tensor (one word per row)")
        self._row_freeze = (mask, frozenset(group_names))

    def scoped_bindings(self, guard=None, live_rows=None):
        """Context manager: bind an overflow guard and / or a live-row count for the launches issued inside (a graph capture), then restore
        what was bound before — so that eager steps on the same optimiser afterwards are not silently gated by a graph's last replay."""
        opt = self

        class _Scope:
            def __enter__(self_):
                self_.prev = (opt._guard, opt._live_rows)
                if guard is not None:
                    opt.set_overflow_guard(*guard)
                if live_rows is not None:
                    opt.set_live_rows(live_rows)
                return opt

            def __exit__(self_, *exc):
                opt._guard, opt._live_rows = self_.prev
                return False
        return _Scope()

    def set_live_rows(self, n_dev):
        """Capturable path: the parameter tensors are the full-capacity buffers of a preallocated map (GaussianStore); update only the
        first n_dev[0] rows of each.  n_dev changes on the device (append / prune) without any pointer or launch changing."""
        if n_dev is not None and (not n_dev.is_cuda or n_dev.dtype != torch.int32 or n_dev.numel() != 1):
            raise RuntimeError("FusedAdam.set_live_rows: expected an int32[1] device tensor")
        self._live_rows = n_dev

    def set_overflow_guard(self, count, limit):
        """Capturable path: skip the whole step (parameters, moments and step count untouched) whenever `count` (a DEVICE int32[1],
        e.g. GaussianRasterizer.num_rendered) exceeds `limit` (the rasteriser's list capacity) at the time the step runs — the
        sync-free forward then rendered nothing and every gradient is zero.  `skipped_steps` counts such steps on the device."""
        if count is None:
            self._guard = None
            return
        if not count.is_cuda or count.numel() != 1 or count.element_size() != 4:
            raise RuntimeError("FusedAdam.set_overflow_guard: count must be a 4-byte integer device tensor with one element")
        self._guard = (count, int(limit))
        if self.skipped_steps is None or self.skipped_steps.device != count.device:
            self.skipped_steps = torch.zeros(1, dtype=torch.int32, device=count.device)

    def sync_lr(self):
        """Push the param groups' current learning rates to the device arrays of the capturable path."""
        for key, entry in self._lr_dev.items():
            new = [float(self.param_groups[gi]["lr"]) for gi, _pi in key[0]]
            if new != entry[1]:
                entry[0].copy_(torch.tensor(new, dtype=torch.float64))
                entry[1] = new

    def _step_capturable(self, lib, prebumped=False):
        # bucket = tensors that share (betas, eps, step counter); identified by (param-group index, position) pairs, which stay valid when
        # a map store re-binds fresh Parameter objects into the same groups [REF scene/gaussian_model.py:409-492] (ids of dead objects
        # can be reused by CPython, group positions cannot)
        buckets = {}
        fresh = {}
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            for pi, p in enumerate(group["params"]):
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("FusedAdam(capturable): parameters and gradients must be contiguous float32 on the HIP device")
                st = self.state[p]
                if len(st) == 0:
                    if p.device not in fresh:
                        fresh[p.device] = torch.zeros((), dtype=torch.int32, device=p.device)
                    st["step"] = fresh[p.device]
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                elif not torch.is_tensor(st["step"]):
                    st["step"] = torch.full((), int(st["step"]), dtype=torch.int32, device=p.device)
                buckets.setdefault((float(b1), float(b2), float(group["eps"]), st["step"].data_ptr()), []).append(
                    (p, p.grad, st["exp_avg"], st["exp_avg_sq"], float(group["lr"]), st["step"], (gi, pi)))
        capturing = torch.cuda.is_current_stream_capturing()
        # every distinct step tensor is bumped exactly once per step(): by the LAST launch that reads it
        last_of = {}
        launches = []
        for (b1, b2, eps, sp), items in buckets.items():
            for i in range(0, len(items), _MAX):
                launches.append((b1, b2, eps, sp, items[i:i + _MAX]))
                last_of[sp] = len(launches) - 1
        live_keys = set()
        for li, (b1, b2, eps, sp, items) in enumerate(launches):
            dev = items[0][0].device
            key = (tuple(t[6] for t in items), b1, b2, eps)
            live_keys.add(key)
            lrs = [t[4] for t in items]
            if key not in self._lr_dev:
                if capturing:
                    raise RuntimeError("FusedAdam(capturable): run one step() outside graph capture first (allocates the device lr array)")
                self._lr_dev[key] = [torch.tensor(lrs, dtype=torch.float64).to(dev), lrs]
            elif not capturing and self._lr_dev[key][1] != lrs:
                self._lr_dev[key][0].copy_(torch.tensor(lrs, dtype=torch.float64))
                self._lr_dev[key][1] = lrs
            lr_dev = self._lr_dev[key][0]
            guard_ptr, guard_lim, skip_ptr = None, 0, None
            if self._guard is not None and self._guard[0].device == dev:
                guard_ptr, guard_lim = ctypes.c_void_p(self._guard[0].data_ptr()), self._guard[1]
                skip_ptr = ctypes.c_void_p(self.skipped_steps.data_ptr())
            with torch.cuda.device(dev):
                stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                n = len(items)
                P = (ctypes.c_void_p * n)(*[t[0].data_ptr() for t in items])
                G = (ctypes.c_void_p * n)(*[t[1].data_ptr() for t in items])
                M = (ctypes.c_void_p * n)(*[t[2].data_ptr() for t in items])
                V = (ctypes.c_void_p * n)(*[t[3].data_ptr() for t in items])
                N = (ctypes.c_longlong * n)(*[t[0].numel() for t in items])
                live_ptr, RW, freeze_ptr, FR = None, None, None, None
                freeze = self._row_freeze if (self._row_freeze is not None and self._row_freeze[0].device == dev) else None
                rows = self._grad_rows if (self._grad_rows is not None and self._grad_rows.device == dev) else None
                if rows is not None and any(t[0].dim() == 0 or t[0].shape[0] != rows.numel() for t in items):
                    raise RuntimeError("FusedAdam.set_grad_row_mask: every parameter must have one row per entry of the mask")
                if (self._live_rows is not None and self._live_rows.device == dev) or freeze is not None or rows is not None:
                    RW = (ctypes.c_int * n)(*[(t[0].numel() // t[0].shape[0]) if t[0].dim() > 0 and t[0].shape[0] > 0 else 0 for t in items])
                if self._live_rows is not None and self._live_rows.device == dev:
                    live_ptr = ctypes.c_void_p(self._live_rows.data_ptr())
                if freeze is not None:
                    flags = [1 if self.param_groups[t[6][0]].get("name") in freeze[1] else 0 for t in items]
                    for t, fl in zip(items, flags):
                        if fl and t[0].shape[0] > freeze[0].numel():
                            raise RuntimeError("FusedAdam.set_row_freeze: the mask has fewer words than the tensor has rows")
                    if any(flags):
                        freeze_ptr, FR = ctypes.c_void_p(freeze[0].data_ptr()), (ctypes.c_int * n)(*flags)
                bump = 2 if prebumped else int(last_of[sp] == li)       # 2: the counter was advanced ahead of this launch, in stream order
                _lib.check(lib.gsicp_adam_step_sparse(n, P, G, M, V, N, ctypes.c_void_p(lr_dev.data_ptr()), b1, b2, eps,
                                                      ctypes.c_void_p(items[0][5].data_ptr()), bump, guard_ptr, guard_lim,
                                                      skip_ptr, live_ptr, RW, freeze_ptr, FR, ctypes.c_void_p(rows.data_ptr()) if rows is not None else None,
                                                      stream), "gsicp_adam_step_sparse")
        # Device lr arrays are NEVER freed: a captured MapperIterationGraph holds their addresses, and an eager step() that happens
        # to see fewer gradients (zero_grad(set_to_none=True), different chunking) must not hand that memory back to the caching allocator
        # while a graph may still replay.  One entry is <= 8 doubles; the number of distinct buckets an optimiser ever sees is a handful.
        del live_keys

    @torch.no_grad()
    def step(self, closure=None, step_already_bumped=False):
        """step_already_bumped (capturable path): the device step counter was advanced ahead of this call, in stream order, by
        mapper_loss_and_grads(step_bump=(self.shared_step_tensor(), ...)) — the update uses the counter as it is and no bump launch follows."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        if step_already_bumped and not self.capturable:
            raise RuntimeError("FusedAdam.step(step_already_bumped=True) needs capturable=True (a device step counter)")
        if self.capturable:
            self._step_capturable(lib, prebumped=bool(step_already_bumped))
            return loss
        buckets = {}   # (beta1, beta2, eps, step, device) -> list of (p, g, m, v, lr)
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam (gfx950): parameters must live on the HIP device; there is no CPU path")
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdam: parameters must be contiguous float32")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] = int(st["step"]) + 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                buckets.setdefault((float(b1), float(b2), float(group["eps"]), st["step"], p.device), []).append(
                    (p, g.to(torch.float32), st["exp_avg"], st["exp_avg_sq"], float(group["lr"])))
        for (b1, b2, eps, step, dev), items in buckets.items():
            with torch.cuda.device(dev):
                stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                for i in range(0, len(items), _MAX):
                    chunk = items[i:i + _MAX]
                    n = len(chunk)
                    P = (ctypes.c_void_p * n)(*[t[0].data_ptr() for t in chunk])
                    G = (ctypes.c_void_p * n)(*[t[1].data_ptr() for t in chunk])
                    M = (ctypes.c_void_p * n)(*[t[2].data_ptr() for t in chunk])
                    V = (ctypes.c_void_p * n)(*[t[3].data_ptr() for t in chunk])
                    N = (ctypes.c_longlong * n)(*[t[0].numel() for t in chunk])
                    LR = (ctypes.c_float * n)(*[t[4] for t in chunk])
                    _lib.check(lib.gsicp_adam_step(n, P, G, M, V, N, LR, b1, b2, eps, step, stream), "gsicp_adam_step")
        return loss
