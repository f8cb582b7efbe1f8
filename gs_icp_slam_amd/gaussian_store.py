"""Capacity-based storage for the map's Gaussians.

The reference grows and prunes the map by re-creating every parameter tensor and both Adam moments:
`densification_postfix` / `cat_tensors_to_optimizer` [REF scene/gaussian_model.py:448-492] run ~20 `torch.cat`s per keyframe and
`prune_points` / `_prune_optimizer` [REF scene/gaussian_model.py:409-447] ~20 boolean-index gathers, each allocating its result,
and every parameter changes address (which also invalidates a captured mapper graph's buffers).

`GaussianStore` preallocates two buffer sets of `capacity` rows for the six parameter tensors, their `exp_avg` / `exp_avg_sq` and
the per-Gaussian statistics.  The live tensors are views `[:n]` of the current set:

  * `append(...)` copies the new rows behind the live ones (moments zero, statistics reset as the reference does) — no allocation;
  * `prune(remove_mask)` moves the surviving rows of ALL arrays into the other set with one order-preserving stream compaction
    (`gsicp_store_compact`: three launches) and swaps the sets;
  * after either, the optimiser's param groups and state are re-bound to the new views exactly the way the reference re-keys them
    (same `state[p] = {"step", "exp_avg", "exp_avg_sq"}` layout), so `torch.optim.Adam` and `FusedAdam` both keep working.

The values equal the reference's cat / mask results bit for bit (tests/test_store_gpu.py).

`GaussianStore(stable=True)` additionally keeps every ADDRESS fixed for the lifetime of the store: `params` are Parameters over the
full-capacity buffers (never re-created), the optimiser's moments are the full-capacity moment buffers, and the live count lives in a
device int (`live_count`).  `append` writes rows in place and bumps the count; `prune` compacts into the second buffer set and copies the
survivors back.  Kernels that take `live_count` (rasteriser, activations, FusedAdam — include/gsicp_hip.h `live_rows_dev`) ignore the
rows behind it, so a captured mapper iteration (gs_icp_slam_amd/graph.py) survives keyframes and pruning without being re-captured.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib

PARAM_NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")


def rows_from_gicp(points, colors, rots, scales, z_values, trackable_idxs=None, max_sh_degree=0):
    """Parameter rows for new Gaussians from the tracker's outputs, initialised exactly as the reference does in
    create_from_pcd2_tensor / add_from_pcd2_tensor [REF scene/gaussian_model.py:134-180]: DC colour = RGB2SH(colour), higher SH
    coefficients zero, log-scales = log(scales / clamp_min(2 z^1.5, 1)), rotation copied, opacity = inverse_sigmoid(0.1).
    -> (dict for GaussianStore.append, trackable mask)."""
    n = points.shape[0]
    dev = points.device
    C0 = 0.28209479177387814
    n_coef = (int(max_sh_degree) + 1) ** 2
    features = torch.zeros((n, 3, n_coef), dtype=torch.float32, device=dev)
    features[:, :3, 0] = (colors - 0.5) / C0
    z = torch.clamp_min((z_values ** 1.5) * 2.0, 1.0).unsqueeze(-1).repeat(1, 3)
    opac = 0.1 * torch.ones((n, 1), dtype=torch.float, device=dev)
    rows = dict(xyz=points, f_dc=features[:, :, 0:1].transpose(1, 2).contiguous(), f_rest=features[:, :, 1:].transpose(1, 2).contiguous(),
                opacity=torch.log(opac / (1 - opac)), scaling=torch.log(scales / z), rotation=rots)
    mask = torch.zeros((n,), dtype=torch.bool, device=dev)
    if trackable_idxs is not None and len(trackable_idxs) != 0:
        mask[trackable_idxs] = True
    return rows, mask


class GaussianStore:
    def __init__(self, capacity, n_rest=0, device="cuda", stable=False):
        self.capacity, self.n_rest, self.device = int(capacity), int(n_rest), torch.device(device)
        self.stable = bool(stable)
        self.n = 0
        shapes = {"xyz": (3,), "f_dc": (1, 3), "f_rest": (self.n_rest, 3), "opacity": (1,), "scaling": (3,), "rotation": (4,)}
        self._shapes = shapes
        f32 = dict(dtype=torch.float32, device=self.device)

        def new_set():
            s = {}
            for k, sh in shapes.items():
                for role in ("p", "m", "v"):                       # parameter, exp_avg, exp_avg_sq
                    s[(role, k)] = torch.zeros((self.capacity,) + sh, **f32)
            s[("aux", "xyz_gradient_accum")] = torch.zeros((self.capacity, 1), **f32)
            s[("aux", "denom")] = torch.zeros((self.capacity, 1), **f32)
            s[("aux", "max_radii2D")] = torch.zeros((self.capacity,), **f32)
            s[("aux", "trackable_mask")] = torch.zeros((self.capacity,), dtype=torch.int32, device=self.device)   # 0 / 1
            s[("aux", "keyframe_idx")] = torch.zeros((self.capacity,), dtype=torch.int32, device=self.device)
            return s
        self._sets = [new_set(), new_set()]
        self._cur = 0
        self._n_dev = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._scratch = None
        self.optimizer = None
        self.params = {}

    # ------------------------------------------------------------------------------------------------ views
    def _buf(self, role, name):
        return self._sets[self._cur][(role, name)]

    def view(self, role, name):
        return self._buf(role, name)[: self.n]

    @property
    def trackable_mask(self):
        return self.view("aux", "trackable_mask").bool()

    @property
    def live_count(self):
        """int32[1] device tensor holding n (stable mode: what the kernels read instead of a host-side P)."""
        return self._n_dev

    def live(self, name):
        """Detached [:n] view of a parameter buffer, whatever the mode (for reading: hand-off to the tracker, statistics)."""
        return self._buf("p", name)[: self.n]

    def _rebind(self):
        """New Parameter views of the current set / length, re-keyed into the optimiser the way the reference does it
        [REF scene/gaussian_model.py:399-406, 415-424, 461-471]."""
        if self.stable:
            if not self.params:   # once: Parameters over the full-capacity buffers of set 0; nothing is ever re-created or re-keyed
                self.params = {k: nn.Parameter(self._sets[0][("p", k)], requires_grad=True) for k in PARAM_NAMES}
            return
        old = self.params
        self.params = {k: nn.Parameter(self.view("p", k), requires_grad=True) for k in PARAM_NAMES}
        if self.optimizer is None:
            return
        for group in self.optimizer.param_groups:
            name = group["name"]
            old_p = group["params"][0]
            st = self.optimizer.state.pop(old_p, None)
            new_p = self.params[name]
            group["params"][0] = new_p
            if st is not None:
                st["exp_avg"] = self.view("m", name)
                st["exp_avg_sq"] = self.view("v", name)
                self.optimizer.state[new_p] = st
        del old

    def attach(self, optimizer_cls, lrs, **kw):
        """Build the optimiser over the store's parameters: lrs maps the six names to learning rates
        [REF scene/gaussian_model.py:222-231]."""
        if not self.params:
            self._rebind()
        groups = [{"params": [self.params[k]], "lr": float(lrs[k]), "name": k} for k in PARAM_NAMES]
        self.optimizer = optimizer_cls(groups, **kw)
        # the moments live in the store from the first step on (a lazily created zeros_like() would sit outside it)
        fused = hasattr(self.optimizer, "capturable") and type(self.optimizer).__name__ == "FusedAdam"
        shared_step = torch.zeros((), dtype=torch.int32, device=self.device) if (fused and self.optimizer.capturable) else None
        for k in PARAM_NAMES:
            if fused:
                step = shared_step if shared_step is not None else 0
            else:
                step = torch.tensor(0.0)                           # torch.optim.Adam keeps a float32 CPU scalar
            m, v = (self._sets[0][("m", k)], self._sets[0][("v", k)]) if self.stable else (self.view("m", k), self.view("v", k))
            self.optimizer.state[self.params[k]] = {"step": step, "exp_avg": m, "exp_avg_sq": v}
        if self.stable and hasattr(self.optimizer, "set_live_rows"):
            self.optimizer.set_live_rows(self._n_dev)
        return self.optimizer

    # ------------------------------------------------------------------------------------------------ growth
    def append(self, new, trackable_mask=None, keyframe_idx=None):
        """new: dict name -> (k, ...) tensors.  Equivalent of densification_postfix [REF scene/gaussian_model.py:474-492]:
        parameters concatenated, their moments extended with zeros, xyz_gradient_accum / denom / max_radii2D reset to zero for ALL
        Gaussians, trackable mask (and keyframe index) concatenated."""
        k = int(new["xyz"].shape[0])
        if self.n + k > self.capacity:
            raise RuntimeError(f"GaussianStore: capacity {self.capacity} exceeded ({self.n} + {k})")
        lo, hi = self.n, self.n + k
        with torch.no_grad():
            for name in PARAM_NAMES:
                self._buf("p", name)[lo:hi].copy_(new[name].reshape((k,) + self._shapes[name]))
                self._buf("m", name)[lo:hi].zero_()
                self._buf("v", name)[lo:hi].zero_()
            for aux in ("xyz_gradient_accum", "denom", "max_radii2D"):
                self._buf("aux", aux)[:hi].zero_()
            tm = self._buf("aux", "trackable_mask")[lo:hi]
            if trackable_mask is None:
                tm.fill_(1)
            else:
                tm.copy_(trackable_mask.to(torch.int32))
            if keyframe_idx is not None:
                self._buf("aux", "keyframe_idx")[lo:hi].copy_(keyframe_idx.reshape(-1).to(torch.int32))
        self.n = hi
        self._n_dev.fill_(hi)          # a launch with the scalar in its arguments: no host synchronisation
        self._rebind()
        return self.params

    def grow(self, new_capacity):
        """Re-house the map in buffers of `new_capacity` rows (the reference's map grows without bound [REF scene/gaussian_model.py:474-492]; its shared
        buffers are sized for 10 M points [REF gs_icp_slam.py:86]).  Every live row of every array — parameters, both Adam moments, statistics, masks —
        is copied; the Parameter OBJECTS and the optimiser's state entries stay (their storage is swapped), so param groups, learning rates and the step
        count carry over.  Every device ADDRESS changes: a hipGraph captured over the old buffers is void — the caller drops it and captures again
        (refglue.add_from_pcd2_tensor does) — and a row-freeze mask bound to the old trackable mask must be bound again (returned flag).  Synchronises."""
        new_capacity = int(new_capacity)
        if new_capacity <= self.capacity:
            return False
        torch.cuda.synchronize(self.device)
        old_sets, old_cur, n = self._sets, self._cur, self.n
        self.capacity = new_capacity
        f32 = dict(dtype=torch.float32, device=self.device)
        new_sets = []
        for _ in range(2):
            s = {}
            for key, t in old_sets[0].items():
                s[key] = torch.zeros((new_capacity,) + tuple(t.shape[1:]), dtype=t.dtype, device=self.device)
            new_sets.append(s)
        with torch.no_grad():
            src = old_sets[old_cur]
            for key, t in src.items():
                if n > 0 and t[0].numel() > 0:
                    new_sets[0][key][:n].copy_(t[:n])
        self._sets, self._cur = new_sets, 0
        if self.params:
            if self.stable:
                for k in PARAM_NAMES:
                    self.params[k].data = self._sets[0][("p", k)]
                    if self.optimizer is not None and self.params[k] in self.optimizer.state:
                        st = self.optimizer.state[self.params[k]]
                        st["exp_avg"], st["exp_avg_sq"] = self._sets[0][("m", k)], self._sets[0][("v", k)]
            else:
                self._rebind()
        del old_sets, f32
        torch.cuda.synchronize(self.device)
        return True

    # ------------------------------------------------------------------------------------------------ pruning
    def prune(self, remove_mask):
        """Equivalent of prune_points(mask) [REF scene/gaussian_model.py:426-447]: rows with remove_mask True disappear from every
        parameter, both moments and all statistics; order preserved.  Synchronises once (to learn the new count)."""
        lib = _lib.load()
        if self.n == 0:
            return self.params
        n_old = self.n
        keep = (~remove_mask.reshape(-1)[: self.n].to(device=self.device, dtype=torch.bool)).contiguous().view(torch.uint8)
        src, dst = self._sets[self._cur], self._sets[self._cur ^ 1]
        keys = list(src.keys())
        n_arr = len(keys)
        S = (ctypes.c_void_p * n_arr)(*[src[k].data_ptr() for k in keys])
        D = (ctypes.c_void_p * n_arr)(*[dst[k].data_ptr() for k in keys])
        RB = (ctypes.c_int * n_arr)(*[max(int(src[k][0].numel()), 0) * src[k].element_size() for k in keys])
        # zero-width arrays (f_rest with no coefficients) carry nothing: leave them out
        live = [i for i in range(n_arr) if RB[i] > 0]
        S = (ctypes.c_void_p * len(live))(*[S[i] for i in live])
        D = (ctypes.c_void_p * len(live))(*[D[i] for i in live])
        RB = (ctypes.c_int * len(live))(*[RB[i] for i in live])
        need = int(lib.gsicp_store_compact_scratch_bytes(self.n))
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(max(need, 4096), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(lib.gsicp_store_compact(self.n, ctypes.c_void_p(keep.data_ptr()), len(live), S, D, RB,
                                               ctypes.c_void_p(self._scratch.data_ptr()), ctypes.c_void_p(self._n_dev.data_ptr()), stream),
                       "gsicp_store_compact")
        self.n = int(self._n_dev.item())
        if self.stable:
            # addresses must not move: the survivors go back into set 0 (the compaction cannot run in place: a parallel scatter would
            # overwrite rows other threads have not read yet).  Rows behind the new count keep stale values; nothing reads them, and append()
            # rewrites parameters and zeroes moments of the rows it claims.
            with torch.no_grad():
                for k in keys:
                    if src[k][0].numel() > 0 and self.n > 0:
                        src[k][: self.n].copy_(dst[k][: self.n])
                # the freed tail [n_new, n_old) still holds copies of old Gaussians; a consumer that ignores `live_count` (torch.optim.Adam,
                # a custom rasteriser factory, a checkpoint written from `params` instead of `live()`) would render / update them.  Make them
                # inert: opacity far below every threshold (sigmoid(-30) = 1e-13 < 1/255: culled by the rasteriser), moments zero.
                if n_old > self.n:
                    src[("p", "opacity")][self.n:n_old].fill_(-30.0)
                    for name in PARAM_NAMES:
                        if src[("m", name)][0].numel() > 0:
                            src[("m", name)][self.n:n_old].zero_()
                            src[("v", name)][self.n:n_old].zero_()
            return self.params
        self._cur ^= 1
        self._rebind()
        return self.params
