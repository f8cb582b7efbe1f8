"""Host-side mirror of the reference's ``pygicp`` extension module, backed by libgsicp_hip.so.

``FastGICP`` keeps the method names, argument meaning, host-numpy-in / host-numpy-out convention and the
RuntimeError-on-failure behaviour of the pybind11 class the tracker drives [REF mp_Tracker.py:53, 109-110, 157-169,
191-200, 231, 256-264, 287-288].  Device residency is internal: every call uploads / downloads through the C ABI,
and all kernels run on the object's own HIP stream.

Additive device-pointer overloads: `set_input_target`, `set_input_source` and
`set_target_covariances_fromqs` also accept torch tensors on the HIP device (no host round trip; the tracker's stream is ordered
after torch's current stream), `set_target_from_gaussians(...)` replaces the whole keyframe hand-off
`get_trackable_gaussians_tensor -> .cpu() -> numpy -> set_input_target + set_target_covariances_fromqs`
[REF scene/gaussian_model.py:207-215; mp_Tracker.py:284-289], and `get_source_rotationsq_tensor` / `get_source_scales_tensor`
return device tensors.  The numpy behaviour is unchanged.
"""
import ctypes

import numpy as np

from . import _lib

_REG = {"NONE": 0, "MIN_EIG": 1, "NORMALIZED_MIN_EIG": 2, "PLANE": 3, "FROBENIUS": 4}


def _vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _is_device_tensor(x):
    return type(x).__module__.startswith("torch") and hasattr(x, "is_cuda") and x.is_cuda


def _dev_f32(t, shape_last=None):
    import torch
    t = t.detach()
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.to(torch.float32).contiguous()
    if shape_last is not None and (t.dim() != 2 or t.shape[1] != shape_last):
        raise RuntimeError(f"pygicp.FastGICP: expected a (N, {shape_last}) tensor")
    return t


def _cur_stream(t):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _rebuild(config):
    reg = FastGICP()
    for name, value in config.items():
        getattr(reg, name)(value)
    return reg


class FastGICP:
    def __init__(self):
        self._lib = _lib.load()
        h = self._lib.gsicp_gicp_create()
        if not h:
            raise RuntimeError("pygicp.FastGICP (gfx950): " + _lib.last_error())
        self._h = ctypes.c_void_p(h)
        self._live = []   # device tensors handed over with wait=0: kept alive until the next synchronous call has returned
        self._config = {}  # setter name -> last value (what __reduce__ replays)

    def __reduce__(self):
        """The reference builds `pygicp.FastGICP()` in the parent and ships the whole Tracker object to the spawned tracking process
        [REF mp_Tracker.py:53; gs_icp_slam.py:121-127], so the object must pickle.  Device state is per process: the copy is a fresh
        registration object with the same configuration (clouds are always set inside the tracking process, after the spawn)."""
        return (_rebuild, (dict(self._config),))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.gsicp_gicp_destroy(h)
            except Exception:
                pass

    def _ck(self, rc, what):
        return _lib.check(rc, "pygicp.FastGICP." + what)

    # ---- configuration
    def set_max_correspondence_distance(self, d):
        self._config["set_max_correspondence_distance"] = d
        self._ck(self._lib.gsicp_gicp_set_max_correspondence_distance(self._h, float(d)), "set_max_correspondence_distance")

    def set_max_knn_distance(self, d):
        self._config["set_max_knn_distance"] = d
        self._ck(self._lib.gsicp_gicp_set_max_knn_distance(self._h, float(d)), "set_max_knn_distance")

    def set_correspondence_randomness(self, k):
        self._config["set_correspondence_randomness"] = k
        self._ck(self._lib.gsicp_gicp_set_correspondence_randomness(self._h, int(k)), "set_correspondence_randomness")

    def set_max_iterations(self, n):
        self._config["set_max_iterations"] = n
        self._ck(self._lib.gsicp_gicp_set_max_iterations(self._h, int(n)), "set_max_iterations")

    def set_num_threads(self, n):
        self._config["set_num_threads"] = n
        self._ck(self._lib.gsicp_gicp_set_num_threads(self._h, int(n)), "set_num_threads")

    def set_regularization_method(self, method):
        self._config["set_regularization_method"] = method
        m = _REG[method.upper()] if isinstance(method, str) else int(method)
        self._ck(self._lib.gsicp_gicp_set_regularization_method(self._h, m), "set_regularization_method")

    def set_scale_semantics(self, mode):
        """Extension (SURVEY 8a unknown): "stddev" (default) — get_*_scales() are sqrt(eigenvalues) and set_target_covariances_fromqs
        squares its scales; "variance" — eigenvalues are exported / consumed as they are."""
        self._config["set_scale_semantics"] = mode
        m = {"stddev": 0, "variance": 1}[mode] if isinstance(mode, str) else int(mode)
        self._ck(self._lib.gsicp_gicp_set_scale_semantics(self._h, m), "set_scale_semantics")

    def set_rotation_epsilon(self, e):
        self._config["set_rotation_epsilon"] = e
        self._ck(self._lib.gsicp_gicp_set_rotation_epsilon(self._h, float(e)), "set_rotation_epsilon")

    def set_transformation_epsilon(self, e):
        self._config["set_transformation_epsilon"] = e
        self._ck(self._lib.gsicp_gicp_set_transformation_epsilon(self._h, float(e)), "set_transformation_epsilon")

    # ---- clouds
    @staticmethod
    def _points(pts):
        pts = np.asarray(pts)
        if pts.ndim != 2 or pts.shape[1] != 3:
            raise RuntimeError("pygicp.FastGICP: points must have shape (N, 3)")
        f64 = pts.dtype == np.float64
        return np.ascontiguousarray(pts, dtype=np.float64 if f64 else np.float32), int(f64)

    @_lib.traced("gicp.set_input_target")
    def set_input_target(self, points):
        if _is_device_tensor(points):
            t = _dev_f32(points, 3)
            self._live.append(t)
            self._ck(self._lib.gsicp_gicp_set_input_target_device(self._h, ctypes.c_void_p(t.data_ptr()), t.shape[0], _cur_stream(t), 0),
                     "set_input_target")
            return
        p, f64 = self._points(points)
        self._ck(self._lib.gsicp_gicp_set_input_target(self._h, _vp(p), p.shape[0], f64), "set_input_target")

    @_lib.traced("gicp.set_input_source")
    def set_input_source(self, points):
        if _is_device_tensor(points):
            t = _dev_f32(points, 3)
            self._live.append(t)
            self._ck(self._lib.gsicp_gicp_set_input_source_device(self._h, ctypes.c_void_p(t.data_ptr()), t.shape[0], _cur_stream(t), 0),
                     "set_input_source")
            return
        p, f64 = self._points(points)
        self._ck(self._lib.gsicp_gicp_set_input_source(self._h, _vp(p), p.shape[0], f64), "set_input_source")

    def set_source_trackable(self, trackable_idx):
        """Device-side `set_source_filter`: trackable_idx[r] = index of the r-th trackable source point (int32 device tensor, e.g.
        DepthFrontEnd.make_pointcloud(...).trackable_idx)."""
        import torch
        t = trackable_idx.detach()
        if not t.is_cuda:
            raise RuntimeError("pygicp.FastGICP.set_source_trackable: expected a device tensor (use set_source_filter for numpy)")
        if t.dtype != torch.int32 or not t.is_contiguous():
            t = t.to(torch.int32).contiguous()
        self._live.append(t)
        self._ck(self._lib.gsicp_gicp_set_source_track_device(self._h, ctypes.c_void_p(t.data_ptr()), t.numel(), _cur_stream(t), 0),
                 "set_source_trackable")

    def set_target_from_gaussians(self, xyz, rotation, scaling, opacity, trackable_mask=None, opacity_th=0.0):
        """Device-side keyframe hand-off: the Gaussians with opacity > opacity_th (and trackable_mask set) become the target cloud,
        in index order, with covariances from their (activated) rotations and scales.  Returns the number of target points."""
        import torch
        xyz, rotation, scaling = _dev_f32(xyz, 3), _dev_f32(rotation, 4), _dev_f32(scaling, 3)
        opacity = _dev_f32(opacity.reshape(-1, 1), 1)
        P = xyz.shape[0]
        if rotation.shape[0] != P or scaling.shape[0] != P or opacity.shape[0] != P:
            raise RuntimeError("pygicp.FastGICP.set_target_from_gaussians: tensor sizes differ")
        m = None
        if trackable_mask is not None:
            m = trackable_mask.detach().to(device=xyz.device, dtype=torch.bool).contiguous().view(torch.uint8)
            if m.numel() != P:
                raise RuntimeError("pygicp.FastGICP.set_target_from_gaussians: mask size differs")
        n = self._ck(self._lib.gsicp_gicp_set_target_from_gaussians_device(
            self._h, P, ctypes.c_void_p(xyz.data_ptr()), ctypes.c_void_p(rotation.data_ptr()), ctypes.c_void_p(scaling.data_ptr()),
            ctypes.c_void_p(opacity.data_ptr()), None if m is None else ctypes.c_void_p(m.data_ptr()), float(opacity_th), _cur_stream(xyz)),
            "set_target_from_gaussians")
        self._live.clear()
        return n

    @_lib.traced("gicp.set_target_filter")
    def set_target_filter(self, num_trackable, input_filter):
        f = np.ascontiguousarray(input_filter, dtype=np.int32).ravel()
        self._ck(self._lib.gsicp_gicp_set_target_filter(self._h, int(num_trackable), _vp(f), f.shape[0]), "set_target_filter")

    @_lib.traced("gicp.set_source_filter")
    def set_source_filter(self, num_trackable, input_filter):
        f = np.ascontiguousarray(input_filter, dtype=np.int32).ravel()
        self._ck(self._lib.gsicp_gicp_set_source_filter(self._h, int(num_trackable), _vp(f), f.shape[0]), "set_source_filter")

    # ---- covariances
    @_lib.traced("gicp.calculate_target_covariance_with_filter")
    def calculate_target_covariance_with_filter(self):
        self._ck(self._lib.gsicp_gicp_calculate_target_covariance_with_filter(self._h), "calculate_target_covariance_with_filter")

    def calculate_source_covariance(self):
        self._ck(self._lib.gsicp_gicp_calculate_source_covariance(self._h), "calculate_source_covariance")

    def _fetch(self, fn, n, width, what):
        out = np.empty(n * width, np.float32)
        got = self._ck(fn(self._h, _vp(out), n), what)
        return out[: got * width]

    @_lib.traced("gicp.get_target_rotationsq")
    def get_target_rotationsq(self):
        return self._fetch(self._lib.gsicp_gicp_get_target_rotationsq, self._lib.gsicp_gicp_num_target(self._h), 4, "get_target_rotationsq")

    @_lib.traced("gicp.get_target_scales")
    def get_target_scales(self):
        return self._fetch(self._lib.gsicp_gicp_get_target_scales, self._lib.gsicp_gicp_num_target(self._h), 3, "get_target_scales")

    @_lib.traced("gicp.get_source_rotationsq")
    def get_source_rotationsq(self):
        return self._fetch(self._lib.gsicp_gicp_get_source_rotationsq, self._lib.gsicp_gicp_num_source(self._h), 4, "get_source_rotationsq")

    @_lib.traced("gicp.get_source_scales")
    def get_source_scales(self):
        return self._fetch(self._lib.gsicp_gicp_get_source_scales, self._lib.gsicp_gicp_num_source(self._h), 3, "get_source_scales")

    def _fetch_tensor(self, fn, width, what, device=None):
        import torch
        n = self._lib.gsicp_gicp_num_source(self._h)
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        out = torch.empty((n, width), dtype=torch.float32, device=dev)
        got = self._ck(fn(self._h, ctypes.c_void_p(out.data_ptr()), n, _cur_stream(out)), what)
        return out[:got]

    def get_source_rotationsq_tensor(self, device=None):
        """(n,4) xyzw quaternions of the source covariances as a device tensor (torch's current stream is ordered after the copy)."""
        return self._fetch_tensor(self._lib.gsicp_gicp_get_source_rotationsq_device, 4, "get_source_rotationsq_tensor", device)

    def get_source_scales_tensor(self, device=None):
        return self._fetch_tensor(self._lib.gsicp_gicp_get_source_scales_device, 3, "get_source_scales_tensor", device)

    @_lib.traced("gicp.set_target_covariances_fromqs")
    def set_target_covariances_fromqs(self, rotations_flat, scales_flat):
        if _is_device_tensor(rotations_flat) and _is_device_tensor(scales_flat):
            r, sc = _dev_f32(rotations_flat).reshape(-1), _dev_f32(scales_flat).reshape(-1)
            self._live += [r, sc]
            self._ck(self._lib.gsicp_gicp_set_target_covariances_fromqs_device(
                self._h, ctypes.c_void_p(r.data_ptr()), r.numel(), ctypes.c_void_p(sc.data_ptr()), sc.numel(), _cur_stream(r), 0),
                "set_target_covariances_fromqs")
            return
        r = np.ascontiguousarray(rotations_flat, dtype=np.float32).ravel()
        s = np.ascontiguousarray(scales_flat, dtype=np.float32).ravel()
        self._ck(self._lib.gsicp_gicp_set_target_covariances_fromqs(self._h, _vp(r), r.size, _vp(s), s.size),
                 "set_target_covariances_fromqs")

    # ---- registration
    @_lib.traced("gicp.align")
    def align(self, initial_guess=None):
        init = np.eye(4) if initial_guess is None else np.asarray(initial_guess)
        if init.shape != (4, 4):
            raise RuntimeError("pygicp.FastGICP.align: initial guess must be 4x4")
        init = np.ascontiguousarray(init, dtype=np.float64)
        out = np.empty((4, 4), np.float64)
        self.iterations = self._ck(self._lib.gsicp_gicp_align(self._h, _vp(init), _vp(out)), "align")
        self._live.clear()   # everything enqueued before the align kernel has completed
        return out.astype(np.float32)   # the reference binding returns an Eigen::Matrix4f

    @_lib.traced("gicp.get_source_correspondence")
    def get_source_correspondence(self):
        n = self._lib.gsicp_gicp_num_source(self._h)
        idx = np.empty(n, np.int32)
        d2 = np.empty(n, np.float32)
        got = self._ck(self._lib.gsicp_gicp_get_source_correspondence(self._h, _vp(idx), _vp(d2), n), "get_source_correspondence")
        return idx[:got], d2[:got]

    def get_final_hessian(self):
        out = np.empty((6, 6), np.float64)
        self._ck(self._lib.gsicp_gicp_get_final_hessian(self._h, _vp(out)), "get_final_hessian")
        return out

    def knn_stats(self):
        out = np.empty(12, np.float64)
        self._ck(self._lib.gsicp_gicp_knn_stats(self._h, _vp(out)), "knn_stats")
        return dict(cell=float(out[0]), dims=(int(out[1]), int(out[2]), int(out[3])), whole_grid=int(out[5]), ring1=int(out[6]),
                    ring2=int(out[7]), ring3=int(out[8]), full_scan=int(out[9]))

    def target_index_stats(self):
        """Sizes of the target search structure as of the last build (diagnostics for map-sized targets)."""
        out = np.empty(12, np.float64)
        self._ck(self._lib.gsicp_gicp_target_index_stats(self._h, _vp(out)), "target_index_stats")
        levels = [dict(table_slots=int(o[0]), bytes=int(o[1]), cell_m=float(o[2]), occupied_cells=int(o[3]), radius_m=float(o[4]))
                  for o in (out[2:7], out[7:12]) if o[0] > 0]
        return dict(targets=int(out[0]), hashed_grid=bool(out[1]), levels=levels)

    def last_align_stats(self):
        out = np.empty(6, np.float64)
        self._lib.gsicp_gicp_last_align_stats(self._h, _vp(out))
        return dict(launches=int(out[0]), lm_trials=int(out[1]), cost=float(out[2]), converged=bool(out[3]), device_us=float(out[4]),
                    failed=bool(out[5]), iterations=getattr(self, "iterations", 0), barrier_retries=int(self._lib.gsicp_gicp_barrier_retries(self._h)))

    def _debug_abort_next_align(self):
        """Test hook: the next align()'s first grid barrier aborts, exercising the single-workgroup recovery path."""
        self._lib.gsicp_gicp_debug_abort_next_align(self._h)
