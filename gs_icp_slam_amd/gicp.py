"""Host-side mirror of the reference's ``pygicp`` extension module, backed by libgsicp_hip.so.

``FastGICP`` keeps the method names, argument meaning, host-numpy-in / host-numpy-out convention and the
RuntimeError-on-failure behaviour of the pybind11 class the tracker drives [REF mp_Tracker.py:53, 109-110, 157-169,
191-200, 231, 256-264, 287-288].  Device residency is internal: every call uploads / downloads through the C ABI,
and all kernels run on the object's own HIP stream.
"""
import ctypes

import numpy as np

from . import _lib

_REG = {"NONE": 0, "MIN_EIG": 1, "NORMALIZED_MIN_EIG": 2, "PLANE": 3, "FROBENIUS": 4}


def _vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class FastGICP:
    def __init__(self):
        self._lib = _lib.load()
        h = self._lib.gsicp_gicp_create()
        if not h:
            raise RuntimeError("pygicp.FastGICP (gfx950): " + _lib.last_error())
        self._h = ctypes.c_void_p(h)

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
        self._ck(self._lib.gsicp_gicp_set_max_correspondence_distance(self._h, float(d)), "set_max_correspondence_distance")

    def set_max_knn_distance(self, d):
        self._ck(self._lib.gsicp_gicp_set_max_knn_distance(self._h, float(d)), "set_max_knn_distance")

    def set_correspondence_randomness(self, k):
        self._ck(self._lib.gsicp_gicp_set_correspondence_randomness(self._h, int(k)), "set_correspondence_randomness")

    def set_max_iterations(self, n):
        self._ck(self._lib.gsicp_gicp_set_max_iterations(self._h, int(n)), "set_max_iterations")

    def set_num_threads(self, n):
        self._ck(self._lib.gsicp_gicp_set_num_threads(self._h, int(n)), "set_num_threads")

    def set_regularization_method(self, method):
        m = _REG[method.upper()] if isinstance(method, str) else int(method)
        self._ck(self._lib.gsicp_gicp_set_regularization_method(self._h, m), "set_regularization_method")

    def set_rotation_epsilon(self, e):
        self._ck(self._lib.gsicp_gicp_set_rotation_epsilon(self._h, float(e)), "set_rotation_epsilon")

    def set_transformation_epsilon(self, e):
        self._ck(self._lib.gsicp_gicp_set_transformation_epsilon(self._h, float(e)), "set_transformation_epsilon")

    # ---- clouds
    @staticmethod
    def _points(pts):
        pts = np.asarray(pts)
        if pts.ndim != 2 or pts.shape[1] != 3:
            raise RuntimeError("pygicp.FastGICP: points must have shape (N, 3)")
        f64 = pts.dtype == np.float64
        return np.ascontiguousarray(pts, dtype=np.float64 if f64 else np.float32), int(f64)

    def set_input_target(self, points):
        p, f64 = self._points(points)
        self._ck(self._lib.gsicp_gicp_set_input_target(self._h, _vp(p), p.shape[0], f64), "set_input_target")

    def set_input_source(self, points):
        p, f64 = self._points(points)
        self._ck(self._lib.gsicp_gicp_set_input_source(self._h, _vp(p), p.shape[0], f64), "set_input_source")

    def set_target_filter(self, num_trackable, input_filter):
        f = np.ascontiguousarray(input_filter, dtype=np.int32).ravel()
        self._ck(self._lib.gsicp_gicp_set_target_filter(self._h, int(num_trackable), _vp(f), f.shape[0]), "set_target_filter")

    def set_source_filter(self, num_trackable, input_filter):
        f = np.ascontiguousarray(input_filter, dtype=np.int32).ravel()
        self._ck(self._lib.gsicp_gicp_set_source_filter(self._h, int(num_trackable), _vp(f), f.shape[0]), "set_source_filter")

    # ---- covariances
    def calculate_target_covariance_with_filter(self):
        self._ck(self._lib.gsicp_gicp_calculate_target_covariance_with_filter(self._h), "calculate_target_covariance_with_filter")

    def calculate_source_covariance(self):
        self._ck(self._lib.gsicp_gicp_calculate_source_covariance(self._h), "calculate_source_covariance")

    def _fetch(self, fn, n, width, what):
        out = np.empty(n * width, np.float32)
        got = self._ck(fn(self._h, _vp(out), n), what)
        return out[: got * width]

    def get_target_rotationsq(self):
        return self._fetch(self._lib.gsicp_gicp_get_target_rotationsq, self._lib.gsicp_gicp_num_target(self._h), 4, "get_target_rotationsq")

    def get_target_scales(self):
        return self._fetch(self._lib.gsicp_gicp_get_target_scales, self._lib.gsicp_gicp_num_target(self._h), 3, "get_target_scales")

    def get_source_rotationsq(self):
        return self._fetch(self._lib.gsicp_gicp_get_source_rotationsq, self._lib.gsicp_gicp_num_source(self._h), 4, "get_source_rotationsq")

    def get_source_scales(self):
        return self._fetch(self._lib.gsicp_gicp_get_source_scales, self._lib.gsicp_gicp_num_source(self._h), 3, "get_source_scales")

    def set_target_covariances_fromqs(self, rotations_flat, scales_flat):
        r = np.ascontiguousarray(rotations_flat, dtype=np.float32).ravel()
        s = np.ascontiguousarray(scales_flat, dtype=np.float32).ravel()
        self._ck(self._lib.gsicp_gicp_set_target_covariances_fromqs(self._h, _vp(r), r.size, _vp(s), s.size),
                 "set_target_covariances_fromqs")

    # ---- registration
    def align(self, initial_guess=None):
        init = np.eye(4) if initial_guess is None else np.asarray(initial_guess)
        if init.shape != (4, 4):
            raise RuntimeError("pygicp.FastGICP.align: initial guess must be 4x4")
        init = np.ascontiguousarray(init, dtype=np.float64)
        out = np.empty((4, 4), np.float64)
        self.iterations = self._ck(self._lib.gsicp_gicp_align(self._h, _vp(init), _vp(out)), "align")
        return out.astype(np.float32)   # the reference binding returns an Eigen::Matrix4f

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

    def last_align_stats(self):
        out = np.empty(6, np.float64)
        self._lib.gsicp_gicp_last_align_stats(self._h, _vp(out))
        return dict(launches=int(out[0]), lm_trials=int(out[1]), cost=float(out[2]), converged=bool(out[3]), device_us=float(out[4]),
                    failed=bool(out[5]), iterations=getattr(self, "iterations", 0))
