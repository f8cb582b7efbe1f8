"""ORACLE — TEST INFRASTRUCTURE ONLY (numpy/ctypes front-end of the CPU restatements in this directory).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  The product path (``gs_icp_slam_amd``) never does; it fails loudly when the HIP library is absent.

PARITY UNPINNED: the reference's native submodules are empty directories and it has no tests or golden
vectors, so these restatements follow the reference's call sites plus the published
upstream algorithms; see the header of each ``*_oracle.cpp``.  Pinned exceptions — pieces the reference restates in Python that
is present, checked against golden vectors produced by RUNNING that Python (tests/golden/make_golden_utils.py,
tests/test_oracle_pinned.py): the (x,y,z,w) quaternion convention and covariance assembly (utils/general_utils.py), SH evaluation
degrees 0-3 (utils/sh_utils.py), camera matrices (utils/graphics_utils.py + scene/shared_objs.py), and the loss (loss_oracle.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(force=False):
    """Compile the oracle shared libraries with g++ (no GPU needed)."""
    srcs = [f for f in os.listdir(_HERE) if f.endswith("_oracle.cpp")]
    for src in srcs:
        name = src[: -len("_oracle.cpp")]
        out = os.path.join(_HERE, f"liboracle_{name}.so")
        src_p = os.path.join(_HERE, src)
        if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src_p):
            subprocess.check_call(["make", "-C", _HERE, f"liboracle_{name}.so"], stdout=subprocess.DEVNULL)


def _lib(name):
    if name not in _LIBS:
        path = os.path.join(_HERE, f"liboracle_{name}.so")
        src = os.path.join(_HERE, f"{name}_oracle.cpp")
        if not os.path.exists(path) or (os.path.exists(src) and os.path.getmtime(path) < os.path.getmtime(src)):
            build()
        _LIBS[name] = ctypes.CDLL(path)
    return _LIBS[name]


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _arr(a, dt, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a, dtype=dt))
    if shape is not None:
        a = a.reshape(shape)
    return a


# ------------------------------------------------------------------------------------------ rasteriser
def raster_set_depth_mode(mode):
    """0 = sum z alpha T (default), 1 = alpha-normalised (divided by 1 - T_final).  Process-global; restore 0 after use."""
    _lib("raster").oracle_raster_set_depth_mode(int(mode))


def raster_forward(means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy, W, H, bg,
                   shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                   scale_modifier=1.0, sh_degree=0, prefiltered=False, dtype=np.float32):
    """Returns a dict with colour (3,H,W), depth (H,W), radii, is_used, geometry, sorted lists, ranges,
    final_T, n_contrib and the per-pixel decision margin (min relative distance of any threshold test)."""
    lib = _lib("raster")
    dt = np.dtype(dtype)
    R = ctypes.c_float if dt == np.float32 else ctypes.c_double
    fn = lib.oracle_raster_forward_f32 if dt == np.float32 else lib.oracle_raster_forward_f64
    fn.restype = ctypes.c_int
    means3D = _arr(means3D, dt, (-1, 3))
    P = means3D.shape[0]
    shs = _arr(shs, dt)
    M = 0 if shs is None else (shs.reshape(P, -1, 3).shape[1] if P > 0 else max(1, int(np.prod(shs.shape[1:])) // 3))
    colors_precomp = _arr(colors_precomp, dt)
    opacities = _arr(opacities, dt, (-1,))
    scales = _arr(scales, dt)
    rotations = _arr(rotations, dt)
    cov3D_precomp = _arr(cov3D_precomp, dt)
    view = _arr(viewmatrix, dt, (16,))
    proj = _arr(projmatrix, dt, (16,))
    campos = _arr(campos, dt, (3,))
    bg = _arr(bg, dt, (3,))
    HW = W * H
    T = ((W + 15) // 16) * ((H + 15) // 16)
    color = np.zeros((3, H, W), dt)
    depth = np.zeros((H, W), dt)
    radii = np.zeros(P, np.int32)
    is_used = np.zeros(P, np.int32)
    geom = np.zeros((P, 12), dt)
    ranges = np.zeros((T, 2), np.uint32)
    final_T = np.zeros((H, W), dt)
    n_contrib = np.zeros((H, W), np.uint32)
    margin = np.zeros((H, W), dt)
    cap = 1 << 16
    while True:
        keys = np.zeros(cap, np.uint64)
        vals = np.zeros(cap, np.uint32)
        n = fn(P, int(sh_degree), M, _p(bg), W, H, _p(means3D), _p(shs), _p(colors_precomp), _p(opacities), _p(scales),
               R(scale_modifier), _p(rotations), _p(cov3D_precomp), _p(view), _p(proj), _p(campos), R(tanfovx), R(tanfovy),
               int(prefiltered), _p(color), _p(depth), _p(radii), _p(is_used), _p(geom), _p(keys), _p(vals),
               ctypes.c_longlong(cap), _p(ranges), _p(final_T), _p(n_contrib), _p(margin))
        if n <= cap:
            break
        cap = n
    return dict(color=color, depth=depth, radii=radii, is_used=is_used, geom=geom, num_rendered=n, keys=keys[:n],
                point_list=vals[:n], ranges=ranges, final_T=final_T, n_contrib=n_contrib, margin=margin)


def raster_backward(means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy, W, H, bg, dL_dcolor,
                    dL_ddepth=None, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                    scale_modifier=1.0, sh_degree=0, dtype=np.float32):
    lib = _lib("raster")
    dt = np.dtype(dtype)
    R = ctypes.c_float if dt == np.float32 else ctypes.c_double
    fn = lib.oracle_raster_backward_f32 if dt == np.float32 else lib.oracle_raster_backward_f64
    fn.restype = ctypes.c_int
    means3D = _arr(means3D, dt, (-1, 3))
    P = means3D.shape[0]
    shs = _arr(shs, dt)
    M = 0 if shs is None else (shs.reshape(P, -1, 3).shape[1] if P > 0 else max(1, int(np.prod(shs.shape[1:])) // 3))
    colors_precomp = _arr(colors_precomp, dt)
    opacities = _arr(opacities, dt, (-1,))
    scales = _arr(scales, dt)
    rotations = _arr(rotations, dt)
    cov3D_precomp = _arr(cov3D_precomp, dt)
    view = _arr(viewmatrix, dt, (16,))
    proj = _arr(projmatrix, dt, (16,))
    campos = _arr(campos, dt, (3,))
    bg = _arr(bg, dt, (3,))
    dL_dcolor = _arr(dL_dcolor, dt, (3, H, W))
    dL_ddepth = _arr(dL_ddepth, dt, (H, W)) if dL_ddepth is not None else None
    out = dict(
        dL_dmeans2D=np.zeros((P, 3), dt), dL_dconic=np.zeros((P, 3), dt), dL_dopacity=np.zeros(P, dt),
        dL_dcolors=np.zeros((P, 3), dt), dL_ddepths=np.zeros(P, dt), dL_dmeans3D=np.zeros((P, 3), dt),
        dL_dcov3D=np.zeros((P, 6), dt), dL_dsh=np.zeros((P, max(M, 1), 3), dt) if shs is not None else None,
        dL_dscales=np.zeros((P, 3), dt) if scales is not None else None,
        dL_drots=np.zeros((P, 4), dt) if rotations is not None else None)
    fn(P, int(sh_degree), M, _p(bg), W, H, _p(means3D), _p(shs), _p(colors_precomp), _p(opacities), _p(scales),
       R(scale_modifier), _p(rotations), _p(cov3D_precomp), _p(view), _p(proj), _p(campos), R(tanfovx), R(tanfovy),
       _p(dL_dcolor), _p(dL_ddepth), _p(out["dL_dmeans2D"]), _p(out["dL_dconic"]), _p(out["dL_dopacity"]),
       _p(out["dL_dcolors"]), _p(out["dL_ddepths"]), _p(out["dL_dmeans3D"]), _p(out["dL_dcov3D"]), _p(out["dL_dsh"]),
       _p(out["dL_dscales"]), _p(out["dL_drots"]))
    return out


# ------------------------------------------------------------------------------------------ simple_knn
def knn_dist2(points):
    lib = _lib("knn")
    pts = _arr(points, np.float32, (-1, 3))
    out = np.zeros(pts.shape[0], np.float32)
    lib.oracle_knn_dist2(pts.shape[0], _p(pts), _p(out))
    return out


# ------------------------------------------------------------------------------------------ GICP
class OracleGICP:
    """CPU restatement of pygicp.FastGICP with the reference's method names (mp_Tracker.py:53-308)."""

    def __init__(self):
        self._lib = _lib("gicp")
        self._lib.oracle_gicp_create.restype = ctypes.c_void_p
        self._h = ctypes.c_void_p(self._lib.oracle_gicp_create())
        self._n = {0: 0, 1: 0}

    def __del__(self):
        try:
            self._lib.oracle_gicp_destroy(self._h)
        except Exception:
            pass

    def _set(self, which, v):
        self._lib.oracle_gicp_set_param(self._h, which, ctypes.c_double(v))

    def set_max_correspondence_distance(self, d): self._set(0, d)
    def set_max_knn_distance(self, d): self._set(1, d)
    def set_correspondence_randomness(self, k): self._set(2, k)
    def set_max_iterations(self, n): self._set(3, n)
    def set_num_threads(self, n): self._set(4, n)
    def set_regularization_method(self, m): self._set(5, m)
    def set_rotation_epsilon(self, e): self._set(6, e)
    def set_transformation_epsilon(self, e): self._set(7, e)
    def set_scale_semantics(self, mode): self._set(8, {"stddev": 0, "variance": 1}[mode] if isinstance(mode, str) else int(mode))

    def _input(self, is_target, pts):
        pts = np.asarray(pts)
        f64 = pts.dtype == np.float64
        pts = np.ascontiguousarray(pts, dtype=np.float64 if f64 else np.float32).reshape(-1, 3)
        self._n[is_target] = pts.shape[0]
        self._lib.oracle_gicp_set_input(self._h, is_target, _p(pts), pts.shape[0], int(f64))

    def set_input_target(self, pts): self._input(1, pts)
    def set_input_source(self, pts): self._input(0, pts)

    def _filter(self, is_target, n, f):
        f = np.ascontiguousarray(f, dtype=np.int32)
        self._lib.oracle_gicp_set_filter(self._h, is_target, int(n), _p(f), f.shape[0])

    def set_target_filter(self, n, f): self._filter(1, n, f)
    def set_source_filter(self, n, f): self._filter(0, n, f)
    def calculate_target_covariance_with_filter(self): self._lib.oracle_gicp_calc_cov(self._h, 1)
    def calculate_source_covariance(self): self._lib.oracle_gicp_calc_cov(self._h, 0)

    def _get(self, fn, is_target, width, dt=np.float32):
        out = np.zeros(self._n[is_target] * width, dt)
        n = fn(self._h, is_target, _p(out), self._n[is_target])
        return out[: n * width]

    def get_target_rotationsq(self): return self._get(self._lib.oracle_gicp_get_rotq, 1, 4)
    def get_target_scales(self): return self._get(self._lib.oracle_gicp_get_scales, 1, 3)
    def get_source_rotationsq(self): return self._get(self._lib.oracle_gicp_get_rotq, 0, 4)
    def get_source_scales(self): return self._get(self._lib.oracle_gicp_get_scales, 0, 3)
    def get_target_covariances(self): return self._get(self._lib.oracle_gicp_get_cov, 1, 6, np.float64).reshape(-1, 6)
    def get_source_covariances(self): return self._get(self._lib.oracle_gicp_get_cov, 0, 6, np.float64).reshape(-1, 6)

    def set_target_covariances_fromqs(self, rots, scales):
        r = np.ascontiguousarray(rots, dtype=np.float32).ravel()
        s = np.ascontiguousarray(scales, dtype=np.float32).ravel()
        if self._lib.oracle_gicp_set_target_cov_fromqs(self._h, _p(r), r.size, _p(s), s.size) != 0:
            raise RuntimeError("set_target_covariances_fromqs: size mismatch with the current target cloud")

    def align(self, initial_pose):
        init = np.ascontiguousarray(initial_pose, dtype=np.float64).reshape(4, 4)
        out = np.zeros((4, 4), np.float64)
        self.iterations = self._lib.oracle_gicp_align(self._h, _p(init), _p(out))
        return out.astype(np.float32)

    def get_source_correspondence(self):
        cap = self._n[0]
        idx = np.zeros(cap, np.int32)
        d2 = np.zeros(cap, np.float32)
        n = self._lib.oracle_gicp_get_corr(self._h, _p(idx), _p(d2), cap)
        return idx[:n], d2[:n]

    def stats(self):
        out = np.zeros(6, np.float64)
        self._lib.oracle_gicp_stats(self._h, _p(out))
        return dict(iterations=int(out[0]), lm_trials=int(out[1]), cost=out[2], converged=bool(out[3]))

    def num_threads(self):
        return self._lib.oracle_gicp_num_threads()
