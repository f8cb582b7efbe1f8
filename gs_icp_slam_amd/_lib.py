"""ctypes binding of libgsicp_hip.so (the C ABI declared in include/gsicp_hip.h).

There is NO fallback: if the shared library is missing or fails to load, importing any product module raises.
PyTorch is used by the callers only for device memory and streams; no torch type crosses this boundary.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgsicp_hip.so")

c_void_p, c_int, c_float, c_double, c_size_t, c_char_p = (ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_double,
                                                          ctypes.c_size_t, ctypes.c_char_p)
RESIZE_FN = ctypes.CFUNCTYPE(c_void_p, c_void_p, c_size_t)

# Every symbol include/gsicp_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "gsicp_abi_version": (c_int, []),
    "gsicp_last_error": (c_char_p, []),
    "gsicp_device_count": (c_int, []),
    "gsicp_profile_enable": (c_int, [c_int]),
    "gsicp_profile_num_stages": (c_int, []),
    "gsicp_profile_stage_name": (c_char_p, [c_int]),
    "gsicp_profile_read": (c_int, [c_void_p, c_void_p, c_int]),
    "gsicp_raster_forward": (c_int, [RESIZE_FN, c_void_p, RESIZE_FN, c_void_p, RESIZE_FN, c_void_p, c_int, c_int, c_int, c_void_p,
                                     c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_float, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_int, c_int, c_int, c_int, c_void_p]),
    "gsicp_raster_forward_async": (c_int, [RESIZE_FN, c_void_p, RESIZE_FN, c_void_p, RESIZE_FN, c_void_p, c_int, c_int, c_int, c_void_p,
                                           c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_float, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "gsicp_raster_backward_scratch_bytes": (c_size_t, [c_int, c_int, c_int]),
    "gsicp_raster_backward": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "gsicp_raster_mark_visible": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gsicp_raster_layout": (c_int, [c_int, c_int, c_int, c_int, ctypes.POINTER(c_size_t)]),
    "gsicp_raster_set_legacy_backward": (c_int, [c_int]),
    "gsicp_knn_dist2": (c_int, [c_int, c_void_p, c_void_p, c_void_p]),
    "gsicp_mapper_loss_scratch_bytes": (c_size_t, [c_int, c_int]),
    "gsicp_mapper_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_float, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p]),
    "gsicp_mapper_loss_sharded": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_float, c_int, c_int, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p]),
    "gsicp_store_compact_scratch_bytes": (c_size_t, [c_int]),
    "gsicp_store_compact": (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gsicp_tiles_chunk_floats": (c_size_t, [c_int, c_int, c_int]),
    "gsicp_tiles_pack": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gsicp_tiles_unpack": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gsicp_rows_pack_scratch_bytes": (c_size_t, [c_int]),
    "gsicp_rows_pack": (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, ctypes.c_uint, c_void_p, c_void_p]),
    "gsicp_rows_unpack": (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "gsicp_mapper_set_view": (c_int, [c_int, c_int] + [c_void_p] * 11),
    "gsicp_mapper_select_view": (c_int, [c_void_p] * 10),
    "gsicp_mapper_select_view_zero": (c_int, [c_void_p] * 10 + [c_size_t, c_void_p]),
    "gsicp_raster_last_zero_region": (c_int, [c_void_p, c_void_p]),
    "gsicp_raster_set_tile_sort_lds": (c_int, [c_int]),
    "gsicp_mapper_loss_indirect": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_float, c_int, c_int, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p]),
    "gsicp_mapper_loss_indirect_bump": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_float, c_int, c_int, c_void_p, c_void_p,
                                                c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_uint, c_void_p, c_void_p]),
    "gsicp_mapper_loss_set_hoist": (c_int, [c_int]),
    "gsicp_mapper_activations_forward": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gsicp_mapper_activations_backward": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                  c_void_p, c_void_p, c_void_p]),
    "gsicp_adam_step": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_int,
                                c_void_p]),
    "gsicp_adam_step_capturable": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float,
                                           c_void_p, c_void_p]),
    "gsicp_adam_step_guarded": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float,
                                        c_void_p, c_int, c_void_p, ctypes.c_uint, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gsicp_adam_step_masked": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float,
                                       c_void_p, c_int, c_void_p, ctypes.c_uint, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gsicp_adam_step_sparse": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float,
                                       c_void_p, c_int, c_void_p, ctypes.c_uint, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gsicp_stream_create_cu_mask": (c_void_p, [c_int, c_int]),
    "gsicp_stream_destroy": (c_int, [c_void_p]),
    "gsicp_gicp_create": (c_void_p, []),
    "gsicp_gicp_destroy": (None, [c_void_p]),
    "gsicp_gicp_set_max_correspondence_distance": (c_int, [c_void_p, c_double]),
    "gsicp_gicp_set_max_knn_distance": (c_int, [c_void_p, c_double]),
    "gsicp_gicp_set_correspondence_randomness": (c_int, [c_void_p, c_int]),
    "gsicp_gicp_set_max_iterations": (c_int, [c_void_p, c_int]),
    "gsicp_gicp_set_num_threads": (c_int, [c_void_p, c_int]),
    "gsicp_gicp_set_regularization_method": (c_int, [c_void_p, c_int]),
    "gsicp_gicp_set_scale_semantics": (c_int, [c_void_p, c_int]),
    "gsicp_gicp_set_rotation_epsilon": (c_int, [c_void_p, c_double]),
    "gsicp_gicp_set_transformation_epsilon": (c_int, [c_void_p, c_double]),
    "gsicp_gicp_set_input_target": (c_int, [c_void_p, c_void_p, c_int, c_int]),
    "gsicp_gicp_set_input_source": (c_int, [c_void_p, c_void_p, c_int, c_int]),
    "gsicp_gicp_set_target_filter": (c_int, [c_void_p, c_int, c_void_p, c_int]),
    "gsicp_gicp_set_source_filter": (c_int, [c_void_p, c_int, c_void_p, c_int]),
    "gsicp_gicp_calculate_target_covariance_with_filter": (c_int, [c_void_p]),
    "gsicp_gicp_calculate_source_covariance": (c_int, [c_void_p]),
    "gsicp_gicp_get_target_rotationsq": (c_int, [c_void_p, c_void_p, c_int]),
    "gsicp_gicp_get_target_scales": (c_int, [c_void_p, c_void_p, c_int]),
    "gsicp_gicp_get_source_rotationsq": (c_int, [c_void_p, c_void_p, c_int]),
    "gsicp_gicp_get_source_scales": (c_int, [c_void_p, c_void_p, c_int]),
    "gsicp_gicp_set_target_covariances_fromqs": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int]),
    "gsicp_gicp_align": (c_int, [c_void_p, c_void_p, c_void_p]),
    "gsicp_gicp_get_source_correspondence": (c_int, [c_void_p, c_void_p, c_void_p, c_int]),
    "gsicp_gicp_knn_stats": (c_int, [c_void_p, c_void_p]),
    "gsicp_gicp_align_trace": (c_int, [c_void_p, c_void_p, c_int]),
    "gsicp_gicp_stream": (c_void_p, [c_void_p]),
    "gsicp_gicp_set_input_target_device": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int]),
    "gsicp_gicp_set_input_source_device": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int]),
    "gsicp_gicp_set_target_covariances_fromqs_device": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int]),
    "gsicp_gicp_set_target_from_gaussians_device": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p]),
    "gsicp_gicp_set_source_track_device": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int]),
    "gsicp_frontend_make_pointcloud": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_float, c_float, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gsicp_gicp_get_source_rotationsq_device": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "gsicp_gicp_get_source_scales_device": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "gsicp_gicp_num_source": (c_int, [c_void_p]),
    "gsicp_gicp_num_target": (c_int, [c_void_p]),
    "gsicp_gicp_target_index_stats": (c_int, [c_void_p, c_void_p]),
    "gsicp_gicp_last_align_stats": (c_int, [c_void_p, c_void_p]),
    "gsicp_gicp_get_final_hessian": (c_int, [c_void_p, c_void_p]),
    "gsicp_gicp_debug_abort_next_align": (c_int, [c_void_p]),
    "gsicp_debug_wave_sort": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gsicp_gicp_barrier_retries": (c_int, [c_void_p]),
}

_lib = None


def load():
    """Load the HIP library (once).  Raises ImportError if it is not built — there is no CPU path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -m gs_icp_slam_amd.build` "
                "(hipcc --offload-arch=gfx950).  gs_icp_slam_amd has no CPU fallback.")
        try:
            import torch  # noqa: F401  (loads the HIP runtime this library is linked against; see build.py)
        except ImportError:
            pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        if lib.gsicp_abi_version() != 5:
            raise ImportError("libgsicp_hip.so ABI version mismatch")
        if os.environ.get("GSICP_ANNOUNCE"):   # tools/run_reference_slam.py: show which processes of the reference run loaded the library
            print(f"GSICP_LOADED {LIB_PATH} pid={os.getpid()}", flush=True)
        _lib = lib
    return _lib


def last_error():
    return load().gsicp_last_error().decode(errors="replace")


def check(rc, what):
    if rc < 0:
        raise RuntimeError(f"{what} failed (rc={rc}): {last_error()}")
    return rc


# ---- optional call trace (GSICP_CALL_TRACE=<dir>): wall-clock enter/exit of every drop-in entry point, one file per process.
# Lets the untouched reference loop be profiled from the outside (tools/run_reference_slam.py --trace): the gaps BETWEEN calls are
# the reference's own host code.  Costs two perf_counter() reads per call when on, one dict lookup when off.
_TRACE_DIR = os.environ.get("GSICP_CALL_TRACE")
_trace_fh = None


def traced(name):
    def deco(fn):
        if not _TRACE_DIR:
            return fn
        import functools
        import time

        @functools.wraps(fn)
        def wrapper(*a, **k):
            global _trace_fh
            t0 = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                t1 = time.perf_counter()
                if _trace_fh is None:
                    os.makedirs(_TRACE_DIR, exist_ok=True)
                    _trace_fh = open(os.path.join(_TRACE_DIR, f"{os.getpid()}.trace"), "a", buffering=1)
                _trace_fh.write(f"{name} {t0:.6f} {t1:.6f}\n")
        return wrapper
    return deco


def profile_enable(on=True):
    load().gsicp_profile_enable(int(bool(on)))


def profile_read():
    """-> {stage_name: (total_ms, launches)} accumulated since the last read (synchronises the recorded events)."""
    lib = load()
    n = lib.gsicp_profile_num_stages()
    ms = (ctypes.c_double * n)()
    cnt = (ctypes.c_int * n)()
    lib.gsicp_profile_read(ms, cnt, n)
    return {lib.gsicp_profile_stage_name(i).decode(): (ms[i], cnt[i]) for i in range(n)}
