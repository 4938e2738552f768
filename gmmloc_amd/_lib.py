"""ctypes binding of the C-ABI declared in include/gmmloc_hip.h."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GMMLOC_HIP_LIB", os.path.join(_HERE, "libgmmloc_hip.so"))  # override: A/B builds


class gl_camera(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("bf", C.c_double), ("width", C.c_int32), ("height", C.c_int32)]


class gl_params(C.Structure):
    _fields_ = [("neighbor_dist_thresh", C.c_double), ("tri_lambda2", C.c_float),
                ("tri_str_thresh", C.c_float), ("ba_lambda2", C.c_float),
                ("tri_check_str_chi2", C.c_int32), ("ba_first_as_prior", C.c_int32),
                ("sigma2_inv", C.c_float * 8)]


class gl_track_anchor(C.Structure):
    _fields_ = [("prior_dev", C.c_void_p), ("F", C.c_int32), ("fixed_pose_dev", C.c_void_p), ("fixed_obs_dev", C.c_void_p),
                ("fixed_oct_dev", C.c_void_p), ("fixed_erase_dev", C.c_void_p)]


_lib = None


def load():
    """Load libgmmloc_hip.so.  Fails loudly when the extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    try:  # make torch's bundled libamdhip64.so.7 the one the loader binds (same SONAME)
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "gmmloc_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    P = C.POINTER
    sig = {
        "gl_default_params": (None, [P(gl_params)]),
        "gl_last_error_string": (C.c_char_p, []),
        "gl_device_count": (i32, []),
        "gl_ctx_create": (i32, [i32, vp, P(vp)]),
        "gl_ctx_destroy": (i32, [vp]),
        "gl_ctx_synchronize": (i32, [vp]),
        "gl_ctx_stream": (vp, [vp]),
        "gl_ctx_set_option": (i32, [vp, C.c_char_p, C.c_double]),
        "gl_ctx_get_option": (i32, [vp, C.c_char_p, P(C.c_double)]),
        "gl_ctx_timing_enable": (i32, [vp, i32]),
        "gl_ctx_timing_read": (i32, [vp, i32, P(C.c_double), P(i64), i32]),
        "gl_ctx_set_stats_buffer": (i32, [vp, vp, i32]),
        "gl_ctx_set_stats_buffers": (i32, [vp, vp, vp, i32]),
        "gl_ctx_set_edge_stats_buffer": (i32, [vp, vp, i32]),
        "gl_ctx_counter_read": (i32, [vp, i32, P(i64), i32]),
        "gl_gmm_create": (i32, [vp, vp, vp, i32, P(gl_params), P(vp)]),
        "gl_gmm_load_file": (i32, [vp, C.c_char_p, P(gl_params), P(vp)]),
        "gl_gmm_save_file": (i32, [vp, C.c_char_p]),
        "gl_gmm_destroy": (i32, [vp]),
        "gl_gmm_count": (i32, [vp]),
        "gl_gmm_file_read": (i32, [C.c_char_p, vp, vp, i32, P(i32)]),
        "gl_gmm_file_write": (i32, [C.c_char_p, vp, vp, vp, i32]),
        "gl_write_tum_trajectory": (i32, [C.c_char_p, vp, vp, i32]),
        "gl_gmm_get": (i32, [vp, i32, vp, C.c_size_t]),
        "gl_gmm_nbs_count": (i32, [vp]),
        "gl_associate3d": (i32, [vp, vp, vp, i32, i32, vp, vp]),
        "gl_gmm_index_info": (i32, [vp, P(C.c_double)]),
        "gl_gmm_index_bytes": (i32, [vp, P(C.c_double)]),
        "gl_assoc_index_work": (i32, [vp, vp, vp, i32, vp]),
        "gl_knn3d": (i32, [vp, vp, vp, i32, i32, vp, vp]),
        "gl_search2d": (i32, [vp, vp, P(gl_camera), i32, vp, i32, vp, vp, i32, vp, vp, i32, vp, vp]),
        "gl_search_by_projection": (i32, [vp, P(gl_camera), C.c_float, i32, i32, i32] + [vp] * 10 + [C.c_float, C.c_float, vp, vp]),
        "gl_search_by_projection_frame": (i32, [vp, P(gl_camera), C.c_float, i32, i32, i32] + [vp] * 13 + [C.c_float, i32, i32, vp, vp]),
        "gl_search_for_triangulation": (i32, [vp, C.c_float, i32, i32, i32, i32, i32] + [vp] * 22 + [i32, i32, vp, vp]),
        "gl_search_by_bow": (i32, [vp, C.c_float, i32, i32, i32, i32, i32, i32] + [vp] * 15),
        "gl_fuse_search": (i32, [vp, vp, C.c_float, i32, i32, i32] + [vp] * 8 + [C.c_float, vp, vp]),
        "gl_project_map_points": (i32, [vp, vp, C.c_float, i32, i32] + [vp] * 12),
        "gl_level_steps": (i32, [C.c_float, vp]),
        "gl_track_frame_chain": (i32, [vp, vp, vp, C.c_float, i32, i32, i32, i32, vp, C.c_float, C.c_float, C.c_float, i32]),
        "gl_track_frame_chain_front": (i32, [vp, vp, vp, C.c_float, i32, i32, i32, i32, vp, C.c_float, i32]),
        "gl_track_frame_chain_back": (i32, [vp, vp, vp, C.c_float, i32, i32, i32, i32, vp, C.c_float, C.c_float]),
        "gl_search_local_points": (i32, [vp, vp, C.c_float, i32, i32, i32] + [vp] * 13 + [C.c_float, C.c_float, vp, vp, vp]),
        "gl_gather_triangulation_matches": (i32, [vp, i32, i32, i32, i32, i32] + [vp] * 32),
        "gl_optimize_point": (i32, [vp, vp, P(gl_camera), P(gl_params), i32] + [vp] * 10),
        "gl_check_map_association": (i32, [vp, vp, P(gl_camera), P(gl_params), i32, i32] + [vp] * 6 + [i32, vp]),
        "gl_optimize_triangulation": (i32, [vp, vp, P(gl_camera), P(gl_params), i32] + [vp] * 11 + [i32, vp]),
        "gl_create_map_points": (i32, [vp, vp, P(gl_camera), P(gl_params), C.c_float, i32] + [vp] * 12 + [i32, vp, vp, vp]),
        "gl_optimize_current_pose": (i32, [vp, P(gl_camera), P(gl_params), i32, i32, vp, vp, vp, vp, vp, vp]),
        "gl_joint_optimization": (i32, [vp, vp, P(gl_camera), P(gl_params), i32, i32, i32, i32, i32] + [vp] * 11),
        "gl_joint_optimization_stoppable": (i32, [vp, vp, P(gl_camera), P(gl_params), i32, i32, i32, i32, i32] + [vp] * 12),
        "gl_track_frames": (i32, [vp, vp, P(gl_camera), P(gl_params), i32, i32, vp, vp, vp, vp, vp, vp]),
        "gl_track_frames_anchored": (i32, [vp, vp, P(gl_camera), P(gl_params), i32, i32, vp, vp, vp, vp, vp, vp, P(gl_track_anchor)]),
        "gl_track_frame_host_anchored": (i32, [vp, vp, P(gl_camera), P(gl_params), i32, vp, vp, vp, vp, vp]),
        "gl_malloc": (i32, [vp, C.c_size_t, P(vp)]),
        "gl_free": (i32, [vp, vp]),
        "gl_memcpy_h2d": (i32, [vp, vp, vp, C.c_size_t]),
        "gl_memcpy_d2h": (i32, [vp, vp, vp, C.c_size_t]),
        "gl_track_frame_host": (i32, [vp, vp, P(gl_camera), P(gl_params), i32, vp, vp, vp, vp, vp]),
        "gl_malloc_host": (i32, [vp, C.c_size_t, P(vp)]),
        "gl_free_host": (i32, [vp, vp]),
        "gl_memcpy_h2d_async": (i32, [vp, vp, vp, C.c_size_t]),
        "gl_memcpy_d2h_async": (i32, [vp, vp, vp, C.c_size_t]),
    }
    missing = []
    for name, (res, args) in sig.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    lib._gl_missing = missing
    lib._gl_signatures = sig
    _lib = lib
    return lib
