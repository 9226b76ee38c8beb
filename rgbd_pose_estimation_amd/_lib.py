"""ctypes loader for librgbdpose_hip.so (include/rgbd_pose_hip.h).  There is no fallback: if the library is
missing it must be built (``python -m rgbd_pose_estimation_amd.build``), and every compute entry point
fails with RPE_ERR_NO_DEVICE on a host without a HIP device."""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
# RPE_LIBRARY: another build of the SAME library (the stamped diagnostic build of scripts/tail_timeline.py); never a fallback
LIB_PATH = os.environ.get("RPE_LIBRARY") or os.path.join(PKG, "lib", "librgbdpose_hip.so")

RPE_OK, RPE_ERR_NO_DEVICE, RPE_ERR_HIP, RPE_ERR_ARG, RPE_ERR_STATE, RPE_ERR_DEGENERATE, RPE_ERR_ALIGN = 0, -1, -2, -3, -4, -5, -6
F32, F64 = 0, 1
XW, XC, BV, NW, NC = 0, 1, 2, 3, 4
MOD_23, MOD_33, MOD_NN = 0, 1, 2
USE_MASK, USE_WEIGHT, SKIP_INVALID = 1, 2, 4
RES_P2P, RES_P2PLANE, RES_BEARING, RES_NORMAL, RES_REPROJ = 0, 1, 2, 3, 4
ROBUST_NONE, ROBUST_HUBER, ROBUST_CAUCHY = 0, 1, 2
VOTE_33, VOTE_23, VOTE_33_23, VOTE_NN_23, VOTE_NN_33, VOTE_NN_33_23, VOTE_23_MATRIX = 0, 1, 2, 3, 4, 5, 6
SCORE_FAST, SCORE_EXACT = 0, 1
DEPTH_U16, DEPTH_F32 = 0, 1
MAP_VERTEX, MAP_NORMAL, MAP_BEARING, MAP_MODEL_VERTEX, MAP_MODEL_NORMAL = 0, 1, 2, 3, 4

# every symbol include/rgbd_pose_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "ao", "ao_ransac", "py2c",
    "rpe_abi_version", "rpe_last_error", "rpe_device_count", "rpe_create", "rpe_destroy", "rpe_synchronize",
    "rpe_set_problem", "rpe_upload", "rpe_download", "rpe_bind", "rpe_upload_mask", "rpe_upload_weight", "rpe_download_mask",
    "rpe_p2p_moments", "rpe_pose_from_moments", "rpe_sine_error_sum", "rpe_normal_eq", "rpe_normal_eq_device", "rpe_gn_solve", "rpe_gn_apply",
    "rpe_normal_eq_joint", "rpe_gn_refine_joint", "rpe_gn_refine_device", "rpe_gn_step", "rpe_comm_unique_id", "rpe_comm_init", "rpe_comm_destroy", "rpe_comm_count", "rpe_device_bus_id", "rpe_gn_step_dist", "rpe_gn_steps_dist", "rpe_gn_steps_dist_device", "rpe_p2p_export", "rpe_p2p_init", "rpe_p2p_pause", "rpe_p2p_destroy", "rpe_host_sqrt_cut", "rpe_hostex_init", "rpe_hostex_destroy", "rpe_host_exchange_open", "rpe_host_exchange_allreduce_f64", "rpe_host_exchange_allreduce_i32", "rpe_host_exchange_set_label", "rpe_host_exchange_labels_collide", "rpe_host_exchange_unlink", "rpe_host_exchange_close", "rpe_gn_refine", "rpe_debug_loop_profile", "rpe_debug_resident_state", "rpe_debug_inject_resident_fault", "rpe_debug_device_gn_update", "rpe_tune_host_thread", "rpe_timing_enable", "rpe_timing_collect", "rpe_timing_calibrate", "rpe_score", "rpe_ransac33_batch", "rpe_ransac_p3p_batch", "rpe_inlier_mask", "rpe_score_session_begin", "rpe_score_session_end", "rpe_prosac_order", "rpe_nl_round", "rpe_run", "rpe_host_hypotheses", "rpe_run_replay",
    "rpe_frame_set_depth", "rpe_frame_download", "rpe_model_from_frame", "rpe_model_upload", "rpe_associate", "rpe_icp",
    "rpe_host_random_elements", "rpe_host_prosac_samples", "rpe_host_update_num_iters", "rpe_host_sort_indexes", "rpe_host_kneip_main",
    "rpe_host_kneip", "rpe_host_nl_2p", "rpe_host_shinji", "rpe_host_se3_exp", "rpe_host_svd3", "rpe_host_calc_err",
]


class RpeProblem(C.Structure):
    _fields_ = [("n", C.c_int), ("dtype", C.c_int), ("bv", C.c_void_p), ("xc", C.c_void_p), ("nc", C.c_void_p), ("xw", C.c_void_p),
                ("nw", C.c_void_p), ("weights", C.c_void_p), ("wcols", C.c_int), ("fx", C.c_double), ("fy", C.c_double)]


class RpeTerm(C.Structure):
    _fields_ = [("kind", C.c_int), ("scale", C.c_double), ("robust", C.c_int), ("robust_k", C.c_double)]


class RpeCamera(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("width", C.c_int), ("height", C.c_int)]


class RpeIcpOptions(C.Structure):
    _fields_ = [("kind", C.c_int), ("max_iter", C.c_int), ("tol", C.c_double), ("dist_thr", C.c_double), ("cos_thr", C.c_double),
                ("use_normals", C.c_int), ("device_resident", C.c_int), ("fused", C.c_int)]


class RpeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rpe status {code}: {msg}")
        self.code = code


_lib = None


def _preload_hip_runtime():
    """One process must hold ONE HIP runtime.  PyTorch-ROCm wheels bundle their own libamdhip64.so; if this library
    pulled in /opt/rocm's copy first, torch's later initialisation fails with hipErrorNoDevice.  So when torch is
    installed, load ITS runtime globally before librgbdpose_hip.so (same soname -> the loader reuses it)."""
    import glob
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    for d in spec.submodule_search_locations:
        for cand in sorted(glob.glob(os.path.join(d, "lib", "libamdhip64.so*"))):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
                return
            except OSError:
                continue


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m rgbd_pose_estimation_amd.build` "
                               "(the HIP extension is the product; there is no CPU fallback)")
        _preload_hip_runtime()
        L = C.CDLL(LIB_PATH)
        L.rpe_last_error.restype = C.c_char_p
        L.rpe_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
        L.rpe_destroy.argtypes = [C.c_void_p]
        L.rpe_destroy.restype = None
        L.rpe_set_problem.argtypes = [C.c_void_p, C.c_int64, C.c_int]
        for name in ("rpe_upload", "rpe_download", "rpe_bind"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        for name in ("rpe_upload_mask", "rpe_upload_weight", "rpe_download_mask"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.rpe_synchronize.argtypes = [C.c_void_p]
        L.rpe_p2p_moments.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.rpe_sine_error_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rpe_pose_from_moments.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rpe_normal_eq.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.rpe_normal_eq_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.rpe_gn_solve.argtypes = [C.c_void_p, C.c_void_p]
        L.rpe_gn_apply.argtypes = [C.c_void_p, C.c_void_p]
        L.rpe_gn_refine.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double,
                                    C.c_void_p, C.c_void_p, C.c_void_p]
        L.rpe_normal_eq_joint.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.rpe_gn_refine_joint.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rpe_gn_refine_device.argtypes = L.rpe_gn_refine_joint.argtypes
        L.rpe_gn_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rpe_comm_unique_id.argtypes = [C.c_void_p]
        L.rpe_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.rpe_comm_destroy.argtypes = [C.c_void_p]
        L.rpe_p2p_export.argtypes = [C.c_void_p, C.c_void_p]
        L.rpe_p2p_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.rpe_p2p_pause.argtypes = [C.c_void_p, C.c_int]
        L.rpe_p2p_destroy.argtypes = [C.c_void_p]
        L.rpe_host_sqrt_cut.argtypes = [C.c_int, C.c_double]
        L.rpe_host_sqrt_cut.restype = C.c_double
        L.rpe_hostex_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.rpe_hostex_destroy.argtypes = [C.c_void_p]
        L.rpe_host_exchange_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
        L.rpe_host_exchange_allreduce_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.rpe_host_exchange_allreduce_i32.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.rpe_host_exchange_set_label.argtypes = [C.c_void_p, C.c_char_p]
        L.rpe_host_exchange_labels_collide.argtypes = [C.c_void_p]
        L.rpe_host_exchange_unlink.argtypes = [C.c_void_p]
        L.rpe_host_exchange_close.argtypes = [C.c_void_p]
        L.rpe_host_exchange_close.restype = None
        L.rpe_gn_step_dist.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rpe_gn_steps_dist.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.rpe_gn_steps_dist_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.rpe_debug_device_gn_update.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rpe_debug_loop_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rpe_debug_resident_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rpe_debug_inject_resident_fault.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.rpe_comm_count.argtypes = [C.c_void_p, C.c_void_p]
        L.rpe_device_bus_id.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.rpe_timing_enable.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.rpe_tune_host_thread.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.rpe_timing_collect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rpe_timing_calibrate.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.rpe_score.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p]
        L.rpe_ransac33_batch.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rpe_ransac_p3p_batch.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rpe_inlier_mask.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p]
        L.rpe_score_session_begin.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]
        L.rpe_score_session_end.argtypes = [C.c_void_p]
        L.rpe_prosac_order.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.rpe_nl_round.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rpe_frame_set_depth.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(RpeCamera), C.c_double, C.c_double, C.c_double, C.c_double]
        L.rpe_frame_download.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.rpe_model_from_frame.argtypes = [C.c_void_p, C.c_void_p]
        L.rpe_model_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(RpeCamera), C.c_void_p]
        L.rpe_associate.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_void_p]
        L.rpe_icp.argtypes = [C.c_void_p, C.POINTER(RpeIcpOptions), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        if hasattr(L, "rpe_run"):
            L.rpe_run.argtypes = [C.c_int, C.POINTER(RpeProblem), C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_double, C.c_uint64,
                                  C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rpe_host_hypotheses.argtypes = [C.c_int, C.POINTER(RpeProblem), C.c_int, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p]
        L.rpe_run_replay.argtypes = [C.c_int, C.POINTER(RpeProblem), C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p,
                                     C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ao.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ao.restype = None
        if hasattr(L, "ao_ransac"):
            L.ao_ransac.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
            L.ao_ransac.restype = None
        L.rpe_host_random_elements.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_void_p]
        L.rpe_host_prosac_samples.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_void_p]
        L.rpe_host_update_num_iters.argtypes = [C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]
        L.rpe_host_sort_indexes.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.rpe_host_kneip_main.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rpe_host_kneip.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rpe_host_nl_2p.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rpe_host_shinji.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.rpe_host_se3_exp.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rpe_host_svd3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rpe_host_calc_err.argtypes = [C.c_void_p] * 6
        for name in ("rpe_host_random_elements", "rpe_host_prosac_samples", "rpe_host_sort_indexes", "rpe_host_nl_2p", "rpe_host_shinji",
                     "rpe_host_se3_exp", "rpe_host_svd3", "rpe_host_calc_err"):
            getattr(L, name).restype = None
        L.py2c.argtypes = [C.c_void_p, C.c_int]
        L.py2c.restype = None
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise RpeError(rc, lib().rpe_last_error().decode())
    return rc


def device_count() -> int:
    return lib().rpe_device_count()
