"""ctypes binding of the C ABI in include/lmh.h (linearmpchumanoid_amd/liblmh_hip.so).

The library is hand-written HIP for gfx950; there is no CPU implementation behind it.  If the
shared object is missing this module raises at load time -- it never falls back to anything.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LMH_DIAG=1 selects the diagnostic build of the same sources (in-kernel phase stamps, scripts/gpu_phase_stamps.py)
# LMH_VARIANT=name selects an experiment build (linearmpchumanoid_amd/build.py); neither is the shipped library
_VAR = os.environ.get("LMH_VARIANT", "").split(":")[0]
SO_PATH = os.path.join(_HERE, "liblmh_hip_diag.so" if os.environ.get("LMH_DIAG") == "1" else ("liblmh_hip_var_%s.so" % _VAR if _VAR else "liblmh_hip.so"))

STATE_STRIDE = 96
OUT_STRIDE = 80
STATUS_STRIDE = 4
DEBUG_STRIDE = 4096
LINK_STRIDE = 13
SEG_STRIDE = 52

FLAG_QP_MAXITER = 1
FLAG_NONFINITE = 2
FLAG_ZMP_RANGE = 4
FLAG_NOT_SPD = 8
FLAG_QP_FP64_ROUTE = 16
FLAG_UNFINISHED = 32          # lmh_rollout: the robot did not get all its ticks (a wait of the kernel's work queue ran out)
ERR_UNFINISHED = -5

PHASE_DOUBLE, PHASE_RIGHT, PHASE_LEFT, PHASE_FLIGHT = 0, 1, 2, 3
PRECISION_FP64, PRECISION_MIXED, PRECISION_FP32 = 0, 1, 2   # lmh_config.precision (include/lmh.h)
SUMMARY_WIDTH = 16

# every symbol include/lmh.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "lmh_config_default", "lmh_last_error", "lmh_device_count", "lmh_create", "lmh_destroy",
    "lmh_num_instances", "lmh_horizon", "lmh_set_model", "lmh_get_mass", "lmh_nominal_links",
    "lmh_set_refs", "lmh_set_refs_stance", "lmh_set_foot_coeffs", "lmh_set_zcom", "lmh_get_mpc_gain",
    "lmh_eval", "lmh_eval_debug", "lmh_rollout", "lmh_ik", "lmh_eval_host", "lmh_set_prev_velocity_host",
    "lmh_synchronize", "lmh_robot_com", "lmh_robot_com_host", "lmh_last_out_host", "lmh_ik_host", "lmh_set_segments", "lmh_set_xscale",
    "lmh_make_summary", "lmh_write_summary", "lmh_read_summary", "lmh_write_log", "lmh_read_log",
    "lmh_gen_walk", "lmh_gen_jump", "lmh_num_ref_samples", "lmh_num_segments", "lmh_get_refs",
]


class LmhConfig(C.Structure):
    """struct lmh_config (include/lmh.h); defaults are the reference's literals."""
    _fields_ = [(n, C.c_double) for n in (
        "dt", "time_horizon", "z_com", "gravity", "alpha", "beta", "mu",
        "kp_joints", "kd_joints", "kp_mom", "kd_mom", "kp_feet", "kd_feet",
        "w_com_lin", "w_com_ang", "w_base_pos", "w_base_ang", "w_joints", "w_force", "w_foot",
        "eps_coeff")] + [("warm_start", C.c_int32), ("max_qp_iters", C.c_int32), ("precision", C.c_int32), ("bpp_rounds", C.c_int32),
                                     ("plant", C.c_int32), ("reserved", C.c_int32)] + [(n, C.c_double) for n in ("contact_k", "contact_d", "contact_dt", "contact_mu", "mpc_dt")]


_lib = None


def lib():
    """Load the HIP library.  Fails loudly if it has not been built (python -m linearmpchumanoid_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            f"{SO_PATH} is missing: the MI355X HIP library has not been built "
            "(run `python __graft_entry__.py` or `python linearmpchumanoid_amd/build.py`). "
            "There is no CPU fallback.")
    L = C.CDLL(SO_PATH)
    vp, ip, dp = C.c_void_p, C.c_int, C.c_double
    L.lmh_config_default.argtypes = [C.POINTER(LmhConfig)]
    L.lmh_config_default.restype = None
    L.lmh_last_error.restype = C.c_char_p
    L.lmh_device_count.restype = ip
    L.lmh_create.argtypes = [C.POINTER(LmhConfig), ip, ip, C.POINTER(vp)]
    L.lmh_destroy.argtypes = [vp]
    L.lmh_num_instances.argtypes = [vp]
    L.lmh_horizon.argtypes = [vp]
    L.lmh_set_model.argtypes = [vp, vp, ip]
    L.lmh_get_mass.argtypes = [vp, vp]
    L.lmh_nominal_links.argtypes = [vp]
    L.lmh_nominal_links.restype = None
    L.lmh_set_refs.argtypes = [vp, vp, vp, vp, ip]
    L.lmh_set_refs_stance.argtypes = [vp, dp, ip]
    L.lmh_set_foot_coeffs.argtypes = [vp, vp, vp, vp, vp]
    L.lmh_set_zcom.argtypes = [vp, vp, ip]
    L.lmh_get_mpc_gain.argtypes = [vp, vp]
    L.lmh_eval.argtypes = [vp, vp, vp, vp, vp]
    L.lmh_eval_debug.argtypes = [vp, vp, vp, vp, vp, vp]
    L.lmh_rollout.argtypes = [vp, vp, vp, vp, vp, ip, vp]
    L.lmh_ik.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.lmh_eval_host.argtypes = [vp, vp, vp, dp, vp, vp, vp, vp]
    L.lmh_set_prev_velocity_host.argtypes = [vp, vp]
    L.lmh_synchronize.argtypes = [vp, vp]
    L.lmh_robot_com.argtypes = [vp, vp, vp, vp]
    L.lmh_last_out_host.argtypes = [vp, vp]
    L.lmh_robot_com_host.argtypes = [vp, vp, vp]
    L.lmh_ik_host.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.lmh_set_segments.argtypes = [vp, vp, ip, vp, ip]
    L.lmh_set_xscale.argtypes = [vp, vp, ip]
    u64, u64p, dpp = C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_double)
    L.lmh_gen_walk.argtypes = [vp, dp, ip, dp, dp, dp, dp, ip, dp]
    L.lmh_gen_jump.argtypes = [vp, dp, dp, dp]
    L.lmh_num_ref_samples.argtypes = [vp]
    L.lmh_num_segments.argtypes = [vp]
    L.lmh_get_refs.argtypes = [vp, vp, vp, vp, vp, vp]
    L.lmh_make_summary.argtypes = [vp, vp, vp, vp, vp, vp]
    L.lmh_write_summary.argtypes = [C.c_char_p, vp, u64, dp]
    L.lmh_read_summary.argtypes = [C.c_char_p, vp, u64, u64p, dpp]
    L.lmh_write_log.argtypes = [C.c_char_p, vp, u64, u64, dp, dp]
    L.lmh_read_log.argtypes = [C.c_char_p, vp, u64, u64p, u64p, dpp, dpp]
    for name in EXPORTS:
        fn = getattr(L, name)
        if fn.restype is C.c_int and name not in ("lmh_last_error", "lmh_config_default", "lmh_nominal_links"):
            fn.restype = ip
    _lib = L
    return L


class LmhError(RuntimeError):
    def __init__(self, msg, code=0):
        super().__init__(msg)
        self.code = code


def check(rc):
    if rc != 0:
        raise LmhError(f"lmh error {rc}: {lib().lmh_last_error().decode()}", rc)
