"""ctypes binding of libirs_hip.so (include/irs_hip.h).

There is NO fallback: if the HIP library has not been built, or a call fails,
this module raises.  Nothing under oracle/ is ever imported from here.
"""
import ctypes
import os
from ctypes import (POINTER, c_char_p, c_double, c_int, c_longlong, c_size_t,
                    c_uint32, c_uint64, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
# (IRS_HIP_LIB: a diagnostic build of the same library, e.g. tools/stamp_descent.sh)
LIB_PATH = os.environ.get("IRS_HIP_LIB") or os.path.join(_HERE, "csrc", "libirs_hip.so")

IRS_OK = 0
MODEL_PENDULUM = 0
MODEL_QUADROTOR = 1
MODEL_BICYCLE = 2
MODEL_THREE_CART = 3
MODEL_PLANAR_HAND = 4
MODEL_BOX_PIVOT = 5
MODEL_BOX_ON_BOX = 6
MODEL_BOX_PUSH = 7
MODEL_PLANAR_HAND_EXACT = 8
MODEL_BOX_PIVOT_EXACT = 9
MODEL_BOX_PUSH_EXACT = 10
SMOOTH_ZERO_ORDER_AB = 0
SMOOTH_FIRST_ORDER = 1
SMOOTH_ZERO_ORDER_B = 2

_dp = c_void_p   # device pointers travel as plain addresses

# name -> (restype, argtypes); must list every symbol declared in include/irs_hip.h
SIGNATURES = {
    "irs_abi_version": (c_int, []),
    "irs_last_error": (c_char_p, []),
    "irs_model_info": (c_int, [c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "irs_dynamics_batch": (c_int, [c_int, POINTER(c_double), c_int, _dp, _dp, c_int, _dp, c_void_p]),
    "irs_jacobian_xu_batch": (c_int, [c_int, POINTER(c_double), c_int, _dp, _dp, c_int, _dp, c_void_p]),
    "irs_contact_samples_f32": (c_int, [c_int, POINTER(c_double), c_int, _dp, _dp, _dp, c_int, _dp, _dp, _dp, c_void_p]),
    "irs_rollout_cost": (c_int, [c_int, POINTER(c_double), c_int, c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, c_void_p]),
    "irs_evaluate_cost": (c_int, [c_int, c_int, c_int, _dp, _dp, _dp, _dp, _dp, _dp, c_void_p]),
    "irs_sums_len": (c_int, [c_int, c_int]),
    "irs_smooth_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "irs_workspace_init": (c_int, [_dp, c_size_t, c_void_p]),
    "irs_smooth": (c_int, [c_int, POINTER(c_double), c_int, c_int, c_int, c_int, _dp, _dp, _dp, _dp,
                           _dp, _dp, _dp, _dp, _dp, _dp, c_size_t, c_void_p]),
    "irs_smooth_rng": (c_int, [c_int, POINTER(c_double), c_int, c_int, c_int, c_int, _dp, _dp,
                               POINTER(c_double), POINTER(c_double), c_uint64, c_uint32,
                               _dp, _dp, _dp, _dp, _dp, _dp, c_size_t, c_void_p]),
    "irs_tvlqr_descent": (c_int, [c_int, POINTER(c_double), c_int, c_int, _dp, _dp, _dp, _dp, _dp, _dp,
                                  c_double, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, c_void_p]),
    "irs_smooth_accumulate": (c_int, [c_int, POINTER(c_double), c_int, c_int, c_int, c_int, _dp, _dp,
                                      _dp, _dp, _dp, _dp, c_size_t, c_void_p]),
    "irs_smooth_accumulate_rng": (c_int, [c_int, POINTER(c_double), c_int, c_int, c_int, c_int, _dp, _dp,
                                          POINTER(c_double), POINTER(c_double), c_uint64, c_uint32,
                                          c_uint64, _dp, _dp, c_size_t, c_void_p]),
    "irs_rng_samples": (c_int, [c_int, c_int, c_int, c_int, POINTER(c_double), POINTER(c_double),
                                c_uint64, c_uint32, c_uint64, _dp, _dp, c_void_p]),
    "irs_smooth_finalize": (c_int, [c_int, POINTER(c_double), c_int, c_int, c_int, c_longlong, _dp, _dp,
                                    _dp, _dp, _dp, _dp, _dp, c_void_p]),
    "irs_smooth_finalize_ws": (c_int, [c_int, POINTER(c_double), c_int, c_int, c_int, c_longlong, _dp, _dp,
                                       _dp, _dp, _dp, _dp, _dp, _dp, c_size_t, c_void_p]),
    "irs_exact_linearize": (c_int, [c_int, POINTER(c_double), c_int, c_int, _dp, _dp, _dp, _dp, _dp, c_void_p]),
    "irs_tvlqr_riccati": (c_int, [c_int, c_int, c_int, _dp, _dp, _dp, _dp, _dp, _dp, c_double, _dp,
                                  _dp, _dp, _dp, c_void_p]),
    "irs_tvlqr_linear_rollout": (c_int, [c_int, c_int, c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, c_void_p]),
    "irs_closed_loop_rollout": (c_int, [c_int, POINTER(c_double), c_int, c_int, _dp, _dp, _dp, _dp, _dp,
                                        _dp, _dp, _dp, _dp, c_void_p]),
}

class SmoothCall(ctypes.Structure):
    """irs_smooth_call (include/irs_hip.h)."""
    _fields_ = [("model", c_int), ("n_params", c_int), ("params", c_double * 12),
                ("mode", c_int), ("T", c_int), ("N", c_int),
                ("x_trj", c_void_p), ("u_trj", c_void_p), ("dx", c_void_p), ("du", c_void_p),
                ("use_rng", c_int), ("iter", c_uint32),
                ("std_x", c_double * 32), ("std_u", c_double * 16),
                ("seed", c_uint64), ("sample_offset", c_uint64),
                ("sums", c_void_p), ("At", c_void_p), ("Bt", c_void_p), ("ct", c_void_p),
                ("info", c_void_p), ("n_total", c_longlong),
                ("workspace", c_void_p), ("workspace_bytes", c_size_t)]


class DescentCall(ctypes.Structure):
    """irs_descent_call (include/irs_hip.h)."""
    _fields_ = [("model", c_int), ("n_params", c_int), ("params", c_double * 12),
                ("T", c_int), ("alpha_R", c_double),
                ("At", c_void_p), ("Bt", c_void_p), ("ct", c_void_p), ("Q", c_void_p), ("Qd", c_void_p),
                ("R", c_void_p), ("xd_trj", c_void_p), ("x0", c_void_p),
                ("K", c_void_p), ("k", c_void_p), ("x_new", c_void_p), ("u_new", c_void_p),
                ("cost", c_void_p), ("info", c_void_p)]


class IterateCall(ctypes.Structure):
    """irs_iterate_call (include/irs_hip.h)."""
    _fields_ = [("model", c_int), ("n_params", c_int), ("params", c_double * 12),
                ("mode", c_int), ("T", c_int), ("N", c_int), ("n_descents", c_int),
                ("std_x", c_void_p), ("std_u", c_void_p), ("seed", c_uint64), ("iter0", c_uint32),
                ("Q", c_void_p), ("Qd", c_void_p), ("R", c_void_p), ("xd_trj", c_void_p), ("alpha_R", c_double),
                ("xlo", c_void_p), ("xhi", c_void_p), ("ulo", c_void_p), ("uhi", c_void_p),
                ("qp_rho", c_double), ("qp_relax", c_double), ("qp_eps", c_double), ("qp_max_iter", c_int),
                ("x_trj0", c_void_p), ("u_trj0", c_void_p),
                ("x_hist", c_void_p), ("u_hist", c_void_p), ("cost_hist", c_void_p), ("info_hist", c_void_p),
                ("scratch", c_void_p), ("scratch_bytes", c_size_t)]


class Timing(ctypes.Structure):
    """irs_timing (include/irs_hip.h)."""
    _fields_ = [("linearise_ms", c_double), ("descent_ms", c_double), ("bounds_ms", c_double), ("descents", c_int),
                ("sample_steps", c_double), ("sample_bytes", c_double)]


ITERATE_EXACT = 3
SIGNATURES["irs_iterate_scratch_bytes"] = (c_size_t, [c_int, c_int, c_int, c_int])
SIGNATURES["irs_iterate"] = (c_int, [POINTER(IterateCall), POINTER(Timing), c_void_p])
SIGNATURES["irs_tvlqr_plan_within_bounds"] = (c_int, [c_int, c_int, c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp,
                                                      _dp, c_void_p])
SIGNATURES["irs_tvlqr_box_descent_if"] = (c_int, [c_int, POINTER(c_double), c_int, c_int, _dp, _dp, _dp, _dp, _dp, _dp,
                                                  c_double, _dp, _dp, _dp, _dp, _dp, _dp, c_double, c_double, c_int,
                                                  c_double, _dp, _dp, _dp, _dp, _dp, c_void_p])
SIGNATURES["irs_cem_rollout_costs"] = (c_int, [c_int, POINTER(c_double), c_int, c_int, c_int, _dp, _dp, _dp, _dp,
                                               _dp, _dp, c_void_p])
SIGNATURES["irs_cem_rollout_costs_quasistatic"] = (c_int, [c_int, POINTER(c_double), c_int, c_int, c_int, _dp, _dp, _dp,
                                                           _dp, _dp, _dp, _dp, c_void_p])
SIGNATURES["irs_cem_refit"] = (c_int, [c_int, c_int, c_int, c_int, _dp, _dp, _dp, _dp, _dp, c_void_p])
SIGNATURES["irs_tvlqr_box_descent"] = (c_int, [c_int, POINTER(c_double), c_int, c_int, _dp, _dp, _dp, _dp, _dp, _dp,
                                               c_double, _dp, _dp, _dp, _dp, _dp, _dp, c_double, c_double, c_int,
                                               c_double, _dp, _dp, _dp, c_void_p])
SIGNATURES["irs_tvlqr_box_lds_bytes"] = (c_size_t, [c_int, c_int])
SIGNATURES["irs_quasistatic_box_descent"] = (c_int, [c_int, POINTER(c_double), c_int, c_int, _dp, _dp, _dp, _dp, _dp, _dp,
                                                     _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, c_int, c_double, c_double,
                                                     c_int, c_double, _dp, _dp, _dp, _dp, c_void_p])
SIGNATURES["irs_quasistatic_box_descent_ws"] = (c_int, [c_int, POINTER(c_double), c_int, c_int, _dp, _dp, _dp, _dp, _dp,
                                                        _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, c_int, c_double,
                                                        c_double, c_int, c_double, _dp, _dp, _dp, _dp, _dp, c_void_p])
SIGNATURES["irs_quasistatic_box_lds_bytes"] = (c_size_t, [c_int, c_int, c_int])
SIGNATURES["irs_quasistatic_box_descent_wsx"] = (c_int, [c_int, POINTER(c_double), c_int, c_int, _dp, _dp, _dp, _dp, _dp,
                                                         _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, c_int, c_double,
                                                         c_double, c_int, c_double, _dp, _dp, _dp, _dp, _dp, _dp,
                                                         c_size_t, c_void_p])
SIGNATURES["irs_quasistatic_descent_workspace_bytes"] = (c_size_t, [c_int, c_int, c_int])
SIGNATURES["irs_tvlqr_box_solve"] = (c_int, [c_int, POINTER(c_double), c_int, c_int, _dp, _dp, _dp, _dp, _dp, _dp, c_double,
                                             _dp, _dp, c_int, _dp, _dp, _dp, _dp, _dp, _dp, c_double, c_double, c_int,
                                             c_double, _dp, _dp, _dp, c_void_p])
SIGNATURES["irs_least_squares"] = (c_int, [c_int, c_int, c_int, _dp, _dp, _dp, _dp, _dp, c_void_p])
SIGNATURES["irs_smooth_run"] = (c_int, [POINTER(SmoothCall), c_void_p])
SIGNATURES["irs_descent_run"] = (c_int, [POINTER(DescentCall), c_void_p])
SIGNATURES["irs_comm_available"] = (c_int, [])
SIGNATURES["irs_comm_unique_id"] = (c_int, [c_void_p])
SIGNATURES["irs_comm_create"] = (c_int, [c_void_p, c_int, c_int, POINTER(c_void_p)])
SIGNATURES["irs_comm_destroy"] = (c_int, [c_void_p])
SIGNATURES["irs_allreduce_sums"] = (c_int, [c_void_p, _dp, c_size_t, c_void_p])
SIGNATURES["irs_smooth_step_collective"] = (c_int, [POINTER(SmoothCall), c_void_p, c_void_p])
SIGNATURES["irs_step_graph_create"] = (c_int, [POINTER(SmoothCall), c_void_p, c_void_p, POINTER(c_void_p)])
SIGNATURES["irs_step_graph_launch"] = (c_int, [c_void_p, c_void_p])
SIGNATURES["irs_step_graph_destroy"] = (c_int, [c_void_p])
SIGNATURES["irs_peer_alloc"] = (c_int, [c_size_t, POINTER(c_void_p), c_void_p])
SIGNATURES["irs_peer_create"] = (c_int, [c_int, c_int, c_void_p, c_size_t, c_void_p, POINTER(c_void_p)])
SIGNATURES["irs_peer_destroy"] = (c_int, [c_void_p, c_void_p])
SIGNATURES["irs_peer_status"] = (c_int, [c_void_p, POINTER(ctypes.c_ulonglong), POINTER(ctypes.c_ulonglong)])
SIGNATURES["irs_peer_allreduce_sums"] = (c_int, [c_void_p, _dp, c_size_t, c_void_p])
SIGNATURES["irs_smooth_step_peer"] = (c_int, [POINTER(SmoothCall), c_void_p, c_void_p])
SIGNATURES["irs_step_graph_create_peer"] = (c_int, [POINTER(SmoothCall), c_void_p, c_void_p, POINTER(c_void_p)])

_lib = None


class IrsHipError(RuntimeError):
    pass


def load():
    """Loads libirs_hip.so (once).  Raises ImportError if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "irs_mpc_amd: %s not found. Build it with `make -C irs_mpc_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.irs_abi_version() != 1:
        raise ImportError("irs_mpc_amd: ABI version mismatch in %s" % LIB_PATH)
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != IRS_OK:
        msg = load().irs_last_error().decode("utf-8", "replace")
        raise IrsHipError("%s failed (status %d): %s" % (what, rc, msg))


def dbl_array(values):
    arr = (c_double * len(values))(*[float(v) for v in values])
    return arr
