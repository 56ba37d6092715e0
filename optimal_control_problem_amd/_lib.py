"""ctypes binding of libmpcqp.so (include/mpcqp.h).  There is no CPU fallback: if the library or a gfx950
device is missing every compute entry point raises."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("MPCQP_LIB") or os.path.join(_HERE, "libmpcqp.so")   # MPCQP_LIB: diagnostic builds only
_LIB = None

MEM_HOST, MEM_DEVICE = 0, 1
OK, ERR_ARG, ERR_HIP, ERR_NO_GPU, ERR_STATE, ERR_LIMIT = 0, 1, 2, 3, 4, 5

STATUS = {1: "solved", 2: "solved_inaccurate", 3: "primal_infeasible", 4: "primal_infeasible_inaccurate",
          5: "dual_infeasible", 6: "dual_infeasible_inaccurate", 7: "max_iter_reached", 9: "non_cvx", 11: "unsolved"}

EXPORTS = ["mpcqp_default_settings", "mpcqp_create", "mpcqp_create_tuned", "mpcqp_create_reduced", "mpcqp_create_presolved", "mpcqp_update", "mpcqp_warm_start", "mpcqp_keep_workspace", "mpcqp_update_vectors", "mpcqp_set_rho", "mpcqp_set_dispatch_hint", "mpcqp_solve", "mpcqp_solve_host",
           "mpcqp_get", "mpcqp_sync", "mpcqp_destroy", "mpcqp_strerror", "mpcqp_last_kernel_ms", "mpcqp_last_phase_ms",
           "mpcqp_plan_info", "mpcqp_oc_info", "mpcqp_debug_scaling", "mpcqp_debug_blockops",
           "mpcqp_stage_default", "mpcqp_stage_create", "mpcqp_stage_create_user", "mpcqp_stage_destroy", "mpcqp_stage_set_weights", "mpcqp_stage_set_path_bounds", "mpcqp_stage_dims", "mpcqp_stage_has_cost", "mpcqp_stage_pattern",
           "mpcqp_stage_eval", "mpcqp_stage_merit", "mpcqp_stage_step",
           "mpcqp_stageqp_pattern", "mpcqp_stageqp_create", "mpcqp_stageqp_handle", "mpcqp_stageqp_update", "mpcqp_stageqp_destroy"]


class Settings(C.Structure):
    _fields_ = [("rho", C.c_double), ("sigma", C.c_double), ("alpha", C.c_double),
                ("eps_abs", C.c_double), ("eps_rel", C.c_double),
                ("eps_prim_inf", C.c_double), ("eps_dual_inf", C.c_double),
                ("adaptive_rho_tolerance", C.c_double),
                ("max_iter", C.c_int), ("check_termination", C.c_int), ("scaling", C.c_int),
                ("adaptive_rho", C.c_int), ("adaptive_rho_interval", C.c_int),
                ("scaled_termination", C.c_int), ("warm_start", C.c_int), ("device", C.c_int)]


class MpcqpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mpcqp error %d: %s" % (code, msg))
        self.code = code


def build(force=False):
    """Compile libmpcqp.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    deps = [os.path.join(src, f) for f in sorted(os.listdir(src)) if f.endswith((".hip", ".hpp")) or f == "Makefile"] + [os.path.join(_HERE, "..", "include", "mpcqp.h")]
    if force or not os.path.exists(SO_PATH) or any(os.path.getmtime(d) > os.path.getmtime(SO_PATH) for d in deps):
        subprocess.check_call(["make", "-C", src, "-j", str(min(8, os.cpu_count() or 1))] + (["-B"] if force else []) + ["../libmpcqp.so"], stdout=subprocess.DEVNULL)
    return SO_PATH


def lib_path():
    """the library file this process loads (MPCQP_LIB overrides: timing build, A/B runs)"""
    return SO_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise MpcqpError(ERR_NO_GPU, "libmpcqp.so is not built (run __graft_entry__.build()); there is no CPU fallback")
        # torch bundles its own HIP runtime under the same SONAME (libamdhip64.so.7) as /opt/rocm's; two
        # copies in one process leave the second without a GPU.  Importing torch first makes libmpcqp.so
        # bind to the runtime torch already loaded, so both share one HIP context per device.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(SO_PATH)
        vp, dp, lg = C.c_void_p, C.c_void_p, C.c_long
        L.mpcqp_default_settings.argtypes = [C.POINTER(Settings)]
        L.mpcqp_create.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.POINTER(Settings), C.POINTER(vp)]
        L.mpcqp_create_tuned.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.POINTER(Settings), C.POINTER(vp)]
        L.mpcqp_create_reduced.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, vp, C.POINTER(Settings), C.POINTER(vp)]
        L.mpcqp_create_presolved.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, dp, lg, dp, lg, C.c_int, C.POINTER(Settings), C.POINTER(vp), C.POINTER(C.c_int)]
        L.mpcqp_update.argtypes = [vp, dp, lg, dp, lg, dp, lg, dp, lg, dp, lg, C.c_int]
        L.mpcqp_warm_start.argtypes = [vp, dp, dp, C.c_int]
        L.mpcqp_set_rho.argtypes = [vp, dp, C.c_int]
        L.mpcqp_keep_workspace.argtypes = [vp, C.c_int]
        L.mpcqp_set_dispatch_hint.argtypes = [vp, C.c_int]
        L.mpcqp_update_vectors.argtypes = [vp, dp, lg, dp, lg, dp, lg, C.c_int]
        L.mpcqp_solve.argtypes = [vp, vp]
        L.mpcqp_solve_host.argtypes = [vp, dp, lg, dp, lg, dp, lg, dp, lg, dp, lg, dp, dp, vp, vp, C.c_int]
        L.mpcqp_get.argtypes = [vp, dp, dp, dp, vp, vp, dp, C.c_int]
        L.mpcqp_sync.argtypes = [vp]
        L.mpcqp_destroy.argtypes = [vp]
        L.mpcqp_destroy.restype = None
        L.mpcqp_strerror.argtypes = [C.c_int]
        L.mpcqp_strerror.restype = C.c_char_p
        L.mpcqp_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.mpcqp_last_phase_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.mpcqp_plan_info.argtypes = [vp, vp]
        L.mpcqp_oc_info.argtypes = [vp, vp]
        L.mpcqp_debug_scaling.argtypes = [vp, C.c_int, dp, dp, dp]
        L.mpcqp_debug_blockops.argtypes = [dp, dp, dp, dp, dp, dp, vp]
        L.mpcqp_stageqp_pattern.argtypes = [vp, vp, vp, vp, vp, vp]
        L.mpcqp_stageqp_create.argtypes = [vp, C.c_int, C.POINTER(Settings), C.POINTER(vp)]
        L.mpcqp_stageqp_handle.argtypes = [vp]
        L.mpcqp_stageqp_handle.restype = vp
        L.mpcqp_stageqp_update.argtypes = [vp, dp, dp, dp, dp, dp, dp, dp, C.c_int, vp]
        L.mpcqp_stageqp_destroy.argtypes = [vp]
        L.mpcqp_stageqp_destroy.restype = None
        _LIB = L
    return _LIB


def check(rc):
    if rc != OK:
        raise MpcqpError(rc, lib().mpcqp_strerror(rc).decode())


def default_settings(**kw):
    s = Settings()
    lib().mpcqp_default_settings(C.byref(s))
    for k, v in kw.items():
        if not hasattr(s, k):
            raise KeyError(k)
        setattr(s, k, v)
    return s
