"""ctypes binding of oracle/liborc.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/osqp_oracle.h).  The product path (optimal_control_problem_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

STATUS = {1: "solved", 2: "solved_inaccurate", 3: "primal_infeasible", 4: "primal_infeasible_inaccurate",
          5: "dual_infeasible", 6: "dual_infeasible_inaccurate", 7: "max_iter_reached", 9: "non_cvx", 11: "unsolved"}


class Settings(C.Structure):
    _fields_ = [("rho", C.c_double), ("sigma", C.c_double), ("alpha", C.c_double),
                ("eps_abs", C.c_double), ("eps_rel", C.c_double),
                ("eps_prim_inf", C.c_double), ("eps_dual_inf", C.c_double),
                ("adaptive_rho_tolerance", C.c_double),
                ("max_iter", C.c_int), ("check_termination", C.c_int), ("scaling", C.c_int),
                ("adaptive_rho", C.c_int), ("adaptive_rho_interval", C.c_int),
                ("scaled_termination", C.c_int), ("warm_start", C.c_int), ("linsys", C.c_int)]


def build(force=False):
    so = os.path.join(_HERE, "liborc.so")
    src = os.path.join(_HERE, "osqp_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liborc.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.orc_pattern_create.restype = C.c_void_p
        L.orc_pattern_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_pattern_destroy.argtypes = [C.c_void_p]
        L.orc_pattern_kkt_nnzL.argtypes = [C.c_void_p]
        L.orc_default_settings.argtypes = [C.POINTER(Settings)]
        L.orc_solve_batch.restype = C.c_int
        L.orc_solve_batch.argtypes = [C.c_void_p, C.POINTER(Settings), C.c_int] + \
            [C.c_void_p, C.c_long] * 5 + [C.c_void_p] * 8 + [C.c_int]
        L.orc_solve_batch_rho.restype = C.c_int
        L.orc_solve_batch_rho.argtypes = [C.c_void_p, C.POINTER(Settings), C.c_int] + \
            [C.c_void_p, C.c_long] * 5 + [C.c_void_p] * 9 + [C.c_int]
        L.orc_state_create.restype = C.c_void_p
        L.orc_state_create.argtypes = [C.c_void_p, C.POINTER(Settings), C.c_int]
        L.orc_state_destroy.argtypes = [C.c_void_p]
        L.orc_state_solve.restype = C.c_int
        L.orc_state_solve.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p, C.c_long] * 5 + [C.c_void_p] * 8 + [C.c_int]
        _LIB = L
    return _LIB


def default_settings(**kw):
    s = Settings()
    lib().orc_default_settings(C.byref(s))
    for k, v in kw.items():
        if not hasattr(s, k):
            raise KeyError(k)
        setattr(s, k, v)
    return s


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Pattern:
    def __init__(self, n, m, Pp, Pi, Ap, Ai):
        self.n, self.m = int(n), int(m)
        self.Pp = np.ascontiguousarray(Pp, dtype=np.int32)
        self.Pi = np.ascontiguousarray(Pi, dtype=np.int32)
        self.Ap = np.ascontiguousarray(Ap, dtype=np.int32)
        self.Ai = np.ascontiguousarray(Ai, dtype=np.int32)
        self.h = lib().orc_pattern_create(self.n, self.m, _p(self.Pp), _p(self.Pi), _p(self.Ap), _p(self.Ai))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_pattern_destroy(self.h)
            self.h = None

    @property
    def nnzL(self):
        return lib().orc_pattern_kkt_nnzL(self.h)

    def solve(self, Px, q, Ax, l, u, settings=None, x0=None, y0=None, nthreads=1, rho0=None):
        """Px [B,nnzP] or [nnzP] (shared); q [B,n]; Ax [B,nnzA] or [nnzA]; l,u [B,m].  Returns dict."""
        s = settings or default_settings()
        q = np.ascontiguousarray(np.atleast_2d(q), dtype=np.float64)
        B = q.shape[0]

        def prep(a, width):
            a = np.ascontiguousarray(a, dtype=np.float64)
            if a.ndim == 1:
                assert a.shape[0] == width, (a.shape, width)
                return a, 0
            assert a.shape == (B, width), (a.shape, B, width)
            return a, width

        Px, sP = prep(Px, len(self.Pi))
        Ax, sA = prep(Ax, len(self.Ai))
        l, sl = prep(np.atleast_2d(l) if np.ndim(l) == 2 else l, self.m)
        u, su = prep(np.atleast_2d(u) if np.ndim(u) == 2 else u, self.m)
        x = np.empty((B, self.n)); y = np.empty((B, self.m)); z = np.empty((B, self.m))
        status = np.empty(B, dtype=np.int32); iters = np.empty(B, dtype=np.int32); info = np.empty((B, 4))
        if x0 is not None:
            x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(B, self.n)
            y0 = np.ascontiguousarray(y0, dtype=np.float64).reshape(B, self.m)
        if rho0 is not None:
            rho0 = np.ascontiguousarray(np.broadcast_to(np.asarray(rho0, dtype=np.float64), (B,)))
        rc = lib().orc_solve_batch_rho(self.h, C.byref(s), B, _p(Px), sP, _p(q), self.n, _p(Ax), sA, _p(l), sl, _p(u), su,
                                       _p(x0), _p(y0), _p(rho0), _p(x), _p(y), _p(z), _p(status), _p(iters), _p(info), int(nthreads))
        if rc != 0:
            raise RuntimeError("orc_solve_batch failed rc=%d" % rc)
        return dict(x=x, y=y, z=z, status=status, iters=iters, obj=info[:, 0], prim_res=info[:, 1],
                    dual_res=info[:, 2], rho=info[:, 3])


class State:
    """Kept workspaces, one per instance (orc_state_*): solve() is a full setup + solve, solve_vectors() replaces q, l, u
    only -- OSQP's osqp_update_data_vec on a kept workspace."""

    def __init__(self, pattern, batch, settings=None):
        self.pat, self.B = pattern, int(batch)
        self.settings = settings or default_settings()
        self.h = lib().orc_state_create(pattern.h, C.byref(self.settings), self.B)
        if not self.h:
            raise RuntimeError("orc_state_create failed")

    def __del__(self):
        try:
            if self.h:
                lib().orc_state_destroy(self.h); self.h = None
        except Exception:
            pass

    def _run(self, vectors_only, Px, q, Ax, l, u, x0, y0, nthreads):
        B, n, m = self.B, self.pat.n, self.pat.m
        def prep(a, width):
            if a is None:
                return None, 0
            a = np.ascontiguousarray(a, dtype=np.float64)
            if a.ndim == 1:
                assert a.shape[0] == width
                return a, 0
            assert a.shape == (B, width), (a.shape, (B, width))
            return a, width
        Px, sP = prep(Px, int(self.pat.Pp[-1])); Ax, sA = prep(Ax, int(self.pat.Ap[-1]))
        q = np.ascontiguousarray(np.broadcast_to(np.asarray(q, np.float64), (B, n)))
        l = np.ascontiguousarray(np.broadcast_to(np.asarray(l, np.float64), (B, m)))
        u = np.ascontiguousarray(np.broadcast_to(np.asarray(u, np.float64), (B, m)))
        x = np.empty((B, n)); y = np.empty((B, m)); z = np.empty((B, m))
        status = np.empty(B, np.int32); iters = np.empty(B, np.int32); info = np.empty((B, 4))
        if x0 is not None:
            x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(B, n); y0 = np.ascontiguousarray(y0, dtype=np.float64).reshape(B, m)
        rc = lib().orc_state_solve(self.h, int(vectors_only), _p(Px), sP, _p(q), n, _p(Ax), sA, _p(l), m, _p(u), m,
                                   _p(x0), _p(y0), _p(x), _p(y), _p(z), _p(status), _p(iters), _p(info), int(nthreads))
        if rc != 0:
            raise RuntimeError("orc_state_solve failed rc=%d" % rc)
        return dict(x=x, y=y, z=z, status=status, iters=iters, obj=info[:, 0], prim_res=info[:, 1], dual_res=info[:, 2], rho=info[:, 3])

    def solve(self, Px, q, Ax, l, u, x0=None, y0=None, nthreads=1):
        return self._run(0, Px, q, Ax, l, u, x0, y0, nthreads)

    def solve_vectors(self, q, l, u, x0=None, y0=None, nthreads=1):
        return self._run(1, None, q, None, l, u, x0, y0, nthreads)
