"""StageEvaluator: getLocalSystem on the GPU for the stage-OCP model zoo (include/mpcqp.h, mpcqp_stage_*).

Replaces the per-iteration evaluation of the CasADi localSystemFunction_ (reference
src/sqp_solver/SQPOptimizationSolver.cpp:100-120) by one HIP kernel over the batch; inputs and outputs are torch CUDA
tensors (torch is only the allocator / stream provider).  The NumPy path of models.StageOCP.local_system is the host
statement of the same formulas and the parity checker in tests/."""
import ctypes as C

import numpy as np

from . import _lib

MODEL_IDS = {"double_integrator": 0, "quadrotor": 1, "cartpole": 2}


class StageDesc(C.Structure):
    _fields_ = [("model", C.c_int), ("horizon", C.c_int), ("dt", C.c_double), ("Q", C.c_double * 16),
                ("R", C.c_double * 8), ("par", C.c_double * 8), ("device", C.c_int)]


def _bind(L):
    if getattr(L, "_stage_bound", False):
        return L
    vp, dp = C.c_void_p, C.c_void_p
    L.mpcqp_stage_default.argtypes = [C.c_int, C.c_int, C.POINTER(StageDesc)]
    L.mpcqp_stage_create.argtypes = [C.POINTER(StageDesc), C.POINTER(vp)]
    L.mpcqp_stage_create_user.argtypes = [C.POINTER(StageDesc), C.c_char_p, C.POINTER(vp)]
    L.mpcqp_stage_destroy.argtypes = [vp]
    L.mpcqp_stage_destroy.restype = None
    L.mpcqp_stage_set_weights.argtypes = [vp, dp, dp]
    L.mpcqp_stage_set_path_bounds.argtypes = [vp, dp, dp]
    L.mpcqp_stage_dims.argtypes = [vp, vp]
    L.mpcqp_stage_has_cost.argtypes = [vp]
    L.mpcqp_stage_pattern.argtypes = [vp, vp, vp, vp, vp]
    L.mpcqp_stage_eval.argtypes = [vp, C.c_int] + [dp] * 11 + [vp]
    L.mpcqp_stage_merit.argtypes = [vp, C.c_int, dp, dp, dp, dp, vp]
    L.mpcqp_stage_step.argtypes = [vp, C.c_int, C.c_double, dp, dp, dp, vp, vp]
    L._stage_bound = True
    return L


def model_params(model):
    """the parameter vector mpcqp_stage_desc.par expects, from a models.StageOCP instance"""
    if model.name == "quadrotor":
        return [model.mass, model.grav, model.arm, model.kappa] + [float(v) for v in model.inertia]
    if model.name == "cartpole":
        return [model.mc, model.mp, model.length, model.grav]
    return []


def _check(t, shape, name):
    import torch
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
        raise ValueError("%s: expected a contiguous float64 CUDA tensor" % name)
    if tuple(t.shape) != tuple(shape):
        raise ValueError("%s: expected shape %s, got %s (dimension mismatch)" % (name, tuple(shape), tuple(t.shape)))
    return t.data_ptr()


class StageEvaluator:
    def __init__(self, model=None, name=None, horizon=None, device=-1, codegen=None):
        """model: a models.StageOCP instance (its N, dt, Q, R are used), or name + horizon for the library's defaults
        (mpcqp_stage_default).  Built-in zoo models run the library's compiled functors; any other model -- or any model
        with codegen=True -- has its discrete map model.F traced, emitted as a functor and compiled for gfx950
        (optimal_control_problem_amd.codegen; the reference's gen_code / load_lib flow)."""
        L = _bind(_lib.lib())
        d = StageDesc()
        self.library = None
        if model is not None:
            from . import models as _m
            # a zoo model is an instance of a zoo class whose dynamics are not overridden (weights, horizon, parameters may differ)
            zoo = False
            for cls in (_m.DoubleIntegrator, _m.Quadrotor, _m.CartPole):
                if isinstance(model, cls) and model.name == cls.name and type(model).F is cls.F and type(model).cdyn is cls.cdyn:
                    zoo = True
            general = bool(getattr(model, "general_cost", False))
            nk = int(getattr(model, "nk", 0))
            use_codegen = (not zoo or model.nh > 0 or nk > 0 or general) if codegen is None else bool(codegen)
            if not use_codegen and (model.nh > 0 or nk > 0 or not zoo or general):
                raise ValueError("this model needs the generated evaluator (codegen=True): it is not a built-in zoo model, or has a path / link "
                                 "constraint or a general stage cost")
            _lib.check(L.mpcqp_stage_default(MODEL_IDS.get(model.name, 0) if zoo else 0, int(model.N), C.byref(d)))
            d.dt = float(model.dt)
            if model.nx > 16 or model.nu > 8:
                raise ValueError("device evaluation supports nx <= 16 and nu <= 8")
            Q0, R0 = model.Qk[0], model.Rk[0]
            for i in range(16): d.Q[i] = float(Q0[i]) if i < len(Q0) else 0.0
            for i in range(8): d.R[i] = float(R0[i]) if i < len(R0) else 0.0
            for i in range(8): d.par[i] = 0.0
            if use_codegen:
                from . import codegen as cg
                h_lo, h_hi = model.path_bounds()
                self.tape = cg.trace(model.F, model.nx, model.nu, model.hfun if model.nh else None, model.nh, h_lo[0] if model.nh else None, h_hi[0] if model.nh else None,
                                     lcost=model.lcost if general else None, lterm=model.lterm if general else None,
                                     kfun=model.kfun if nk else None, nk=nk, k_lo=model.k_lo if nk else None, k_hi=model.k_hi if nk else None)
                self.library = cg.build_device_library(self.tape)
            else:
                for i, v in enumerate(model_params(model)): d.par[i] = float(v)
        else:
            _lib.check(L.mpcqp_stage_default(MODEL_IDS[name], int(horizon), C.byref(d)))
        d.device = int(device)
        self.desc = d
        self._h = C.c_void_p()
        if self.library is not None:
            _lib.check(L.mpcqp_stage_create_user(C.byref(d), self.library.encode(), C.byref(self._h)))
        else:
            _lib.check(L.mpcqp_stage_create(C.byref(d), C.byref(self._h)))
        if model is not None and getattr(model, "varying_weights", False) and not getattr(model, "general_cost", False):
            Qk = np.ascontiguousarray(model.Qk, dtype=np.float64); Rk = np.ascontiguousarray(model.Rk, dtype=np.float64)
            _lib.check(L.mpcqp_stage_set_weights(self._h, Qk.ctypes.data, Rk.ctypes.data))
        if model is not None and model.nh and np.ndim(model.h_lo) == 2:      # bounds that differ by frame (terminal constraints)
            lo, hi = [np.ascontiguousarray(v, dtype=np.float64) for v in model.path_bounds()]
            _lib.check(L.mpcqp_stage_set_path_bounds(self._h, lo.ctypes.data, hi.ctypes.data))
        dims = np.zeros(8, np.int32)
        _lib.check(L.mpcqp_stage_dims(self._h, dims.ctypes.data))
        self.nx, self.nu, self.np, self.n, self.m, self.nnzP, self.nnzA, self.nvar = [int(v) for v in dims]
        self.ng = self.m - self.n
        self.general_cost = bool(L.mpcqp_stage_has_cost(self._h))
        self.Pp = np.zeros(self.n + 1, np.int32); self.Pi = np.zeros(self.nnzP, np.int32)
        self.Ap = np.zeros(self.n + 1, np.int32); self.Ai = np.zeros(self.nnzA, np.int32)
        _lib.check(L.mpcqp_stage_pattern(self._h, self.Pp.ctypes.data, self.Pi.ctypes.data, self.Ap.ctypes.data, self.Ai.ctypes.data))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().mpcqp_stage_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def alloc(self, batch, device="cuda"):
        """output buffers of one evaluation: dict P, q, A, l, u"""
        import torch
        mk = lambda w: torch.empty((batch, w), dtype=torch.float64, device=device)
        return dict(P=mk(self.nnzP), q=mk(self.n), A=mk(self.nnzA), l=mk(self.m), u=mk(self.m))

    def eval(self, p, x, lbx, ubx, lbg, ubg, out=None, stream=None):
        B = x.shape[0]
        if out is None:
            out = self.alloc(B, x.device)
        args = [_check(p, (B, self.np), "p"), _check(x, (B, self.nvar), "x"), _check(lbx, (B, self.nvar), "lbx"),
                _check(ubx, (B, self.nvar), "ubx"), _check(lbg, (B, self.ng), "lbg"), _check(ubg, (B, self.ng), "ubg"),
                _check(out["P"], (B, self.nnzP), "P"), _check(out["q"], (B, self.n), "q"), _check(out["A"], (B, self.nnzA), "A"),
                _check(out["l"], (B, self.m), "l"), _check(out["u"], (B, self.m), "u")]
        _lib.check(_lib.lib().mpcqp_stage_eval(self._h, B, *args, stream))
        return out

    def merit(self, p, x, stream=None):
        import torch
        B = x.shape[0]
        f = torch.empty(B, dtype=torch.float64, device=x.device); g = torch.empty(B, dtype=torch.float64, device=x.device)
        _lib.check(_lib.lib().mpcqp_stage_merit(self._h, B, _check(p, (B, self.np), "p"), _check(x, (B, self.nvar), "x"),
                                                f.data_ptr(), g.data_ptr(), stream))
        return f, g

    def step(self, alpha, dw, x, stream=None, status=None):
        """x += alpha * dw[:, np:] in place; returns max|alpha dx| per instance.  status (int32 CUDA tensor [B], optional):
        instances whose QP did not return a point keep their x"""
        import torch
        B = x.shape[0]
        sm = torch.empty(B, dtype=torch.float64, device=x.device)
        _lib.check(_lib.lib().mpcqp_stage_step(self._h, B, float(alpha), _check(dw, (B, self.n), "dw"), _check(x, (B, self.nvar), "x"),
                                               sm.data_ptr(), None if status is None else status.data_ptr(), stream))
        return sm
