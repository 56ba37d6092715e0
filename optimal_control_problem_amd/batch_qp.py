"""BatchQP: thin object wrapper over the C ABI (include/mpcqp.h) for a batch of QPs sharing a sparsity.

Accepts NumPy arrays (host memory, copied by the library like CuCaQP copies into its members,
reference include/optimal_control_problem/sqp_solver/CuCaQP.h:83-87) or torch CUDA tensors (device
memory, borrowed; torch is only the allocator/stream provider here)."""
import ctypes as C

import numpy as np

from . import _lib


def _is_torch(a):
    return hasattr(a, "data_ptr") and hasattr(a, "is_cuda")


def _ptr_stride(a, width, batch, name):
    """-> (pointer, stride_in_doubles, mem, keepalive)"""
    if a is None:
        return None, 0, None, None
    if _is_torch(a):
        import torch
        if a.dtype != torch.float64 or not a.is_contiguous():
            raise ValueError("%s: torch tensor must be contiguous float64" % name)
        shape = tuple(a.shape)
        mem = _lib.MEM_DEVICE if a.is_cuda else _lib.MEM_HOST
        ptr = a.data_ptr()
        keep = a
    else:
        keep = np.ascontiguousarray(a, dtype=np.float64)
        shape = keep.shape
        mem = _lib.MEM_HOST
        ptr = keep.ctypes.data
    if len(shape) == 1:
        if shape[0] != width:
            raise ValueError("%s: expected %d values, got %d (dimension mismatch)" % (name, width, shape[0]))
        return ptr, 0, mem, keep
    if shape != (batch, width):
        raise ValueError("%s: expected shape (%d, %d), got %s (dimension mismatch)" % (name, batch, width, shape))
    return ptr, width, mem, keep


class BatchQP:
    def __init__(self, n, m, batch, Pp, Pi, Ap, Ai, settings=None, fixed_rows=None, tuned=False, presolve_bounds=None, **kw):
        """fixed_rows: opt-in reduced form (mpcqp_create_reduced) -- singleton rows of A with l = u in every instance, whose
        variables are substituted before the solve; None (default) = the full form, what the reference's OSQP solves.
        presolve_bounds = (l, u): the same with the rows FOUND from the first update's bounds (mpcqp_create_presolved); self.nfixed says how many.
        tuned: opt-in mpcqp_create_tuned -- the kernel family chosen by measurement on this pattern instead of by rule"""
        self.n, self.m, self.batch = int(n), int(m), int(batch)
        self.Pp = np.ascontiguousarray(Pp, dtype=np.int32); self.Pi = np.ascontiguousarray(Pi, dtype=np.int32)
        self.Ap = np.ascontiguousarray(Ap, dtype=np.int32); self.Ai = np.ascontiguousarray(Ai, dtype=np.int32)
        if len(self.Pp) != self.n + 1 or len(self.Ap) != self.n + 1:
            raise ValueError("colptr arrays must have n + 1 entries")
        self.settings = settings if settings is not None else _lib.default_settings(**kw)
        self._h = C.c_void_p()
        self._keep = []
        L = _lib.lib()
        self.nfixed = 0
        if presolve_bounds is not None:
            lp, ls_, lmem, lk = _ptr_stride(presolve_bounds[0], self.m, self.batch, "l"); up, us_, umem, uk = _ptr_stride(presolve_bounds[1], self.m, self.batch, "u")
            if lmem != umem:
                raise ValueError("l and u must live in the same memory space")
            nf = C.c_int(0)
            _lib.check(L.mpcqp_create_presolved(self.n, self.m, self.batch, self.Pp.ctypes.data, self.Pi.ctypes.data, self.Ap.ctypes.data, self.Ai.ctypes.data,
                                                lp, ls_, up, us_, lmem, C.byref(self.settings), C.byref(self._h), C.byref(nf)))
            self.nfixed = int(nf.value)
        elif fixed_rows is None:
            _lib.check((L.mpcqp_create_tuned if tuned else L.mpcqp_create)(self.n, self.m, self.batch, self.Pp.ctypes.data, self.Pi.ctypes.data,
                                      self.Ap.ctypes.data, self.Ai.ctypes.data, C.byref(self.settings), C.byref(self._h)))
        else:
            fr = np.ascontiguousarray(fixed_rows, dtype=np.int32)
            _lib.check(L.mpcqp_create_reduced(self.n, self.m, self.batch, self.Pp.ctypes.data, self.Pi.ctypes.data,
                                              self.Ap.ctypes.data, self.Ai.ctypes.data, len(fr), fr.ctypes.data, C.byref(self.settings), C.byref(self._h)))

    @property
    def nnzP(self):
        return int(self.Pp[-1])

    @property
    def nnzA(self):
        return int(self.Ap[-1])

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().mpcqp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def update(self, P, q, A, l, u):
        B = self.batch
        items = [_ptr_stride(P, self.nnzP, B, "P"), _ptr_stride(q, self.n, B, "q"), _ptr_stride(A, self.nnzA, B, "A"),
                 _ptr_stride(l, self.m, B, "l"), _ptr_stride(u, self.m, B, "u")]
        mems = {it[2] for it in items if it[2] is not None}
        if len(mems) != 1:
            raise ValueError("all of P, q, A, l, u must live in the same memory space")
        self._keep = [it[3] for it in items]
        args = []
        for ptr, stride, _, _ in items:
            args += [ptr, stride]
        _lib.check(_lib.lib().mpcqp_update(self._h, *args, mems.pop()))

    def set_dispatch_hint(self, enable=True):
        """longest-first dispatch order from the previous solve's iteration counts (on by default; changes no result)"""
        _lib.check(_lib.lib().mpcqp_set_dispatch_hint(self._h, 1 if enable else 0))

    def keep_workspace(self, enable=True):
        """keep scaling, factorisation and rho across solves so that update_vectors() can skip the setup"""
        _lib.check(_lib.lib().mpcqp_keep_workspace(self._h, 1 if enable else 0))

    def update_vectors(self, q, l, u):
        """replace q, l, u only (OSQP's osqp_update_data_vec on a kept workspace); needs keep_workspace() and a previous
        update() + solve()"""
        B = self.batch
        items = [_ptr_stride(q, self.n, B, "q"), _ptr_stride(l, self.m, B, "l"), _ptr_stride(u, self.m, B, "u")]
        mems = {it[2] for it in items if it[2] is not None}
        if len(mems) != 1:
            raise ValueError("q, l, u must live in the same memory space")
        self._keep_vec = [it[3] for it in items]
        args = []
        for ptr, stride, _, _ in items:
            args += [ptr, stride]
        _lib.check(_lib.lib().mpcqp_update_vectors(self._h, *args, mems.pop()))

    def warm_start(self, x0, y0):
        px, _, memx, kx = _ptr_stride(x0, self.n, self.batch, "x0")
        py, _, memy, ky = _ptr_stride(y0, self.m, self.batch, "y0")
        if memx != memy:
            raise ValueError("x0 and y0 must live in the same memory space")
        self._keep_ws = (kx, ky)
        _lib.check(_lib.lib().mpcqp_warm_start(self._h, px, py, memx))

    def set_rho(self, rho0):
        """per-instance starting rho [batch] (NumPy or CUDA tensor), None = settings.rho"""
        if rho0 is None:
            self._keep_rho = None
            _lib.check(_lib.lib().mpcqp_set_rho(self._h, None, _lib.MEM_HOST))
            return
        if _is_torch(rho0):
            import torch
            if rho0.dtype != torch.float64 or not rho0.is_contiguous() or tuple(rho0.shape) != (self.batch,):
                raise ValueError("rho0: expected a contiguous float64 tensor of %d values" % self.batch)
            self._keep_rho = rho0
            _lib.check(_lib.lib().mpcqp_set_rho(self._h, rho0.data_ptr(), _lib.MEM_DEVICE if rho0.is_cuda else _lib.MEM_HOST))
        else:
            a = np.ascontiguousarray(np.broadcast_to(np.asarray(rho0, dtype=np.float64), (self.batch,)))
            self._keep_rho = a
            _lib.check(_lib.lib().mpcqp_set_rho(self._h, a.ctypes.data, _lib.MEM_HOST))

    def solve(self, stream=None):
        _lib.check(_lib.lib().mpcqp_solve(self._h, stream))

    def solve_host(self, P, q, A, l, u, chunks=0, out=None):
        """fused, pipelined host-buffer step (mpcqp_solve_host): NumPy arrays or CPU torch tensors, ideally pinned; returns
        dict x, y, status, iters (pinned torch-backed NumPy arrays when torch is available, reused through `out`)"""
        B = self.batch
        items = [_ptr_stride(P, self.nnzP, B, "P"), _ptr_stride(q, self.n, B, "q"), _ptr_stride(A, self.nnzA, B, "A"),
                 _ptr_stride(l, self.m, B, "l"), _ptr_stride(u, self.m, B, "u")]
        if any(it[2] != _lib.MEM_HOST for it in items):
            raise ValueError("solve_host takes host arrays")
        if out is None:
            try:
                import torch
                mk = lambda shape, dt: torch.empty(shape, dtype=dt).pin_memory().numpy()
                out = dict(x=mk((B, self.n), torch.float64), y=mk((B, self.m), torch.float64), status=mk((B,), torch.int32), iters=mk((B,), torch.int32))
            except (ImportError, RuntimeError):
                out = dict(x=np.empty((B, self.n)), y=np.empty((B, self.m)), status=np.empty(B, np.int32), iters=np.empty(B, np.int32))
        args = []
        for ptr, stride, _, _ in items:
            args += [ptr, stride]
        _lib.check(_lib.lib().mpcqp_solve_host(self._h, *args, out["x"].ctypes.data, out["y"].ctypes.data, out["status"].ctypes.data,
                                               out["iters"].ctypes.data, int(chunks)))
        return out

    def sync(self):
        _lib.check(_lib.lib().mpcqp_sync(self._h))

    def get(self, want=("x", "y", "z", "status", "iters", "info")):
        B = self.batch
        out = {}
        if "x" in want: out["x"] = np.empty((B, self.n))
        if "y" in want: out["y"] = np.empty((B, self.m))
        if "z" in want: out["z"] = np.empty((B, self.m))
        if "status" in want: out["status"] = np.empty(B, dtype=np.int32)
        if "iters" in want: out["iters"] = np.empty(B, dtype=np.int32)
        if "info" in want: out["info"] = np.empty((B, 4))
        p = lambda k: out[k].ctypes.data if k in out else None
        _lib.check(_lib.lib().mpcqp_get(self._h, p("x"), p("y"), p("z"), p("status"), p("iters"), p("info"), _lib.MEM_HOST))
        if "info" in out:
            info = out.pop("info")
            out.update(obj=info[:, 0], prim_res=info[:, 1], dual_res=info[:, 2], rho=info[:, 3])
        return out

    def get_device(self, x=None, y=None, z=None, status=None, iters=None, info=None):
        """Copy results into caller-owned torch CUDA tensors (device-to-device, on the solve stream)."""
        p = lambda t: None if t is None else t.data_ptr()
        _lib.check(_lib.lib().mpcqp_get(self._h, p(x), p(y), p(z), p(status), p(iters), p(info), _lib.MEM_DEVICE))

    def last_kernel_ms(self):
        ms = C.c_float()
        _lib.check(_lib.lib().mpcqp_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def last_phase_ms(self):
        """(set-up kernel ms, iteration kernel ms) of the last solve where the handle runs them as two kernels (the on-chip mode), else (0, whole kernel)"""
        a, b = C.c_float(), C.c_float()
        _lib.check(_lib.lib().mpcqp_last_phase_ms(self._h, C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)

    def oc_info(self):
        """chain / hub shape of the on-chip mode's factor (all zero for other kernel families): what bench.py prices the chains with"""
        a = np.zeros(12, dtype=np.int64)
        _lib.check(_lib.lib().mpcqp_oc_info(self._h, a.ctypes.data))
        return dict(zip(["chain_blocks", "has_hub", "chain_e", "chain_f", "lds_blocks", "positions_per_wave", "hub_blocks_in_registers", "launch_pairs_for_rho_updates",
                         "slots_A", "slots_At", "slots_P", "chain_pairs"], a.tolist()))

    def plan_info(self):
        a = np.zeros(16, dtype=np.int64)
        _lib.check(_lib.lib().mpcqp_plan_info(self._h, a.ctypes.data))
        keys = ["n", "m", "batch", "npad", "mpad", "n_blocks", "L_blocks", "lds_bytes", "workspace_bytes_per_qp",
                "ordering", "nnzP_triu", "nnzA", "T_blocks", "factor_ops", "ell_slots", "variant"]
        d = dict(zip(keys, a.tolist()))
        d["tiles"] = d["factor_ops"] if d["variant"] >= 200 else 0      # on-chip kernels report their dense tiles of A in that slot
        return d

    def debug_scaling(self, b=0):
        D = np.empty(self.n); E = np.empty(self.m); c = np.empty(1)
        _lib.check(_lib.lib().mpcqp_debug_scaling(self._h, int(b), D.ctypes.data, E.ctypes.data, c.ctypes.data))
        return D, E, float(c[0])


def solve_local_system(ls, settings=None, **kw):
    """Convenience: solve a models.LocalSystem batch on the GPU, return the result dict."""
    qp = BatchQP(ls.n, ls.m, ls.batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai, settings, **kw)
    try:
        qp.update(ls.P, ls.q, ls.A, ls.l, ls.u)
        qp.solve()
        return qp.get()
    finally:
        qp.close()
