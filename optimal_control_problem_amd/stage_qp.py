"""StageQP: the structured stage form of the C ABI (include/mpcqp.h "Structured stage form", csrc/stageqp.hip) -- a batch of OCP-structured
QPs given by their stage blocks instead of CSC value arrays.

    w = [p; frame_0; ...; frame_{N-1}],  frame_k = [s_k; u_k]          (reference src/OCP_config/OCPConfig.cpp:29-46,102: stage-interleaved)
    rows [p; frames; dynamics],  dynamics row block k:  lg_k <= s_{k+1} - A_k s_k - B_k u_k <= ug_k
                                                                         (reference src/sqp_solver/SQPOptimizationSolver.cpp:47-77)

A StageQP IS a BatchQP on the pattern `stage_pattern` returns (solve / get / warm_start / set_rho / keep_workspace / update_vectors / plan_info
are BatchQP's); only the matrices arrive differently: update_blocks(H, Hp, Hpp, AB, q, l, u)."""
import ctypes as C

import numpy as np

from . import _lib
from .batch_qp import BatchQP, _is_torch


class StageDims(C.Structure):
    _fields_ = [("N", C.c_int), ("nx", C.c_int), ("nu", C.c_int), ("np", C.c_int), ("cost_mask", C.c_void_p), ("dyn_mask", C.c_void_p)]


def _dims(N, nx, nu, np_, cost_mask, dyn_mask):
    f = nx + nu
    keep = []
    d = StageDims(int(N), int(nx), int(nu), int(np_), None, None)
    if cost_mask is not None:
        cm = np.ascontiguousarray(np.asarray(cost_mask) != 0, dtype=np.uint8)
        if cm.shape != (f + np_, f + np_):
            raise ValueError("cost_mask: expected shape (%d, %d) over [s; u; p]" % (f + np_, f + np_))
        keep.append(cm); d.cost_mask = cm.ctypes.data
    if dyn_mask is not None:
        dm = np.ascontiguousarray(np.asarray(dyn_mask) != 0, dtype=np.uint8)
        if dm.shape != (nx, f):
            raise ValueError("dyn_mask: expected shape (%d, %d) over [A_k B_k]" % (nx, f))
        keep.append(dm); d.dyn_mask = dm.ctypes.data
    return d, keep


def stage_pattern(N, nx, nu, np_=0, cost_mask=None, dyn_mask=None):
    """-> n, m, Pp, Pi, Ap, Ai of the stage form (host only: no GPU needed)"""
    L = _lib.lib()
    d, keep = _dims(N, nx, nu, np_, cost_mask, dyn_mask)
    sizes = np.zeros(4, np.int32)
    _lib.check(L.mpcqp_stageqp_pattern(C.byref(d), sizes.ctypes.data, None, None, None, None))
    n, m, nnzP, nnzA = (int(v) for v in sizes)
    Pp = np.zeros(n + 1, np.int32); Pi = np.zeros(max(nnzP, 1), np.int32); Ap = np.zeros(n + 1, np.int32); Ai = np.zeros(max(nnzA, 1), np.int32)
    _lib.check(L.mpcqp_stageqp_pattern(C.byref(d), None, Pp.ctypes.data, Pi.ctypes.data, Ap.ctypes.data, Ai.ctypes.data))
    return n, m, Pp, Pi[:nnzP], Ap, Ai[:nnzA]


def blocks_from_dense(Pd, Ad, N, nx, nu, np_=0):
    """the stage blocks of dense matrices in the stage form's variable / row order (what a caller that holds blocks would pass directly):
    Pd [B, n, n] symmetric, Ad [B, m, n] -> H [B, N, f, f], Hp [B, N, np, f], Hpp [B, np, np], AB [B, N-1, nx, f]"""
    Pd = np.asarray(Pd, float); Ad = np.asarray(Ad, float)
    B = Pd.shape[0]; f = nx + nu; n = np_ + N * f
    H = np.zeros((B, N, f, f)); Hp = np.zeros((B, N, np_, f)); AB = np.zeros((B, max(N - 1, 1), nx, f))
    for k in range(N):
        sl = slice(np_ + k * f, np_ + (k + 1) * f)
        H[:, k] = Pd[:, sl, sl]
        Hp[:, k] = Pd[:, :np_, sl]
        if k < N - 1:
            AB[:, k] = -Ad[:, n + k * nx:n + (k + 1) * nx, sl]
    return H, Hp, np.ascontiguousarray(Pd[:, :np_, :np_]), AB[:, :N - 1]


class StageQP(BatchQP):
    def __init__(self, N, nx, nu, batch, np_=0, cost_mask=None, dyn_mask=None, settings=None, **kw):
        self.N, self.nx, self.nu, self.np = int(N), int(nx), int(nu), int(np_)
        self.f = self.nx + self.nu
        self.batch = int(batch)
        self.settings = settings if settings is not None else _lib.default_settings(**kw)
        self.n, self.m, self.Pp, self.Pi, self.Ap, self.Ai = stage_pattern(N, nx, nu, np_, cost_mask, dyn_mask)
        d, keep = _dims(N, nx, nu, np_, cost_mask, dyn_mask)
        self._sq = C.c_void_p()
        self._h = C.c_void_p()
        self._keep = []
        L = _lib.lib()
        _lib.check(L.mpcqp_stageqp_create(C.byref(d), self.batch, C.byref(self.settings), C.byref(self._sq)))
        self._h = C.c_void_p(L.mpcqp_stageqp_handle(self._sq))

    def close(self):
        if getattr(self, "_sq", None) is not None and self._sq.value:
            _lib.lib().mpcqp_stageqp_destroy(self._sq)       # (destroys the handle underneath as well)
            self._sq = C.c_void_p(); self._h = C.c_void_p()

    def update(self, P, q, A, l, u):
        raise TypeError("a StageQP takes its matrices in blocks: update_blocks(H, Hp, Hpp, AB, q, l, u)")

    def update_blocks(self, H, Hp, Hpp, AB, q, l, u, stream=None):
        """H [B, N, f, f], Hp [B, N, np, f] and Hpp [B, np, np] (None when np = 0), AB [B, N-1, nx, f], q [B, n], l, u [B, m]: NumPy arrays (copied)
        or torch CUDA tensors (borrowed; the gather kernel runs on `stream`, the stream of the solve)"""
        B, N, f, npar, nx = self.batch, self.N, self.f, self.np, self.nx
        want = [("H", H, (B, N, f, f)), ("Hp", Hp, (B, N, npar, f)), ("Hpp", Hpp, (B, npar, npar)), ("AB", AB, (B, N - 1, nx, f)),
                ("q", q, (B, self.n)), ("l", l, (B, self.m)), ("u", u, (B, self.m))]
        ptrs, mems, keep = [], set(), []
        for name, a, shape in want:
            if a is None:
                if name in ("Hp", "Hpp") and npar == 0:
                    ptrs.append(None); continue
                raise ValueError("%s is missing" % name)
            if _is_torch(a):
                import torch
                if a.dtype != torch.float64 or not a.is_contiguous() or tuple(a.shape) != shape:
                    raise ValueError("%s: expected a contiguous float64 tensor of shape %s (dimension mismatch)" % (name, shape))
                mems.add(_lib.MEM_DEVICE if a.is_cuda else _lib.MEM_HOST); ptrs.append(a.data_ptr()); keep.append(a)
            else:
                b = np.ascontiguousarray(a, dtype=np.float64)
                if b.shape != shape:
                    raise ValueError("%s: expected shape %s, got %s (dimension mismatch)" % (name, shape, b.shape))
                mems.add(_lib.MEM_HOST); ptrs.append(b.ctypes.data if b.size else None); keep.append(b)
        if len(mems) != 1:
            raise ValueError("all blocks and vectors must live in the same memory space")
        if npar == 0:
            ptrs[1] = ptrs[2] = None
        self._keep = keep
        _lib.check(_lib.lib().mpcqp_stageqp_update(self._sq, *ptrs, mems.pop(), stream))
