#!/usr/bin/env python3
"""Diagnostic: the parts of one Ruiz pass of the set-up kernel (needs a library built with -DMPCQP_TIMING -DMPCQP_TIMING_RUIZ).
usage: MPCQP_LIB=tools/probes/bin/libmpcqp_tr.so python tools/ruiz_breakdown.py [workload] [batch] [horizon]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

name = sys.argv[1] if len(sys.argv) > 1 else "quadrotor"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
horizon = int(sys.argv[3]) if len(sys.argv) > 3 else None
from optimal_control_problem_amd import _lib, models
from optimal_control_problem_amd.batch_qp import BatchQP

mdl, ls, _ = models.make_workload(name, batch, N=horizon) if horizon else models.make_workload(name, batch)
qp = BatchQP(ls.n, ls.m, batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
qp.update(ls.P, ls.q, ls.A, ls.l, ls.u)
for _ in range(2):
    qp.solve(); got = qp.get()
raw = np.zeros(batch * 16 + 128, np.int64); out = raw[:batch * 16].reshape(batch, 16)
L = _lib.lib()
L.mpcqp_debug_timing.argtypes = [C.c_void_p, C.c_void_p]
_lib.check(L.mpcqp_debug_timing(qp._h, raw.ctypes.data))
m = out.mean(axis=0)
print("%s N=%s x %d, variant %d: set-up phases, cycles per QP (shader clock)" % (name, horizon, batch, qp.plan_info()["variant"]))
print("  load %.0f, Ruiz %.0f, scale + write %.0f, factorisation %.0f" % (m[0], m[1], m[2], m[3]))
names = ["sweep over A (row norms, column atomics)", "barrier behind it", "cost scale + D update", "sweep over P + cost terms (no barrier behind it)"]
for k, nm in enumerate(names):
    print("  Ruiz, 10 passes: %-45s %8.0f  (%.0f per pass)" % (nm, m[9 + k], m[9 + k] / 10))
print("  set-up %.3f ms, iteration %.3f ms" % qp.last_phase_ms())
