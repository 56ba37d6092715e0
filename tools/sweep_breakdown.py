#!/usr/bin/env python3
"""Diagnostic: where the A' sweep of an ADMM iteration spends its time (needs a library built with -DMPCQP_TIMING -DMPCQP_TIMING_SWEEP).
usage: MPCQP_LIB=tools/probes/bin/libmpcqp_ts.so python tools/sweep_breakdown.py [workload] [batch] [horizon]"""
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
it = got["iters"].astype(float)
per = lambda k: (out[:, k] / it).mean()
print("%s N=%s x %d, variant %d, mean iters %.1f; cycles per ADMM iteration (shader clock)" % (name, horizon, batch, qp.plan_info()["variant"], it.mean()))
print("  A' phase as wave 0 sees it between the barriers: %.0f" % per(4))
print("  wave 0, first chunk: issue of the loads %.0f, wait for them %.0f, gathers + fma + store %.0f; its other chunks %.0f" % (per(9), per(10), per(11), per(12)))
print("  wave NW-1: loads issued + landed %.0f, the rest of its sweep %.0f" % (per(14), per(15)))
print("  solve %.0f, A sweep + x %.0f" % (per(5), per(6)))
