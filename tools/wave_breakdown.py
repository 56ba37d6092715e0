#!/usr/bin/env python3
"""Diagnostic: the two sweeps of the iteration kernel wave by wave (needs a library built with -DMPCQP_TIMING -DMPCQP_TIMING_WAVES=0 for the A' sweep,
=1 for the A sweep): cycles per ADMM iteration each wave spends on its own chunks, and waiting at the barrier behind the sweep.
usage: MPCQP_LIB=tools/probes/bin/libmpcqp_w0.so python tools/wave_breakdown.py [workload] [batch] [horizon]"""
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
nw = qp.plan_info()["variant"] % 100
per = out / np.maximum(got["iters"], 1)[:, None]
m = per.mean(axis=0)
print("%s N=%s x %d, variant %d, %s: cycles per ADMM iteration, wave by wave" % (name, horizon, batch, qp.plan_info()["variant"], os.path.basename(os.environ.get("MPCQP_LIB", "libmpcqp.so"))))
print("  own chunks of the sweep : " + " ".join("%7.0f" % v for v in m[:nw]))
print("  wait at the barrier     : " + " ".join("%7.0f" % v for v in m[8:8 + nw]))
print("  set-up %.3f ms, iteration %.3f ms" % qp.last_phase_ms())
