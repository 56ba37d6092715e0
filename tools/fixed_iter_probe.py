#!/usr/bin/env python3
"""Diagnostic: kernel time of a FIXED number of ADMM iterations (no termination, no rho updates), for timing experiments whose results need not be right.
usage: MPCQP_LIB=<build> python tools/fixed_iter_probe.py [workload] [batch] [horizon] [iterations]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

name = sys.argv[1] if len(sys.argv) > 1 else "quadrotor"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
horizon = int(sys.argv[3]) if len(sys.argv) > 3 else 20
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 50
from optimal_control_problem_amd import _lib, models
from optimal_control_problem_amd.batch_qp import BatchQP

mdl, ls, _ = models.make_workload(name, batch, N=horizon)
st = _lib.default_settings(max_iter=iters, eps_abs=0.0, eps_rel=0.0, eps_prim_inf=0.0, eps_dual_inf=0.0, adaptive_rho=0, check_termination=0)
qp = BatchQP(ls.n, ls.m, batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai, settings=st)
qp.update(ls.P, ls.q, ls.A, ls.l, ls.u)
best = None
for _ in range(4):
    qp.solve(); qp.sync()
    ms = qp.last_phase_ms()
    best = ms if best is None or ms[1] < best[1] else best
got = qp.get()
print("%s N=%d x %d, %s: %d iterations each (mean %.1f): set-up %.3f ms, iteration kernel %.3f ms = %.1f us per iteration of the batch"
      % (name, horizon, batch, os.path.basename(os.environ.get("MPCQP_LIB", "libmpcqp.so")), iters, got["iters"].mean(), best[0], best[1], 1e3 * best[1] / iters))
