#!/usr/bin/env python3
"""Diagnostic: per-segment cycle breakdown of the resident kernel (needs libmpcqp_timing.so, see csrc/Makefile).
usage: MPCQP_LIB=optimal_control_problem_amd/libmpcqp_timing.so python tools/timing_breakdown.py [workload] [batch] [variant|-] [horizon]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

name = sys.argv[1] if len(sys.argv) > 1 else "quadrotor"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
if len(sys.argv) > 3 and sys.argv[3] != "-":
    os.environ["MPCQP_VARIANT"] = sys.argv[3]
horizon = int(sys.argv[4]) if len(sys.argv) > 4 else None
from optimal_control_problem_amd import _lib, models
from optimal_control_problem_amd.batch_qp import BatchQP

mdl, ls, _ = models.make_workload(name, batch, N=horizon) if horizon else models.make_workload(name, batch)
qp = BatchQP(ls.n, ls.m, batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
qp.update(ls.P, ls.q, ls.A, ls.l, ls.u)
for _ in range(2):
    qp.solve(); got = qp.get()
ms = qp.last_kernel_ms()
raw = np.zeros(batch * 16 + 128, np.int64); out = raw[:batch * 16].reshape(batch, 16)
L = _lib.lib()
L.mpcqp_debug_timing.argtypes = [C.c_void_p, C.c_void_p]
_lib.check(L.mpcqp_debug_timing(qp._h, raw.ctypes.data))
names = ["load", "ruiz", "apply-scale+init", "factor(first)", "At pass", "schedule(solve)", "A pass + x", "check", "store",
         "oc:F1 chains", "oc:F2+F3 hub", "oc:B1 diag+hub", "f:rho/dvec/T", "f:assemble", "f:LDL", "f:LDL sweeps"]
tot = out[:, :9].sum(axis=1).mean()
print("variant", qp.plan_info()["variant"], "kernel %.2f ms for %d QPs, mean iters %.1f, mean cycles/QP %.0f (%s ticks: the shader clock)" % (
    ms, batch, got["iters"].mean(), tot, "s_memtime"))
for k, nm in enumerate(names):
    if nm != "-" and out[:, k].mean() > 0:
        print("  %-20s %10.0f cyc  %5.1f %%" % (nm, out[:, k].mean(), 100 * out[:, k].mean() / tot))
it = got["iters"].mean()
print("  per ADMM iteration: At %.0f, solve %.0f, A+x %.0f cycles" % (out[:, 4].mean() / it, out[:, 5].mean() / it, out[:, 6].mean() / it))

tr = raw[batch * 16:]
if tr[127] > 0:
    print("  workgroup 0: %d s_memtime ticks in %d ticks of the constant 100 MHz counter: s_memtime runs at %.0f MHz" % (tr[126], tr[127], 100.0 * tr[126] / tr[127]))
if tr[0] > 0:
    print("  wave-0 solve: %d cycles; per segment (cycles, flags[T=1,SET=2,EACH=4,END=8,BAR=16,NOP=32], ops):" % (tr[1] - tr[0]))
    prev = tr[0]; out_ = []
    for g in range(60):
        t_, meta = tr[2 + 2 * g], tr[3 + 2 * g]
        if t_ <= 0: break
        out_.append((int(t_ - prev), int(meta >> 32), int(meta & 0xffffffff))); prev = t_
    print("   ", out_)
