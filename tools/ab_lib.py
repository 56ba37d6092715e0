#!/usr/bin/env python3
"""A/B of two builds of libmpcqp.so on the same box: kernel time of the bench workloads through a minimal ctypes binding
(only the entry points every build has).  usage: python tools/ab_lib.py libA.so libB.so ..."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from optimal_control_problem_amd import models
from optimal_control_problem_amd._lib import Settings

torch.zeros(1, device="cuda")
cases = [("quadrotor", 20, 8192), ("cartpole", 30, 8192), ("double_integrator", 20, 4096)]
if os.environ.get("AB_CASES"):      # e.g. AB_CASES=quadrotor:50:4096,cartpole:100:8192
    cases = [(c.split(":")[0], int(c.split(":")[1]), int(c.split(":")[2])) for c in os.environ["AB_CASES"].split(",")]
data = {}
for name, N, B in cases:
    mdl, ls, _ = models.make_workload(name, B, N=N)
    data[(name, N, B)] = (ls, [torch.as_tensor(a, device="cuda") for a in (ls.P, ls.q, ls.A, ls.l, ls.u)])
for rep in range(2):
    for path in sys.argv[1:]:
        L = C.CDLL(os.path.abspath(path))
        vp = C.c_void_p
        L.mpcqp_create.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.POINTER(vp)]
        L.mpcqp_update.argtypes = [vp] + [vp, C.c_long] * 5 + [C.c_int]
        L.mpcqp_solve.argtypes = [vp, vp]; L.mpcqp_sync.argtypes = [vp]; L.mpcqp_destroy.argtypes = [vp]
        L.mpcqp_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
        out = []
        for key, (ls, d) in data.items():
            h = vp()
            assert L.mpcqp_create(ls.n, ls.m, key[2], ls.Pp.ctypes.data, ls.Pi.ctypes.data, ls.Ap.ctypes.data, ls.Ai.ctypes.data, None, C.byref(h)) == 0
            ms = []
            for _ in range(6):
                assert L.mpcqp_update(h, d[0].data_ptr(), ls.P.shape[1], d[1].data_ptr(), ls.n, d[2].data_ptr(), ls.A.shape[1], d[3].data_ptr(), ls.m, d[4].data_ptr(), ls.m, 1) == 0
                assert L.mpcqp_solve(h, None) == 0 and L.mpcqp_sync(h) == 0
                t = C.c_float(); L.mpcqp_last_kernel_ms(h, C.byref(t)); ms.append(t.value)
            L.mpcqp_destroy(h)
            out.append("%s N=%d: %.3f ms" % (key[0], key[1], min(ms[2:])))
        print(os.path.basename(path), "|", " | ".join(out))
