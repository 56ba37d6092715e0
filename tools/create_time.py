#!/usr/bin/env python3
"""host cost of mpcqp_create (plan building + uploads) per workload: python tools/create_time.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimal_control_problem_amd import models
from optimal_control_problem_amd.batch_qp import BatchQP

import torch
torch.zeros(1, device="cuda")
for name, N in [("double_integrator", 20), ("quadrotor", 20), ("cartpole", 30), ("quadrotor", 50), ("cartpole", 100), ("quadrotor", 100), ("cartpole", 200)]:
    mdl, ls, _ = models.make_workload(name, 4, N=N)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        qp = BatchQP(ls.n, ls.m, 1024, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
        ts.append(time.perf_counter() - t0)
        qp.close()
    print("%s N=%d n=%d m=%d: mpcqp_create %.1f ms (first %.1f ms)" % (name, N, ls.n, ls.m, min(ts) * 1e3, ts[0] * 1e3), flush=True)
