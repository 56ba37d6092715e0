#!/usr/bin/env python3
"""Same-box A/B of one environment switch on several workloads: kernel time with and without it, results compared bitwise.
usage (GPU box): python tools/ab_env.py MPCQP_NO_PW [quadrotor:20:8192 quadrotor:50:8192 cartpole:100:16384 ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from optimal_control_problem_amd import models
from optimal_control_problem_amd.batch_qp import BatchQP

var = sys.argv[1]
cases = sys.argv[2:] or ["quadrotor:20:8192", "quadrotor:50:8192", "cartpole:100:16384", "quadrotor:10:8192", "cartpole:30:8192"]
dev = torch.device("cuda", 0)
for cs in cases:
    name, N, B = cs.split(":"); N = int(N); B = int(B)
    mdl, ls, _ = models.make_workload(name, B, seed={"quadrotor": 2024, "cartpole": 7, "double_integrator": 1234}[name], N=N)
    d = [torch.from_numpy(a).to(dev) for a in (ls.P, ls.q, ls.A, ls.l, ls.u)]
    res = {}
    for tag, on in (("default", False), (var + "=1", True), ("default again", False)):
        if on:
            os.environ[var] = "1"
        qp = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
        qp.set_dispatch_hint(False)
        ms = []
        for _ in range(5):
            qp.update(*d); qp.solve(); torch.cuda.synchronize(); ms.append(qp.last_kernel_ms())
        got = qp.get(); v = qp.plan_info()["variant"]; qp.close()
        os.environ.pop(var, None)      # (only now: some switches are read at every solve)
        res[tag] = got
        print("%-18s N=%3d x %5d %-16s variant %3d: kernel %.3f ms (min of %s)" % (name, N, B, tag, v, min(ms[1:]), " ".join("%.3f" % m for m in ms)), flush=True)
    a, b = res["default"], res[var + "=1"]
    print("   bitwise equal x: %s, iters: %s, status: %s" % (np.array_equal(a["x"], b["x"], equal_nan=True), np.array_equal(a["iters"], b["iters"]), np.array_equal(a["status"], b["status"])), flush=True)
