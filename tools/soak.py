#!/usr/bin/env python3
"""Soak: the same batch solved over and over on every kernel family; every result must be bitwise identical to the first
(exposes rare synchronisation hazards in the barrier-elided schedules).  usage: python tools/soak.py [seconds_per_case]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from optimal_control_problem_amd import models
from optimal_control_problem_amd.batch_qp import BatchQP

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
cases = [("quadrotor", 20, 8192, None), ("quadrotor", 20, 256, None), ("cartpole", 30, 8192, None), ("double_integrator", 20, 4096, None),
         ("quadrotor", 50, 2048, None), ("quadrotor", 10, 4096, None), ("cartpole", 50, 4096, None), ("cartpole", 20, 4096, None), ("double_integrator", 10, 8192, None),
         ("quadrotor", 10, 4096, "res4"), ("cartpole", 30, 4096, "res8"), ("quadrotor", 20, 2048, "stream"),
         ("quadrotor", 12, 4096, "oc4"), ("cartpole", 30, 4096, "oc4"), ("quadrotor", 20, 8192, "gres4"),
         ("quadrotor", 50, 2048, "oc8"), ("cartpole", 100, 4096, "oc8"), ("quadrotor", 30, 2048, "oc8")]
bad = 0
for name, N, B, variant in cases:
    if variant: os.environ["MPCQP_VARIANT"] = variant
    else: os.environ.pop("MPCQP_VARIANT", None)
    mdl, ls, _ = models.make_workload(name, B, N=N)
    d = [torch.as_tensor(a, device="cuda") for a in (ls.P, ls.q, ls.A, ls.l, ls.u)]
    qp = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    x = torch.empty(B, ls.n, dtype=torch.float64, device="cuda"); it = torch.empty(B, dtype=torch.int32, device="cuda")
    qp.update(*d); qp.solve(); qp.get_device(x=x, iters=it); x0, it0 = x.clone(), it.clone()
    t0 = time.time(); n = 0; diff = 0
    while time.time() - t0 < budget:
        for _ in range(20):
            qp.update(*d); qp.solve(); qp.get_device(x=x, iters=it)
            n += 1
            if not (torch.equal(x, x0) and torch.equal(it, it0)): diff += 1
    bad += diff
    print("%s N=%d batch=%d variant=%s (kernel %d): %d solves, %d differing" % (name, N, B, variant or "auto", qp.plan_info()["variant"], n, diff), flush=True)
    qp.close()
print("soak done, %d differing results" % bad)
sys.exit(1 if bad else 0)
