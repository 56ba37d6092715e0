import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
from optimal_control_problem_amd import models
from optimal_control_problem_amd.batch_qp import BatchQP
for name, N, B in (("double_integrator", 120, 8192), ("double_integrator", 100, 8192), ("cartpole", 100, 8192)):
    mdl, ls, _ = models.make_workload(name, B, N=N)
    qp = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); g = qp.get()
    print(name, N, "n", ls.n, "m", ls.m, qp.plan_info(), qp.oc_info(), "ms", qp.last_phase_ms(), "iters mean", g["iters"].mean(), "max", g["iters"].max(), np.bincount(g["iters"] // 25)[:12])
    qp.close()
