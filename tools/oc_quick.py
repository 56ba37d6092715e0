import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MPCQP_VARIANT"] = "oc4"
from optimal_control_problem_amd import models
from optimal_control_problem_amd.batch_qp import BatchQP
from tests.support import problems
for name, B, N in [("quadrotor", 24, 20), ("cartpole", 6, 30), ("double_integrator", 40, 20)]:
    mdl, ls, _ = models.make_workload(name, B, N=N)
    qp = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    print(name, qp.plan_info())
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); got = qp.get(); qp.close()
    ref = problems.oracle_solve(ls)
    print(" status", got["status"][:8], ref["status"][:8]); print(" iters", got["iters"][:8], ref["iters"][:8])
    fin = np.isfinite(ref["x"]) & np.isfinite(got["x"])
    print(" max|dx|", np.abs(got["x"][fin]-ref["x"][fin]).max() if fin.any() else None, "nan frac", 1-np.isfinite(got["x"]).mean())
