"""8192 quadrotor QPs (the bench workload) through the batched CuCaQP call sequence: setSystem / initSolver / solve / getSolution
(reference src/sqp_solver/CuCaQP.cpp:183-224, 271-288) for a whole batch at once.  Needs an MI355X."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from optimal_control_problem_amd import BatchQP, models  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
mdl, ls, meta = models.make_workload("quadrotor", batch)           # P, q, A, l, u in the reference's [p; x] formulation
qp = BatchQP(ls.n, ls.m, ls.batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai)     # eps_abs = eps_rel = 1e-3, max_iter 10000 as the reference sets them
for _ in range(3):
    t0 = time.perf_counter()
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); qp.sync()
    dt = time.perf_counter() - t0
res = qp.get()                                                     # x, y, z, status, iters, obj, prim_res, dual_res, rho
print("n = %d, m = %d, batch = %d: %.1f ms per batch including the host-to-device copies, kernel %.2f ms"
      % (ls.n, ls.m, batch, dt * 1e3, qp.last_kernel_ms()))
print("solved: %d of %d, ADMM iterations min / mean / max: %d / %.1f / %d"
      % ((res["status"] == 1).sum(), batch, res["iters"].min(), res["iters"].mean(), res["iters"].max()))
print("first instance, step of the second frame's inputs (the first frame is pinned):", np.round(res["x"][0, mdl.np + mdl.f + mdl.nx:mdl.np + 2 * mdl.f], 4))
qp.close()
