#!/usr/bin/env python3
"""Generates tests/golden/qp_fixtures.npz (committed).  Run from the repo root: python tests/golden/make_golden.py

The reference holds no machine-checkable vectors at the QP boundary ("parity unpinned"), and
it cannot be built or imported here (C++ with CasADi/Eigen/OSQP/ROS 2 dependencies that are not installed).
So each fixture carries
  * the QP data (P, q, A, l, u) exactly as CuCaQP::setSystem would receive it
    (reference src/sqp_solver/CuCaQP.cpp:271-288; formulation of SQPOptimizationSolver.cpp:47-120),
  * x_star / y_star: an optimum obtained independently of any ADMM run-time choice -- active set read off a
    1e-10-accurate solve, equality-constrained KKT system solved by dense least squares, and the KKT
    conditions (primal/dual feasibility, stationarity, complementarity) verified to 1e-8,
  * pins of the CPU oracle at the reference's settings (x, y, iterations, status), so that a change in the
    oracle's iterate sequence is noticed,
and, for the reference's own test/test.cpp cases 1-7, the analytic optimum printed there (test/test.cpp:13-185).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from optimal_control_problem_amd import models  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.support import problems  # noqa: E402


def kkt_optimum(ls, b):
    """KKT-verified optimum of instance b, or None for infeasible / non-convex instances."""
    P, A = ls.dense(b)
    P = np.triu(P) + np.triu(P, 1).T
    q, l, u = ls.q[b], np.maximum(ls.l[b], -1e30), np.minimum(ls.u[b], 1e30)
    one = models.LocalSystem(ls.n, ls.m, ls.Pp, ls.Pi, ls.Ap, ls.Ai, ls.P[b:b + 1], ls.q[b:b + 1], ls.A[b:b + 1], ls.l[b:b + 1], ls.u[b:b + 1])
    r = problems.oracle_solve(one, eps_abs=1e-10, eps_rel=1e-10, max_iter=400000)
    if r["status"][0] != 1:
        return None
    x, y = r["x"][0], r["y"][0]
    eq = (u - l) < 1e-9
    lo = (~eq) & (y < -1e-7) & (l > -1e29)
    up = (~eq) & (y > 1e-7) & (u < 1e29)
    act = eq | lo | up
    bnd = np.where(up, u, l)[act]
    Aa = A[act]
    na = Aa.shape[0]
    K = np.block([[P, Aa.T], [Aa, np.zeros((na, na))]])
    rhs = np.concatenate([-q, bnd])
    sol = np.linalg.lstsq(K, rhs, rcond=None)[0]
    xs = sol[:ls.n]; lam = sol[ls.n:]
    ys = np.zeros(ls.m); ys[act] = lam
    ax = A @ xs
    scale = 1.0 + max(np.abs(xs).max(), np.abs(q).max())
    ok = (ax >= l - 1e-8 * scale).all() and (ax <= u + 1e-8 * scale).all()
    ok &= np.abs(P @ xs + q + A.T @ ys).max() <= 1e-8 * scale
    ok &= (ys[lo] <= 1e-8).all() and (ys[up] >= -1e-8).all()
    ok &= np.abs(xs - x).max() <= 1e-5 * scale
    if not ok:
        raise RuntimeError("KKT verification failed")
    return xs, ys


def main():
    fixtures = []
    for mdl, arg, expected in models.reference_test_cases():
        fixtures.append(("testcpp_" + mdl.name, problems.toy_local_system(mdl, arg), expected))
    for seed in range(6):
        fixtures.append(("random_%d" % seed, problems.random_qp(11 + 3 * seed, 17 + 5 * seed, seed), None))
    fixtures.append(("primal_infeasible", problems.random_qp(12, 20, 101, infeasible="primal"), None))
    fixtures.append(("dual_infeasible", problems.random_qp(12, 20, 102, infeasible="dual"), None))
    for name, batch, N in (("double_integrator", 4, None), ("quadrotor", 2, None), ("cartpole", 1, 40)):
        mdl, ls, _ = models.make_workload(name, batch, N=N)
        fixtures.append((name, ls, None))
    out = {"names": np.array([f[0] for f in fixtures])}
    for name, ls, expected in fixtures:
        B = ls.batch
        out[name + "/dims"] = np.array([ls.n, ls.m, B, ls.np])
        for k in ("Pp", "Pi", "Ap", "Ai", "P", "q", "A", "l", "u"):
            out[name + "/" + k] = getattr(ls, k)
        ref = problems.oracle_solve(ls)
        for k in ("x", "y", "iters", "status"):
            out[name + "/oracle_" + k] = ref[k]
        xs = np.full((B, ls.n), np.nan); ys = np.full((B, ls.m), np.nan)
        for b in range(B):
            opt = kkt_optimum(ls, b) if ref["status"][b] == 1 else None
            if opt is not None:
                xs[b], ys[b] = opt
        out[name + "/x_star"] = xs; out[name + "/y_star"] = ys
        if expected is not None:
            out[name + "/analytic"] = np.asarray(expected, float)
        print("%-22s n=%4d m=%4d B=%d status=%s iters=%s kkt=%s" % (name, ls.n, ls.m, B, ref["status"].tolist(), ref["iters"].tolist(),
                                                                   "ok" if np.isfinite(xs).all() else "none"))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "qp_fixtures.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
