"""CPU: the batched SQP outer loop (host logic mirroring reference SQPOptimizationSolver.cpp:127-216), with the
CPU oracle standing in for the GPU QP backend, against the reference's own test/test.cpp known answers."""
import numpy as np
import pytest

from optimal_control_problem_amd import models
from optimal_control_problem_amd.sqp import SQPOptimizationSolver
from tests.support.oracle_backend import OracleCuCaQP


def _arg(arg):
    return {k: np.asarray(v, float) for k, v in arg.items()}


@pytest.mark.parametrize("idx", range(7))
def test_testcpp_known_answers_full_steps(built, idx):
    """alpha = 1 (the semantics of the one-argument constructor test/test.cpp was written for): the SQP lands on the
    expected optimum printed in test/test.cpp:34,57,82,108,134,159,183"""
    mdl, arg, expected = models.reference_test_cases()[idx]
    s = SQPOptimizationSolver(mdl, {"max_iter": 3, "alpha": 1.0, "verbose": False}, qp_solver=OracleCuCaQP())
    res = s.getOptimalSolution(_arg(arg))
    assert np.abs(res["x"][0] - np.asarray(expected)).max() < 5e-3
    assert abs(res["f"][0] - mdl.f(np.concatenate([np.asarray(arg["p"], float), res["x"][0]]))) < 1e-12


def test_damped_fixed_count_steps(built):
    """alpha = 0.1, 10 steps (reference defaults, include/optimal_control_problem/OptimalControlProblem.h:25-26):
    a pure QP gets 1 - 0.9^10 of the way, there is no convergence test"""
    mdl, arg, expected = models.reference_test_cases()[1]
    s = SQPOptimizationSolver(mdl, {"max_iter": 10, "alpha": 0.1, "verbose": False}, qp_solver=OracleCuCaQP())
    res = s.getOptimalSolution(_arg(arg))
    assert np.abs(res["x"][0] - (1 - 0.9 ** 10) * np.asarray(expected)).max() < 5e-3
    # the iterate persists across calls and x0 is ignored (SQPOptimizationSolver.cpp:88-91,117)
    res2 = s.getOptimalSolution(dict(_arg(arg), x0=np.array([100.0, 100.0])))
    assert np.abs(res2["x"][0] - (1 - 0.9 ** 20) * np.asarray(expected)).max() < 5e-3


def test_parameter_is_carried_as_qp_variable(built):
    """case 6: p enters as variable with 0 <= dp <= 0 rows and is sliced off the step (SQPOptimizationSolver.cpp:175-176)"""
    mdl, arg, expected = models.reference_test_cases()[5]
    s = SQPOptimizationSolver(mdl, {"max_iter": 2, "alpha": 1.0}, qp_solver=OracleCuCaQP())
    res = s.getOptimalSolution(_arg(arg))
    assert res["x"].shape == (1, 2) and np.abs(res["x"][0] - [5.0, 0.0]).max() < 5e-3


def test_batched_double_integrator_mpc_tick(built):
    """horizon 20 double-integrator OCP, batch of independent instances, first frame pinned
    (reference src/OptimalControlProblem.cpp:93-96): dynamics are linear so one full step satisfies them"""
    B = 6
    mdl, ls, meta = models.make_workload("double_integrator", B)
    s = SQPOptimizationSolver(mdl, {"max_iter": 2, "alpha": 1.0}, batch=B, qp_solver=OracleCuCaQP(batch=B))
    arg = dict(lbx=meta["lbx"], ubx=meta["ubx"], lbg=meta["lbg"], ubg=meta["ubg"], p=meta["p"])
    res = s.getOptimalSolution(arg)
    assert np.abs(mdl.constraints(res["x"])).max() < 5e-3
    assert np.abs(res["x"][:, :3] - meta["frame0"]).max() < 5e-3
    lo, hi = mdl.frame_bounds()
    X = res["x"].reshape(B, mdl.N, mdl.f)
    assert (X[:, 1:] >= lo - 1e-2).all() and (X[:, 1:] <= hi + 1e-2).all()


def test_model_derivatives_match_finite_differences():
    """the hand-written local systems stand in for CasADi's AD (reference AutoDifferentiator.cpp:16-28)"""
    rng = np.random.default_rng(0)
    for mdl in (models.DoubleIntegrator(5), models.Quadrotor(4), models.CartPole(5)):
        B = 2
        p = rng.normal(size=(B, mdl.np)); x = rng.normal(size=(B, mdl.nvar)) * 0.3
        lbx = np.full((B, mdl.nvar), -1.0); ubx = np.full((B, mdl.nvar), 1.0)
        z = np.zeros((B, mdl.ng))
        ls = mdl.local_system(p, x, lbx, ubx, z, z)
        Pd, Ad = ls.dense(1)
        w = np.concatenate([p[1], x[1]])
        fun = lambda w_: mdl.objective(w_[None, :mdl.np], w_[None, mdl.np:])[0]
        con = lambda w_: np.concatenate([w_, mdl.constraints(w_[None, mdl.np:])[0]])
        h = 1e-6
        g = np.array([(fun(w + h * e) - fun(w - h * e)) / (2 * h) for e in np.eye(len(w))])
        J = np.array([(con(w + h * e) - con(w - h * e)) / (2 * h) for e in np.eye(len(w))]).T
        assert np.abs(g - ls.q[1]).max() < 1e-5 * (1 + np.abs(g).max())
        assert np.abs(J - Ad).max() < 1e-6 * (1 + np.abs(J).max())
        assert np.abs(Pd - Pd.T).max() == 0                       # both triangles, like CasADi's hessian()
        e0 = np.zeros(len(w)); e0[mdl.np] = 1
        q2 = mdl.local_system(p, x + h * e0[None, mdl.np:] * np.array([[0], [1]]), lbx, ubx, z, z).q[1]
        assert np.abs((q2 - ls.q[1]) / h - Pd[:, mdl.np]).max() < 1e-4 * (1 + np.abs(Pd).max())
        # bounds are shifted by the current value of [p; x; g] (SQPOptimizationSolver.cpp:66-71)
        assert np.abs(ls.l[1] - (np.concatenate([p[1], lbx[1], z[1]]) - con(w))).max() < 1e-12


def test_warm_start_and_rho_carry_over(built):
    """SURVEY.md section 8 row f2 on the host loop with the oracle backend: ADMM started from the previous SQP iteration's
    (x, y), and additionally from its adapted rho, needs fewer ADMM iterations and reaches the same objective"""
    from optimal_control_problem_amd.sqp import SQPOptimizationSolver
    from tests.support.oracle_backend import OracleCuCaQP
    B = 4
    mdl, ls, meta = models.make_workload("cartpole", B, N=30)
    arg = dict(lbx=meta["lbx"], ubx=meta["ubx"], lbg=meta["lbg"], ubg=meta["ubg"], p=meta["p"])
    runs = {}
    for key, opt in (("cold", {}), ("warm", {"warm_start_admm": True}), ("warm+rho", {"warm_start_admm": True, "carry_rho": True})):
        s = SQPOptimizationSolver(mdl, dict({"max_iter": 6, "alpha": 0.5}, **opt), batch=B, qp_solver=OracleCuCaQP(batch=B))
        s.setInitialGuess(meta["x_iterate"])
        r = s.getOptimalSolution(arg)
        runs[key] = (r, np.stack(s.admm_iterations).sum())
    assert runs["warm"][1] < runs["cold"][1] and runs["warm+rho"][1] <= runs["warm"][1]
    for key in ("warm", "warm+rho"):
        assert np.abs(runs[key][0]["f"] - runs["cold"][0]["f"]).max() <= 2e-2 * np.abs(runs["cold"][0]["f"]).max()


def test_per_frame_weights_host_formulation():
    """terminal costs / ramps: Q, R given per frame [N, nx], [N, nu]; gradient and Hessian of the stage cost against central
    differences of the objective in the augmented variables w = [p; x]"""
    N = 6
    rng = np.random.default_rng(0)
    Qk = rng.uniform(0.1, 5.0, (N, 4)); Rk = rng.uniform(0.01, 1.0, (N, 1)); Qk[-1] *= 20.0        # heavy terminal weight
    class CP(models.CartPole):
        def __init__(self): models.StageOCP.__init__(self, N, 0.02, Qk, Rk)
    mdl = CP()
    assert mdl.varying_weights
    B = 2
    x = rng.normal(0, 0.5, (B, mdl.nvar)); p = rng.normal(0, 0.3, (B, 4))
    lbx, ubx, lbg, ubg = mdl.stacked_bounds(x[:, :mdl.f].copy())
    ls = mdl.local_system(p, x, lbx, ubx, lbg, ubg)
    P, _ = ls.dense(0)
    f = lambda w: mdl.objective(w[None, :4], w[None, 4:])[0]
    w0 = np.concatenate([p[0], x[0]]); n = len(w0); h = 1e-5
    g = np.array([(f(w0 + h * np.eye(n)[i]) - f(w0 - h * np.eye(n)[i])) / (2 * h) for i in range(n)])
    assert np.abs(g - ls.q[0]).max() < 1e-6 * (1 + np.abs(g).max())
    H = np.zeros((n, n))
    for i in range(n):
        e = h * np.eye(n)[i]
        qp_ = mdl.local_system((w0 + e)[None, :4], (w0 + e)[None, 4:], lbx[:1], ubx[:1], lbg[:1], ubg[:1]).q[0]
        qm_ = mdl.local_system((w0 - e)[None, :4], (w0 - e)[None, 4:], lbx[:1], ubx[:1], lbg[:1], ubg[:1]).q[0]
        H[:, i] = (qp_ - qm_) / (2 * h)
    assert np.abs(H - P).max() < 1e-5 * (1 + np.abs(P).max())
    assert P[4 + (N - 1) * 5, 4 + (N - 1) * 5] == 2.0 * Qk[-1, 0]
