"""GPU: local-system evaluation on device (mpcqp_stage_*) against the host NumPy statement of
the same formulas (models.StageOCP.local_system, which mirrors reference src/sqp_solver/SQPOptimizationSolver.cpp:47-120),
and the device-resident SQP loop against the host loop.

Tolerances: P (constants) and the identity entries of A bit-exact; everything that passes through sin/cos
(Jacobian blocks, dynamics residuals) to 1e-12 relative -- forward-mode duals on the device vs complex-step on the host,
same operation order, different libm."""
import numpy as np
import pytest

from optimal_control_problem_amd import models

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

CASES = [("double_integrator", 20, 64), ("quadrotor", 20, 96), ("quadrotor", 2, 5), ("cartpole", 30, 64), ("cartpole", 100, 17)]


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")


def _close(a, b, tol):
    """elementwise |a - b| <= tol * max(1, |b|), with infinities required to match exactly"""
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin) and np.array_equal(a[~fin], b[~fin])
    return (np.abs(a[fin] - b[fin]) <= tol * np.maximum(1.0, np.abs(b[fin]))).all()


@pytest.mark.parametrize("name,N,B", CASES)
def test_eval_matches_host(built, name, N, B):
    from optimal_control_problem_amd.stage_eval import StageEvaluator
    mdl, ls, meta = models.make_workload(name, B, N=N)
    ev = StageEvaluator(mdl)
    assert (ev.n, ev.m, ev.nnzP, ev.nnzA) == (ls.n, ls.m, len(ls.Pi), len(ls.Ai))
    assert (ev.Pp == ls.Pp).all() and (ev.Pi == ls.Pi).all() and (ev.Ap == ls.Ap).all() and (ev.Ai == ls.Ai).all()
    rng = np.random.default_rng(3)
    p = meta["p"] + rng.normal(0.0, 0.2, meta["p"].shape)          # nonzero reference state exercises the p-coupling
    ref = mdl.local_system(p, meta["x_iterate"], meta["lbx"], meta["ubx"], meta["lbg"], meta["ubg"])
    out = ev.eval(_dev(p), _dev(meta["x_iterate"]), _dev(meta["lbx"]), _dev(meta["ubx"]), _dev(meta["lbg"]), _dev(meta["ubg"]))
    got = {k: v.cpu().numpy() for k, v in out.items()}
    assert np.array_equal(got["P"], ref.P)
    assert _close(got["A"], ref.A, 1e-12)
    ident = np.isin(ref.A[0], [1.0])                                # structural ones are exact
    assert np.array_equal(got["A"][:, ident], ref.A[:, ident])
    assert _close(got["q"], ref.q, 1e-12)
    assert _close(got["l"], ref.l, 1e-12) and _close(got["u"], ref.u, 1e-12)
    # merit + step
    f, g = ev.merit(_dev(p), _dev(meta["x_iterate"]))
    assert _close(f.cpu().numpy(), mdl.objective(p, meta["x_iterate"]), 1e-12)
    assert _close(g.cpu().numpy(), np.abs(mdl.constraints(meta["x_iterate"])).max(axis=1), 1e-11)
    dw = rng.normal(size=(B, ls.n)); x = _dev(meta["x_iterate"])
    sm = ev.step(0.5, _dev(dw), x)
    assert np.array_equal(x.cpu().numpy(), meta["x_iterate"] + 0.5 * dw[:, mdl.np:])
    assert np.array_equal(sm.cpu().numpy(), np.abs(0.5 * dw[:, mdl.np:]).max(axis=1))
    # with a status array, instances whose QP returned no point keep their iterate
    status = torch.ones(B, dtype=torch.int32, device="cuda"); status[0] = 3; status[B - 1] = 9; status[1] = 2; status[2] = 7
    before = x.clone(); sm = ev.step(1.0, _dev(dw), x, status=status)
    moved = (x != before).any(dim=1).cpu().numpy()
    assert not moved[0] and not moved[B - 1] and moved[1:B - 1].all() and float(sm[0]) == 0.0
    ev.close()


def test_eval_feeds_the_qp_without_leaving_the_device(built):
    """device-evaluated QP data -> mpcqp_update(MPCQP_MEM_DEVICE) -> solve: same statuses / iteration counts / solution as
    the host-evaluated data through the host path"""
    from optimal_control_problem_amd.batch_qp import BatchQP, solve_local_system
    from optimal_control_problem_amd.stage_eval import StageEvaluator
    B = 256
    mdl, ls, meta = models.make_workload("quadrotor", B)
    ev = StageEvaluator(mdl)
    out = ev.eval(*[_dev(meta[k]) for k in ("p", "x_iterate", "lbx", "ubx", "lbg", "ubg")])
    qp = BatchQP(ev.n, ev.m, B, ev.Pp, ev.Pi, ev.Ap, ev.Ai)
    qp.update(out["P"], out["q"], out["A"], out["l"], out["u"]); qp.solve(); got = qp.get(); qp.close()
    ref = solve_local_system(ls)
    assert (got["status"] == ref["status"]).all() and (got["status"] == 1).all()
    assert (got["iters"] == ref["iters"]).mean() >= 0.99            # rounding-level input differences may move a check boundary
    same = got["iters"] == ref["iters"]
    assert np.abs(got["x"][same] - ref["x"][same]).max() < 1e-7
    ev.close()


@pytest.mark.parametrize("name,N,alpha,warm", [("cartpole", 30, 0.5, False), ("quadrotor", 10, 0.5, False), ("cartpole", 30, 0.5, True),
                                               ("double_integrator", 20, 1.0, False)])
def test_device_sqp_equals_host_sqp(built, name, N, alpha, warm):
    from optimal_control_problem_amd.sqp import DeviceSQPOptimizationSolver, SQPOptimizationSolver
    B = 16
    mdl, ls, meta = models.make_workload(name, B, N=N)
    arg = dict(lbx=meta["lbx"], ubx=meta["ubx"], lbg=meta["lbg"], ubg=meta["ubg"], p=meta["p"])
    opt = {"max_iter": 6, "alpha": alpha, "warm_start_admm": warm}
    host = SQPOptimizationSolver(mdl, opt, batch=B)
    dev = DeviceSQPOptimizationSolver(mdl, opt, batch=B)
    rh = host.getOptimalSolution(arg); rd = dev.getOptimalSolution(arg)
    scale = 1.0 + np.abs(rh["x"]).max()
    assert np.abs(rd["x"] - rh["x"]).max() <= 1e-6 * scale
    assert _close(rd["f"], rh["f"], 1e-6)
    ih = np.stack(host.admm_iterations); idv = np.stack([t.cpu().numpy() for t in dev.admm_iterations])
    assert (ih == idv).mean() >= 0.95
    # second call continues from the stored iterate, like the reference's result_ member
    rh2 = host.getOptimalSolution(arg); rd2 = dev.getOptimalSolution(arg)
    assert np.abs(rd2["x"] - rh2["x"]).max() <= 1e-6 * scale
    assert float(dev.gmax.max()) <= np.abs(mdl.constraints(rd2["x"])).max() * (1 + 1e-9) + 1e-15
    host.qpSolver_.close(); dev.close()


def test_config4_cartpole_n100_warm_started_sqp(built):
    """BASELINE.json configs[3] as stated: cart-pole swing-up, N = 100, fixed-step SQP (alpha = 0.5), ADMM of each QP warm-started
    from the previous SQP iteration.  The device-resident loop against the host loop over the CPU oracle (same bar as
    test_device_sqp_equals_host_sqp; 8 SQP iterations, 16 instances keep the oracle cheap)."""
    from optimal_control_problem_amd.sqp import DeviceSQPOptimizationSolver, SQPOptimizationSolver
    from tests.support.oracle_backend import OracleCuCaQP
    B = 16
    mdl, ls, meta = models.make_workload("cartpole", B, N=100)
    arg = dict(lbx=meta["lbx"], ubx=meta["ubx"], lbg=meta["lbg"], ubg=meta["ubg"], p=meta["p"])
    opt = {"max_iter": 8, "alpha": 0.5, "warm_start_admm": True}
    host = SQPOptimizationSolver(mdl, opt, batch=B, qp_solver=OracleCuCaQP(batch=B, nthreads=8))
    dev = DeviceSQPOptimizationSolver(mdl, opt, batch=B)
    host.setInitialGuess(meta["x_iterate"]); dev.setInitialGuess(meta["x_iterate"])   # hanging pole, theta ~ pi + N(0, 0.05^2)
    rh = host.getOptimalSolution(arg); rd = dev.getOptimalSolution(arg)
    scale = 1.0 + np.abs(rh["x"]).max()
    assert np.isfinite(rd["x"]).all()
    assert np.abs(rd["x"] - rh["x"]).max() <= 1e-6 * scale
    assert _close(rd["f"], rh["f"], 1e-6)
    ih = np.stack(host.admm_iterations); idv = np.stack([t.cpu().numpy() for t in dev.admm_iterations])
    assert ih.shape == (8, B) and (ih == idv).mean() >= 0.95
    assert idv[-1].mean() < idv[0].mean()                      # the warm start pays: late QPs need fewer ADMM iterations than the first
    dev.close()


def test_sqp_tol_stops_both_loops_at_the_same_iteration(built):
    """opt-in convergence stop: every instance's step below sqp_tol ends the loop -- without `verbose`,
    unlike the reference's only stop (SQPOptimizationSolver.cpp:183-197).  Device loop = host loop, fewer iterations than step_num;
    with the option absent both run the fixed count."""
    from optimal_control_problem_amd.sqp import DeviceSQPOptimizationSolver, SQPOptimizationSolver
    B = 16
    mdl, ls, meta = models.make_workload("double_integrator", B)
    arg = dict(lbx=meta["lbx"], ubx=meta["ubx"], lbg=meta["lbg"], ubg=meta["ubg"], p=meta["p"])
    opt = {"max_iter": 12, "alpha": 1.0, "sqp_tol": 1e-2}
    host = SQPOptimizationSolver(mdl, opt, batch=B); dev = DeviceSQPOptimizationSolver(mdl, opt, batch=B)
    rh = host.getOptimalSolution(arg); rd = dev.getOptimalSolution(arg)
    assert host.iterations_done == dev.iterations_done and 1 < dev.iterations_done < 12
    assert float(dev.step_max.max()) < 1e-2 and host.step_max.max() < 1e-2
    assert np.abs(rd["x"] - rh["x"]).max() <= 1e-6 * (1.0 + np.abs(rh["x"]).max())
    host.qpSolver_.close(); dev.close()
    plain = DeviceSQPOptimizationSolver(mdl, {"max_iter": 5, "alpha": 1.0}, batch=B)
    plain.getOptimalSolution(arg)
    assert plain.iterations_done == 5
    plain.close()


def test_stage_errors(built):
    from optimal_control_problem_amd import _lib
    from optimal_control_problem_amd.stage_eval import StageDesc, StageEvaluator, _bind
    import ctypes as C
    L = _bind(_lib.lib())
    d = StageDesc()
    assert L.mpcqp_stage_default(7, 10, C.byref(d)) == _lib.ERR_ARG
    assert L.mpcqp_stage_default(1, 1, C.byref(d)) == _lib.OK
    h = C.c_void_p()
    assert L.mpcqp_stage_create(C.byref(d), C.byref(h)) == _lib.ERR_ARG and not h.value      # horizon < 2
    ev = StageEvaluator(name="cartpole", horizon=5)
    B = 3
    good = [torch.zeros((B, w), dtype=torch.float64, device="cuda") for w in (ev.np, ev.nvar, ev.nvar, ev.nvar, ev.ng, ev.ng)]
    ev.eval(*good)
    with pytest.raises(ValueError, match="dimension mismatch"):
        ev.eval(good[0], good[1][:, :-1].contiguous(), *good[2:])
    with pytest.raises(ValueError, match="CUDA"):
        ev.eval(good[0].cpu(), *good[1:])
    assert L.mpcqp_stage_eval(ev._h, 0, *([good[0].data_ptr()] * 11), None) == _lib.ERR_ARG
    assert L.mpcqp_stage_eval(ev._h, B, None, *([good[0].data_ptr()] * 10), None) == _lib.ERR_ARG
    ev.close()


# ------------------------------------------------------------------------------------- generated (user) dynamics
@pytest.mark.parametrize("name,N,B", [("quadrotor", 10, 40), ("cartpole", 30, 33)])
def test_generated_dynamics_equal_builtin(built, name, N, B):
    """the traced + generated + hipcc-compiled functor of a zoo model against the library's hand-written one"""
    from optimal_control_problem_amd.stage_eval import StageEvaluator
    mdl, ls, meta = models.make_workload(name, B, N=N)
    a = StageEvaluator(mdl); g = StageEvaluator(mdl, codegen=True)
    assert g.library is not None and a.library is None and (g.Ai == a.Ai).all()
    args = [_dev(meta[k]) for k in ("p", "x_iterate", "lbx", "ubx", "lbg", "ubg")]
    oa = a.eval(*args); og = g.eval(*args)
    for k in ("P", "q", "A", "l", "u"):
        assert _close(og[k].cpu().numpy(), oa[k].cpu().numpy(), 1e-12), k
    fa, ga = a.merit(args[0], args[1]); fg, gg = g.merit(args[0], args[1])
    assert _close(fg.cpu().numpy(), fa.cpu().numpy(), 1e-12) and _close(gg.cpu().numpy(), ga.cpu().numpy(), 1e-10)
    a.close(); g.close()


def test_facade_with_custom_dynamics_runs_device_resident(built):
    """OptimalControlProblem subclass with its own dynamics (not in the zoo): gen_code: true compiles them for the GPU and
    the whole SQP tick stays on the device; same trajectory as the host path (NumPy local system + GPU QP)"""
    import yaml
    from optimal_control_problem_amd.ocp import Dynamics, OptimalControlProblem
    from tests.test_codegen import pendulum_on_cart_with_drag
    text = """
      discretization_settings: {dt: 0.01, horizon: 12}
      solver_settings: {verbose: false, gen_code: %s, load_lib: false, max_iter: 1000, warm_start: true, solve_method: CUDA_SQP,
                        SQP_settings: {alpha: 0.7, step_num: 5}}
      OCP_variables:
        - {name: state, size: 3, lower_bound: [-.inf, -.inf, -3.0], upper_bound: [.inf, .inf, 3.0]}
        - {name: input, size: 1, lower_bound: [-5.0], upper_bound: [5.0]}
    """

    class Pendulum(OptimalControlProblem):
        def deployConstraintsAndAddCost(self):
            cfg = self.OCPConfigPtr_
            ref = self.setReference(3)
            for k in range(cfg.getHorizon()):
                self.addVectorCost([5.0, 0.5, 0.2], cfg.getVariable(k, "state") - ref)
                self.addVectorCost([0.05], cfg.getVariable(k, "input"))
            for k in range(cfg.getHorizon() - 1):
                self.addEquationConstraint("dynamics", cfg.getVariable(k + 1, "state"),
                                           Dynamics(pendulum_on_cart_with_drag, cfg.getVariable(k, "state"), cfg.getVariable(k, "input")))

    B = 12
    rng = np.random.default_rng(8)
    frame = np.concatenate([rng.normal(0, 0.4, (B, 3)), np.zeros((B, 1))], axis=1); ref = np.zeros((B, 3))
    res = {}
    for flag in ("false", "true"):
        ocp = Pendulum(yaml.safe_load(text % flag), batch=B)
        ocp.deployConstraintsAndAddCost(); ocp.genSolver()
        assert ocp.deviceResident == (flag == "true")
        res[flag] = ocp.computeOptimalTrajectory(frame, ref)
        assert type(ocp.OSQPSolverPtr_).__name__ == ("DeviceSQPOptimizationSolver" if flag == "true" else "SQPOptimizationSolver")
    assert np.abs(res["true"] - res["false"]).max() <= 1e-6 * (1 + np.abs(res["false"]).max())
    assert np.abs(res["true"][:, :4] - frame).max() < 5e-3


def test_device_sqp_constant_matrices_matches_oracle_kept_workspace(built):
    """linear MPC ticks on the kept workspace (mpcqp_update_vectors): the device loop with constant_matrices against the same
    sequence of QPs on the oracle's kept workspaces"""
    from optimal_control_problem_amd.sqp import DeviceSQPOptimizationSolver
    from oracle import oracle as orc
    B = 12
    mdl, ls, meta = models.make_workload("double_integrator", B)
    dev = DeviceSQPOptimizationSolver(mdl, {"max_iter": 1, "alpha": 1.0, "constant_matrices": True}, batch=B)
    pat = orc.Pattern(ls.n, ls.m, ls.Pp, ls.Pi, ls.Ap, ls.Ai); st = orc.State(pat, B, orc.default_settings())
    arg = dict(lbx=meta["lbx"].copy(), ubx=meta["ubx"].copy(), lbg=meta["lbg"], ubg=meta["ubg"], p=meta["p"])
    x = np.zeros((B, mdl.nvar)); rng = np.random.default_rng(2)
    for tick in range(4):
        sys_ = mdl.local_system(meta["p"], x, arg["lbx"], arg["ubx"], arg["lbg"], arg["ubg"])
        ref = st.solve(sys_.P, sys_.q, sys_.A, sys_.l, sys_.u) if tick == 0 else st.solve_vectors(sys_.q, sys_.l, sys_.u)
        got = dev.getOptimalSolution(arg)
        x = x + ref["x"][:, mdl.np:]
        assert (dev.iters.cpu().numpy() == ref["iters"]).all()
        assert np.abs(got["x"] - x).max() <= 1e-6 * (1 + np.abs(x).max())
        s0 = arg["lbx"][:, :mdl.nx] + rng.normal(0, 0.05, (B, mdl.nx))
        arg["lbx"][:, :mdl.nx] = s0; arg["ubx"][:, :mdl.nx] = s0
    dev.close()


def test_path_constraints_on_device(built):
    """per-stage path constraint lo <= h(s_k, u_k) <= hi through the generated evaluator: QP data equal to the host formulation,
    the QP honours the linearised constraint, and the device SQP loop equals the host loop"""
    from optimal_control_problem_amd.batch_qp import BatchQP
    from optimal_control_problem_amd.sqp import DeviceSQPOptimizationSolver, SQPOptimizationSolver
    from optimal_control_problem_amd.stage_eval import StageEvaluator
    from tests.test_codegen import CartPoleWall
    B, N = 20, 12
    mdl = CartPoleWall(N, 0.02)
    rng = np.random.default_rng(3)
    x = rng.normal(0, 0.3, (B, mdl.nvar)); p = np.zeros((B, 4))
    frame0 = x[:, :mdl.f].copy(); frame0[:, 0] = rng.uniform(-0.5, 0.5, B)
    lbx, ubx, lbg, ubg = mdl.stacked_bounds(frame0)
    ref = mdl.local_system(p, x, lbx, ubx, lbg, ubg)
    ev = StageEvaluator(mdl)
    assert ev.library is not None and (ev.m, ev.ng) == (mdl.m, mdl.ng) and (ev.Ai == mdl.Ai).all() and (ev.Ap == mdl.Ap).all()
    out = ev.eval(_dev(p), _dev(x), _dev(lbx), _dev(ubx), _dev(lbg), _dev(ubg))
    for k, r in (("P", ref.P), ("q", ref.q), ("A", ref.A), ("l", ref.l), ("u", ref.u)):
        assert _close(out[k].cpu().numpy(), r, 1e-12), k
    f, g = ev.merit(_dev(p), _dev(x))
    hv = mdl.path_values(x).reshape(B, N, 2)
    viol = np.maximum(np.maximum(np.asarray(mdl.h_lo) - hv, hv - np.asarray(mdl.h_hi)).max(axis=(1, 2)), np.abs(mdl.constraints(x)).max(axis=1))
    assert _close(g.cpu().numpy(), viol, 1e-11)
    qp = BatchQP(ev.n, ev.m, B, ev.Pp, ev.Pi, ev.Ap, ev.Ai)
    qp.update(out["P"], out["q"], out["A"], out["l"], out["u"]); qp.solve(); got = qp.get(); qp.close()
    assert (got["status"] == 1).all()
    z = got["z"][:, mdl.n + mdl.ngd:]                       # A dx on the path rows stays inside the shifted bounds (ADMM tolerance)
    assert (z <= ref.u[:, mdl.n + mdl.ngd:] + 1e-2).all()
    ev.close()
    arg = dict(lbx=lbx, ubx=ubx, lbg=lbg, ubg=ubg, p=p)
    host = SQPOptimizationSolver(mdl, {"max_iter": 5, "alpha": 0.6}, batch=B); dev = DeviceSQPOptimizationSolver(mdl, {"max_iter": 5, "alpha": 0.6}, batch=B)
    host.setInitialGuess(x); dev.setInitialGuess(x)
    rh = host.getOptimalSolution(arg); rd = dev.getOptimalSolution(arg)
    assert np.abs(rd["x"] - rh["x"]).max() <= 1e-6 * (1 + np.abs(rh["x"]).max())
    assert (mdl.path_values(rd["x"]).reshape(B, N, 2)[:, 1:, 0] <= 1.5 + 5e-2).all()
    host.qpSolver_.close(); dev.close()


def test_facade_path_constraint(built):
    """OptimalControlProblem.addInequalityConstraint with a per-frame Path(...) expression, device-resident (gen_code: true)"""
    import yaml
    from optimal_control_problem_amd.ocp import Dynamics, OptimalControlProblem, Path
    from tests.test_codegen import pendulum_on_cart_with_drag
    text = """
      discretization_settings: {dt: 0.01, horizon: 10}
      solver_settings: {verbose: false, gen_code: %s, load_lib: false, max_iter: 1000, warm_start: true, solve_method: CUDA_SQP,
                        SQP_settings: {alpha: 0.8, step_num: 4}}
      OCP_variables:
        - {name: state, size: 3, lower_bound: [-.inf, -.inf, -3.0], upper_bound: [.inf, .inf, 3.0]}
        - {name: input, size: 1, lower_bound: [-5.0], upper_bound: [5.0]}
    """
    def power(s, u):                                       # |velocity * force| bounded: a coupled state-input constraint
        return np.stack([s[..., 2] * u[..., 0]], axis=-1)

    class Pendulum(OptimalControlProblem):
        def deployConstraintsAndAddCost(self):
            cfg = self.OCPConfigPtr_
            ref = self.setReference(3)
            for k in range(cfg.getHorizon()):
                self.addVectorCost([5.0, 0.5, 0.2], cfg.getVariable(k, "state") - ref)
                self.addVectorCost([0.05], cfg.getVariable(k, "input"))
                # interleaved with the dynamics on purpose: the facade sorts the rows into [dynamics; path]
                self.addInequalityConstraint("power", [-0.4], Path(power, cfg.getVariable(k, "state"), cfg.getVariable(k, "input"), 1), [0.4])
                if k < cfg.getHorizon() - 1:
                    self.addEquationConstraint("dynamics", cfg.getVariable(k + 1, "state"),
                                               Dynamics(pendulum_on_cart_with_drag, cfg.getVariable(k, "state"), cfg.getVariable(k, "input")))

    B = 8
    rng = np.random.default_rng(9)
    frame = np.concatenate([rng.normal(0, 0.4, (B, 3)), np.zeros((B, 1))], axis=1); ref = np.zeros((B, 3))
    res = {}
    for flag in ("false", "true"):
        ocp = Pendulum(yaml.safe_load(text % flag), batch=B)
        ocp.deployConstraintsAndAddCost(); ocp.genSolver()
        assert (ocp.model_.nh, ocp.model_.ng) == (1, 9 * 3 + 10)
        res[flag] = ocp.computeOptimalTrajectory(frame, ref)
    assert np.abs(res["true"] - res["false"]).max() <= 1e-6 * (1 + np.abs(res["false"]).max())
    X = res["true"].reshape(B, 10, 4)
    assert (np.abs(X[:, 1:, 2] * X[:, 1:, 3]) <= 0.4 + 5e-2).all()


def test_link_constraints_on_device(built):
    """constraints that couple consecutive frames beyond the dynamics (rate limit u_{k+1} - u_k, slew limit of the pole tip):
    rows [p; x; g; h; r] through the generated evaluator equal the host formulation, the QP honours them, the device SQP loop equals
    the host loop, and the rate limit is active in the result"""
    from optimal_control_problem_amd.batch_qp import BatchQP
    from optimal_control_problem_amd.sqp import DeviceSQPOptimizationSolver, SQPOptimizationSolver
    from optimal_control_problem_amd.stage_eval import StageEvaluator
    from tests.test_codegen import CartPoleRate
    B, N = 20, 12
    mdl = CartPoleRate(N, 0.02)
    rng = np.random.default_rng(3)
    x = rng.normal(0, 0.3, (B, mdl.nvar)); p = np.zeros((B, 4))
    frame0 = x[:, :mdl.f].copy()
    lbx, ubx, lbg, ubg = mdl.stacked_bounds(frame0)
    ref = mdl.local_system(p, x, lbx, ubx, lbg, ubg)
    ev = StageEvaluator(mdl)
    assert ev.library is not None and (ev.m, ev.ng) == (mdl.m, mdl.ng) and (ev.Ai == mdl.Ai).all() and (ev.Ap == mdl.Ap).all()
    out = ev.eval(_dev(p), _dev(x), _dev(lbx), _dev(ubx), _dev(lbg), _dev(ubg))
    for k, r in (("P", ref.P), ("q", ref.q), ("A", ref.A), ("l", ref.l), ("u", ref.u)):
        assert _close(out[k].cpu().numpy(), r, 1e-12), k
    f, g = ev.merit(_dev(p), _dev(x))
    kv = mdl.link_values(x).reshape(B, N - 1, 2)
    viol = np.maximum(np.maximum(np.asarray(mdl.k_lo) - kv, kv - np.asarray(mdl.k_hi)).max(axis=(1, 2)), np.abs(mdl.constraints(x)).max(axis=1))
    assert _close(g.cpu().numpy(), viol, 1e-11)
    qp = BatchQP(ev.n, ev.m, B, ev.Pp, ev.Pi, ev.Ap, ev.Ai)
    qp.update(out["P"], out["q"], out["A"], out["l"], out["u"]); qp.solve(); got = qp.get(); qp.close()
    assert (got["status"] == 1).all()
    z = got["z"][:, mdl.n + mdl.ngd:]                       # A dx on the link rows stays inside the shifted bounds (ADMM tolerance)
    assert (z <= ref.u[:, mdl.n + mdl.ngd:] + 1e-2).all() and (z >= ref.l[:, mdl.n + mdl.ngd:] - 1e-2).all()
    ev.close()
    arg = dict(lbx=lbx, ubx=ubx, lbg=lbg, ubg=ubg, p=p)
    opt = {"max_iter": 6, "alpha": 0.7}
    host = SQPOptimizationSolver(mdl, opt, batch=B); dev = DeviceSQPOptimizationSolver(mdl, opt, batch=B)
    host.setInitialGuess(x); dev.setInitialGuess(x)
    rh = host.getOptimalSolution(arg); rd = dev.getOptimalSolution(arg)
    assert np.abs(rd["x"] - rh["x"]).max() <= 1e-6 * (1 + np.abs(rh["x"]).max())
    du = np.diff(rd["x"].reshape(B, N, mdl.f)[:, :, 4], axis=1)
    assert (np.abs(du) <= 4.0 + 0.3).all()
    free = models.CartPole(N, 0.02)                          # the same problem without the limits moves the force faster: the limit binds
    hf = SQPOptimizationSolver(free, opt, batch=B); hf.setInitialGuess(x)
    fl = free.stacked_bounds(frame0)
    rf = hf.getOptimalSolution(dict(lbx=fl[0], ubx=fl[1], lbg=fl[2], ubg=fl[3], p=p))
    assert np.abs(np.diff(rf["x"].reshape(B, N, mdl.f)[:, :, 4], axis=1)).max() > np.abs(du).max() + 0.5
    host.qpSolver_.close(); hf.qpSolver_.close(); dev.close()


def test_facade_rate_limit(built):
    """OptimalControlProblem.addInequalityConstraint with a Link(...) expression on every stage (a rate limit on the input):
    host loop = device-resident loop (gen_code), and the limit holds"""
    import yaml
    from optimal_control_problem_amd.ocp import Dynamics, Link, OptimalControlProblem
    text = """
      discretization_settings: {dt: 0.05, horizon: 12}
      solver_settings: {verbose: false, gen_code: %s, load_lib: false, max_iter: 1000, warm_start: true, solve_method: CUDA_SQP,
                        SQP_settings: {alpha: 1.0, step_num: 3}}
      OCP_variables:
        - {name: state, size: 2, lower_bound: [-.inf, -2.0], upper_bound: [.inf, 2.0]}
        - {name: input, size: 1, lower_bound: [-1.0], upper_bound: [1.0]}
    """
    def step(s, u):
        h = 0.05
        return np.stack([s[..., 0] + h * s[..., 1] + 0.5 * h * h * u[..., 0], s[..., 1] + h * u[..., 0]], axis=-1)

    def rate(s, u, sn, un):
        return np.stack([un[..., 0] - u[..., 0]], axis=-1)

    class DI(OptimalControlProblem):
        def deployConstraintsAndAddCost(self):
            cfg = self.OCPConfigPtr_
            ref = self.setReference(2)
            for k in range(cfg.getHorizon()):
                self.addVectorCost([10.0, 1.0], cfg.getVariable(k, "state") - ref)
                self.addVectorCost([0.1], cfg.getVariable(k, "input"))
                if k < cfg.getHorizon() - 1:
                    self.addEquationConstraint("dynamics", cfg.getVariable(k + 1, "state"), Dynamics(step, cfg.getVariable(k, "state"), cfg.getVariable(k, "input")))
                    self.addInequalityConstraint("rate", [-0.15], Link(rate, cfg.getVariable(k, "state"), cfg.getVariable(k, "input"),
                                                                      cfg.getVariable(k + 1, "state"), cfg.getVariable(k + 1, "input"), 1), [0.15])

    B = 6
    rng = np.random.default_rng(4)
    frame = np.concatenate([rng.uniform(-2, 2, (B, 1)), rng.uniform(-1, 1, (B, 1)), np.zeros((B, 1))], axis=1); ref = np.zeros((B, 2))
    res = {}
    for flag in ("false", "true"):
        ocp = DI(yaml.safe_load(text % flag), batch=B)
        ocp.deployConstraintsAndAddCost(); ocp.genSolver()
        assert (ocp.model_.nk, ocp.model_.ng) == (1, 11 * 2 + 11)
        res[flag] = ocp.computeOptimalTrajectory(frame, ref)
    assert np.abs(res["true"] - res["false"]).max() <= 1e-6 * (1 + np.abs(res["false"]).max())
    U = res["true"].reshape(B, 12, 3)[:, :, 2]
    assert (np.abs(np.diff(U, axis=1)) <= 0.15 + 2e-2).all() and np.abs(np.diff(U, axis=1)).max() > 0.1      # the limit holds and binds


def test_per_frame_weights_on_device(built):
    """mpcqp_stage_set_weights (terminal cost / weight ramps): device evaluation and merit equal the host formulation, for a
    built-in functor and for generated dynamics; the facade compiles per-step addVectorCost weights into them"""
    from optimal_control_problem_amd.stage_eval import StageEvaluator
    N, B = 12, 24
    rng = np.random.default_rng(1)
    Qk = rng.uniform(0.1, 5.0, (N, 4)); Rk = rng.uniform(0.01, 1.0, (N, 1)); Qk[-1] *= 20.0

    class CP(models.CartPole):
        def __init__(self): models.StageOCP.__init__(self, N, 0.02, Qk, Rk)

    mdl = CP()
    x = rng.normal(0, 0.4, (B, mdl.nvar)); p = rng.normal(0, 0.3, (B, 4))
    lbx, ubx, lbg, ubg = mdl.stacked_bounds(x[:, :mdl.f].copy())
    ref = mdl.local_system(p, x, lbx, ubx, lbg, ubg)
    for cg in (False, True):
        ev = StageEvaluator(mdl, codegen=cg)
        out = ev.eval(_dev(p), _dev(x), _dev(lbx), _dev(ubx), _dev(lbg), _dev(ubg))
        for k, r in (("P", ref.P), ("q", ref.q), ("A", ref.A), ("l", ref.l), ("u", ref.u)):
            assert _close(out[k].cpu().numpy(), r, 1e-12), (cg, k)
        f, _ = ev.merit(_dev(p), _dev(x))
        assert _close(f.cpu().numpy(), mdl.objective(p, x), 1e-12)
        ev.close()
    # facade: a terminal weight through per-step addVectorCost, device-resident = host
    import yaml
    from optimal_control_problem_amd.ocp import Dynamics, OptimalControlProblem
    text = """
      discretization_settings: {dt: 0.05, horizon: 10}
      solver_settings: {verbose: false, gen_code: %s, load_lib: false, max_iter: 1000, warm_start: true, solve_method: CUDA_SQP,
                        SQP_settings: {alpha: 1.0, step_num: 2}}
      OCP_variables:
        - {name: state, size: 2, lower_bound: [-.inf, -2.0], upper_bound: [.inf, 2.0]}
        - {name: input, size: 1, lower_bound: [-1.0], upper_bound: [1.0]}
    """
    F = lambda s, u: np.stack([s[..., 0] + 0.05 * s[..., 1] + 0.00125 * u[..., 0], s[..., 1] + 0.05 * u[..., 0]], axis=-1)

    class DI(OptimalControlProblem):
        def deployConstraintsAndAddCost(self):
            cfg = self.OCPConfigPtr_; ref_ = self.setReference(2); Nh = cfg.getHorizon()
            for k in range(Nh):
                self.addVectorCost([10.0, 1.0] if k < Nh - 1 else [400.0, 40.0], cfg.getVariable(k, "state") - ref_)
                self.addVectorCost([0.1], cfg.getVariable(k, "input"))
            for k in range(Nh - 1):
                self.addEquationConstraint("dynamics", cfg.getVariable(k + 1, "state"), Dynamics(F, cfg.getVariable(k, "state"), cfg.getVariable(k, "input")))

    frame = np.concatenate([rng.uniform(-1, 1, (8, 2)), np.zeros((8, 1))], axis=1); refv = np.zeros((8, 2)); res = {}
    for flag in ("false", "true"):
        ocp = DI(yaml.safe_load(text % flag), batch=8); ocp.deployConstraintsAndAddCost(); ocp.genSolver()
        assert ocp.model_.varying_weights and ocp.model_.Qk[-1, 0] == 400.0
        res[flag] = ocp.computeOptimalTrajectory(frame, refv)
    assert np.abs(res["true"] - res["false"]).max() <= 1e-6 * (1 + np.abs(res["false"]).max())


def test_terminal_constraint_as_path_with_per_frame_bounds(built):
    """a path constraint whose bounds differ by frame: loose everywhere but on the last frame = a terminal constraint (the cart ends
    inside a box, at rest).  mpcqp_stage_set_path_bounds feeds the violation measure; device SQP = host SQP; facade = the same."""
    import yaml
    from optimal_control_problem_amd.ocp import Dynamics, OptimalControlProblem, Path
    from optimal_control_problem_amd.sqp import DeviceSQPOptimizationSolver, SQPOptimizationSolver
    from optimal_control_problem_amd.stage_eval import StageEvaluator
    N, B = 10, 12
    lo = np.full((N, 2), -np.inf); hi = np.full((N, 2), np.inf)
    lo[-1] = [-0.05, -0.1]; hi[-1] = [0.05, 0.1]

    def terminal_box(s, u):
        return np.stack([s[..., 0], s[..., 2]], axis=-1)      # cart position and velocity

    class CP(models.CartPole):
        name = "cartpole_terminal_box"; nh = 2; h_lo = lo; h_hi = hi
        hfun = staticmethod(terminal_box)

    mdl = CP(N, 0.05)
    rng = np.random.default_rng(12)
    x = rng.normal(0, 0.1, (B, mdl.nvar)); p = np.zeros((B, 4))
    frame0 = x[:, :mdl.f].copy(); frame0[:, 0] = rng.uniform(-0.3, 0.3, B); frame0[:, 1] = rng.normal(0, 0.05, B)
    lbx, ubx, lbg, ubg = mdl.stacked_bounds(frame0)
    assert np.isinf(lbg[:, mdl.ngd:-2]).all() and np.array_equal(ubg[0, -2:], [0.05, 0.1])
    ev = StageEvaluator(mdl)
    f, g = ev.merit(_dev(p), _dev(x))
    hv = mdl.path_values(x).reshape(B, N, 2)
    viol = np.maximum(np.maximum(lo - hv, hv - hi).max(axis=(1, 2)), np.abs(mdl.constraints(x)).max(axis=1))
    assert _close(g.cpu().numpy(), viol, 1e-11)
    ev.close()
    arg = dict(lbx=lbx, ubx=ubx, lbg=lbg, ubg=ubg, p=p)
    opts = {"max_iter": 6, "alpha": 0.8}
    host = SQPOptimizationSolver(mdl, opts, batch=B); dev = DeviceSQPOptimizationSolver(mdl, opts, batch=B)
    host.setInitialGuess(x); dev.setInitialGuess(x)
    rh = host.getOptimalSolution(arg); rd = dev.getOptimalSolution(arg)
    assert np.abs(rd["x"] - rh["x"]).max() <= 1e-6 * (1 + np.abs(rh["x"]).max())
    last = rd["x"].reshape(B, N, mdl.f)[:, -1]
    assert (np.abs(last[:, 0]) <= 0.05 + 2e-2).all() and (np.abs(last[:, 2]) <= 0.1 + 2e-2).all()
    assert float(dev.gmax.max()) <= 5e-2
    host.qpSolver_.close(); dev.close()
    # the facade: the same Path on every frame, bounds loose but on the last
    text = """
      discretization_settings: {dt: 0.05, horizon: 10}
      solver_settings: {verbose: false, gen_code: %s, load_lib: false, max_iter: 1000, warm_start: true, solve_method: CUDA_SQP,
                        SQP_settings: {alpha: 0.8, step_num: 6}}
      OCP_variables:
        - {name: state, size: 4, lower_bound: [-2.4, -.inf, -.inf, -.inf], upper_bound: [2.4, .inf, .inf, .inf]}
        - {name: input, size: 1, lower_bound: [-20.0], upper_bound: [20.0]}
    """
    cp = models.CartPole(N, 0.05); Fd = cp.F

    class Terminal(OptimalControlProblem):
        def deployConstraintsAndAddCost(self):
            cfg = self.OCPConfigPtr_; ref = self.setReference(4); Nh = cfg.getHorizon()
            for k in range(Nh):
                st, inp = cfg.getVariable(k, "state"), cfg.getVariable(k, "input")
                self.addVectorCost([1.0, 10.0, 0.1, 0.1], st - ref); self.addVectorCost([0.01], inp)
                self.addInequalityConstraint("terminal_box", lo[k], Path(terminal_box, st, inp, 2), hi[k])
                if k < Nh - 1:
                    self.addEquationConstraint("dynamics", cfg.getVariable(k + 1, "state"), Dynamics(Fd, st, inp))

    frame = frame0[:, :5].copy(); frame[:, 4] = 0.0
    res = {}
    for flag in ("false", "true"):
        ocp = Terminal(yaml.safe_load(text % flag), batch=B)
        ocp.deployConstraintsAndAddCost(); ocp.genSolver()
        assert np.shape(ocp.model_.h_lo) == (N, 2)
        res[flag] = ocp.computeOptimalTrajectory(frame, np.zeros((B, 4)))
    assert np.abs(res["true"] - res["false"]).max() <= 1e-6 * (1 + np.abs(res["false"]).max())
