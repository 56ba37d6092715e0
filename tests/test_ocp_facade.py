"""Host-API facade (OCPConfig / OptimalControlProblem mirror): YAML semantics, builders, first-frame pinning,
CUDA_SQP dispatch.  CPU tests use the oracle-backed QP stand-in; the gpu-marked test runs the real engine."""
import numpy as np
import pytest
import yaml

from optimal_control_problem_amd import models
from optimal_control_problem_amd.ocp import Dynamics, OCPConfig, OptimalControlProblem

# the README example (reference readme.md:43-62: dt 0.005, horizon 20, solve_method CUDA_SQP) completed with the keys the
# code actually reads (max_iter, warm_start, SQP_settings.{alpha, step_num}, gen_code, load_lib)
YAML_TEXT = """
optimal_control_problem:
  discretization_settings:
    dt: 0.005
    horizon: 20
  solver_settings:
    verbose: false
    gen_code: false
    load_lib: false
    max_iter: 1000
    warm_start: true
    solve_method: CUDA_SQP
    SQP_settings:
      alpha: 1.0
      step_num: 2
  OCP_variables:
    - name: "state"
      size: 2
      lower_bound: [-.inf, -2.0]
      upper_bound: [.inf, 2.0]
    - name: "input"
      size: 1
      lower_bound: ["-1.0"]
      upper_bound: [1.0]
"""


def _node():
    return yaml.safe_load(YAML_TEXT)["optimal_control_problem"]      # the ctor takes the inner node


class DoubleIntegratorOCP(OptimalControlProblem):
    def deployConstraintsAndAddCost(self):
        cfg = self.OCPConfigPtr_
        h = cfg.getDt()

        def F(s, u):
            return np.stack([s[..., 0] + h * s[..., 1] + 0.5 * h * h * u[..., 0], s[..., 1] + h * u[..., 0]], axis=-1)

        ref = self.setReference(2)
        for k in range(cfg.getHorizon()):
            self.addVectorCost([10.0, 1.0], cfg.getVariable(k, "state") - ref)
            self.addVectorCost([0.1], cfg.getVariable(k, "input"))
        for k in range(cfg.getHorizon() - 1):
            self.addEquationConstraint("dynamics", cfg.getVariable(k + 1, "state"),
                                       Dynamics(F, cfg.getVariable(k, "state"), cfg.getVariable(k, "input")))


def test_ocpconfig_yaml_semantics():
    cfg = OCPConfig(_node())
    assert (cfg.getHorizon(), cfg.getDt(), cfg.getFrameSize(), cfg.getVariables()) == (20, 0.005, 3, 60)
    lo, hi = cfg.getLowerBounds(), cfg.getUpperBounds()
    assert len(lo) == 20 and np.array_equal(lo[7], [-np.inf, -2.0, -1.0]) and np.array_equal(hi[19], [np.inf, 2.0, 1.0])
    v = cfg.getVariable(3, "input")                                   # X[k * frameSize + offset : + size]
    assert (v.start, v.stop) == (3 * 3 + 2, 3 * 3 + 3)
    with pytest.raises(IndexError):
        cfg.getVariable(20, "state")
    with pytest.raises(ValueError):
        cfg.getVariable(0, "nope")
    bad = _node(); del bad["OCP_variables"][0]["lower_bound"]
    with pytest.raises(ValueError, match="Missing lower_bound"):
        OCPConfig(bad)
    bad = _node(); bad["OCP_variables"][1]["size"] = 0
    with pytest.raises(ValueError, match="must be positive"):
        OCPConfig(bad)


def test_validate_config_and_solver_type():
    bad = _node(); del bad["solver_settings"]["SQP_settings"]["alpha"]
    with pytest.raises(RuntimeError, match="Invalid configuration"):
        DoubleIntegratorOCP(bad)
    bad = _node(); bad["solver_settings"]["solve_method"] = "NOPE"
    with pytest.raises(ValueError, match="Unknown solver type"):
        DoubleIntegratorOCP(bad)
    ipopt = _node(); ipopt["solver_settings"]["solve_method"] = "IPOPT"
    ocp = DoubleIntegratorOCP(ipopt); ocp.deployConstraintsAndAddCost()
    with pytest.raises(NotImplementedError):
        ocp.genSolver()


def _tick(ocp, B):
    rng = np.random.default_rng(5)
    frame = np.concatenate([rng.uniform([-1, -1], [1, 1], size=(B, 2)), np.zeros((B, 1))], axis=1)
    ref = np.zeros((B, 2))
    return frame, ref, ocp.computeOptimalTrajectory(frame, ref)


def test_plumbing_with_oracle_backend(built):
    """BASELINE config 0: horizon 20 / dt 0.005 OCP through the YAML facade, batch 1 and 3, no GPU"""
    from tests.support.oracle_backend import OracleCuCaQP
    for B in (1, 3):
        ocp = DoubleIntegratorOCP(_node(), batch=B, qp_solver=OracleCuCaQP(batch=B))
        ocp.deployConstraintsAndAddCost()
        ocp.genSolver()
        assert (ocp.model_.nx, ocp.model_.nu, ocp.model_.n, ocp.model_.m) == (2, 1, 62, 100)
        frame, ref, traj = _tick(ocp, B)
        assert traj.shape == (B, 60)
        assert np.abs(traj[:, :3] - frame).max() < 5e-3                              # first frame pinned
        assert np.abs(ocp.model_.constraints(traj)).max() < 5e-3                     # linear dynamics hold after a full step
        X = traj.reshape(B, 20, 3)
        assert (X[:, 1:, 2] >= -1 - 1e-2).all() and (X[:, 1:, 2] <= 1 + 1e-2).all() and (np.abs(X[:, 1:, 1]) <= 2 + 1e-2).all()
        with pytest.raises(ValueError, match="State dimension mismatch"):
            ocp.computeOptimalTrajectory(np.zeros((B, 2)), ref)
        with pytest.raises(ValueError, match="Reference dimension mismatch"):
            ocp.computeOptimalTrajectory(frame, np.zeros((B, 3)))


def test_facade_model_equals_model_zoo():
    """the compiled facade model produces the same QP data as the hand-written DoubleIntegrator (same formulation)"""
    ocp = DoubleIntegratorOCP(_node(), batch=2, qp_solver=object())
    ocp.deployConstraintsAndAddCost()
    fm = ocp._compile_stage_model()
    zm = models.DoubleIntegrator(20, 0.005)
    rng = np.random.default_rng(1)
    p = rng.normal(size=(2, 2)); x = rng.normal(size=(2, 60))
    lbx = np.tile(np.concatenate(ocp.OCPConfigPtr_.getLowerBounds()), (2, 1)); ubx = np.tile(np.concatenate(ocp.OCPConfigPtr_.getUpperBounds()), (2, 1))
    z = np.zeros((2, 38))
    a = fm.local_system(p, x, lbx, ubx, z, z); b = zm.local_system(p, x, lbx, ubx, z, z)
    for k in ("P", "q", "A", "l", "u"):
        assert np.allclose(getattr(a, k), getattr(b, k), rtol=0, atol=1e-12), k
    assert np.array_equal(a.Ai, b.Ai) and np.array_equal(a.Pi, b.Pi)


@pytest.mark.gpu
def test_plumbing_on_gpu(built):
    from tests.support.oracle_backend import OracleCuCaQP
    B = 16
    g = DoubleIntegratorOCP(_node(), batch=B); g.deployConstraintsAndAddCost(); g.genSolver()
    c = DoubleIntegratorOCP(_node(), batch=B, qp_solver=OracleCuCaQP(batch=B)); c.deployConstraintsAndAddCost(); c.genSolver()
    _, _, tg = _tick(g, B); _, _, tc = _tick(c, B)
    assert np.abs(tg - tc).max() <= 1e-6 * (1 + np.abs(tc).max())
    g.OSQPSolverPtr_.qpSolver_.close()


@pytest.mark.gpu
def test_admm_warm_start_across_sqp_iterations(built):
    """BASELINE config 4's mechanism: cart-pole SQP with the ADMM of each QP started from the previous SQP iteration's
    solution needs fewer ADMM iterations and reaches the same trajectory (within the ADMM tolerance)"""
    from optimal_control_problem_amd.sqp import SQPOptimizationSolver
    B = 32
    mdl, ls, meta = models.make_workload("cartpole", B, N=30)
    arg = dict(lbx=meta["lbx"], ubx=meta["ubx"], lbg=meta["lbg"], ubg=meta["ubg"], p=meta["p"])
    cold = SQPOptimizationSolver(mdl, {"max_iter": 8, "alpha": 0.5}, batch=B)
    warm = SQPOptimizationSolver(mdl, {"max_iter": 8, "alpha": 0.5, "warm_start_admm": True}, batch=B)
    rc = cold.getOptimalSolution(arg); rw = warm.getOptimalSolution(arg)
    ic = np.sum([i.sum() for i in cold.admm_iterations[1:]]); iw = np.sum([i.sum() for i in warm.admm_iterations[1:]])
    assert iw < 0.75 * ic
    # the two runs stop their QPs at eps = 1e-3 from different starts, so weakly-weighted inputs may differ; what must
    # agree is the quality of the SQP iterate: objective and dynamics violation
    assert np.abs(rc["f"] - rw["f"]).max() <= 2e-2 * (1 + np.abs(rc["f"]).max())
    assert np.abs(mdl.constraints(rw["x"])).max() <= 2 * np.abs(mdl.constraints(rc["x"])).max() + 1e-3
    cold.qpSolver_.close(); warm.qpSolver_.close()


def test_ocp_module_snake_case_surface(built):
    """the module the reference's (commented-out) pybind file defines -- ocp_module with SolverType and snake_case methods
    (reference src/pybind/python_bindings.cpp:409-446) -- over the same facade"""
    from optimal_control_problem_amd import ocp_module
    from tests.support.oracle_backend import OracleCuCaQP

    class DI(ocp_module.OptimalControlProblem):
        def deploy_constraints_and_add_cost(self):
            cfg = self.OCPConfigPtr_
            h = cfg.getDt()
            F = lambda s, u: np.stack([s[..., 0] + h * s[..., 1] + 0.5 * h * h * u[..., 0], s[..., 1] + h * u[..., 0]], axis=-1)
            ref = self.set_reference(2)
            for k in range(cfg.getHorizon()):
                self.add_vector_cost([10.0, 1.0], cfg.getVariable(k, "state") - ref)
                self.add_vector_cost([0.1], cfg.getVariable(k, "input"))
            for k in range(cfg.getHorizon() - 1):
                self.add_equation_constraint("dynamics", cfg.getVariable(k + 1, "state"),
                                             ocp_module.Dynamics(F, cfg.getVariable(k, "state"), cfg.getVariable(k, "input")))

    ocp = DI(_node(), batch=2, qp_solver=OracleCuCaQP(batch=2))
    assert ocp.get_solver_type() is ocp_module.CUDA_SQP
    ocp.set_solver_type(ocp_module.SolverType.IPOPT); assert ocp.get_solver_type() is ocp_module.IPOPT
    ocp.set_solver_type(ocp_module.CUDA_SQP)
    ocp.deploy_constraints_and_add_cost()
    assert len(ocp.get_constraints()) == 19 and len(ocp.get_cost_function()) == 40 and len(ocp.get_constraint_lower_bounds()) == 19
    ocp.gen_solver()
    frame, ref, traj = _tick(ocp, 2)
    assert traj.shape == (2, 60) and np.array_equal(ocp.get_optimal_trajectory(), traj)
    lib = ocp.gen_code()                                    # traced dynamics -> gfx950 library (cross-compiles without a GPU)
    import os
    assert os.path.exists(lib) and lib.endswith(".so")
    with pytest.raises(NotImplementedError):
        ocp_module.OptimalControlProblem(_node()).deploy_constraints_and_add_cost()


# ---------------------------------------------------------------------------------------------- general (slow) path
# Anything the stage pattern does not cover -- the reference accepts arbitrary SX over the whole decision vector
# (src/OptimalControlProblem.cpp:444-497) -- is evaluated on the host over the whole vector (general_nlp.GeneralNLP) and its QPs are
# solved by the same engine.
from optimal_control_problem_amd.ocp import General  # noqa: E402

TESTCPP_YAML = """
optimal_control_problem:
  discretization_settings: {dt: 0.1, horizon: 2}
  solver_settings:
    verbose: false
    gen_code: false
    load_lib: false
    max_iter: 1000
    warm_start: true
    solve_method: CUDA_SQP
    SQP_settings: {alpha: 1.0, step_num: 10}
  OCP_variables:
    - name: "x"
      size: %d
      lower_bound: %s
      upper_bound: %s
"""


def _fmt(b):
    return "[" + ", ".join(".inf" if v == np.inf else "-.inf" if v == -np.inf else repr(float(v)) for v in b) + "]"


def _testcpp_through_builders(idx, qp_solver=None, batch=1):
    """the NLPs of the reference's test/test.cpp:13-185 (cases 1-7) stated through the OptimalControlProblem builders: the decision
    variables are frame 1 of a two-frame problem (frame 0 is pinned by computeOptimalTrajectory, reference :93-96, and appears nowhere),
    cost and constraints are General expressions, case 6's parameter is the reference vector.  Returns (solution of frame 1, expected)."""
    mdl, arg, expect = models.reference_test_cases()[idx]
    nx, npar = mdl.nx, mdl.np

    class Problem(OptimalControlProblem):
        def deployConstraintsAndAddCost(self):
            self.setReference(max(npar, 1))
            w_of = lambda X, p: [p[i] for i in range(npar)] + [X[nx + i] for i in range(nx)]           # the case's w = [p; x], x = frame 1
            if idx == 5:
                self.addScalarCost(General(lambda X, p: (X[nx] - p[0]) ** 2 + X[nx + 1] ** 2))
            else:
                c = {0: [0, 0], 1: [3, -2], 2: [2, 3], 3: [0, 0], 4: [1, 2, 3], 6: [3, 4]}[idx]
                self.addScalarCost(General(lambda X, p: sum((X[nx + i] - float(c[i])) ** 2 for i in range(nx))))
            lbg, ubg = np.asarray(arg["lbg"], float), np.asarray(arg["ubg"], float)
            if idx in (0, 2):
                self.addInequalityConstraint("sum", lbg, General(lambda X, p: [X[nx] + X[nx + 1] - 1.0], 1), ubg)
            elif idx == 3:
                self.addInequalityConstraint("each", lbg, General(lambda X, p: [X[nx], X[nx + 1]], 2), ubg)
            elif idx == 4:
                self.addInequalityConstraint("sum", lbg, General(lambda X, p: [X[nx] + X[nx + 1] + X[nx + 2] - 5.0], 1), ubg)
            else:       # the reference refuses a problem without constraints ("Constraints are empty", :231-233): a loose row
                self.addInequalityConstraint("none", [-np.inf], General(lambda X, p: [X[nx]], 1), [np.inf])

    node = yaml.safe_load(TESTCPP_YAML % (nx, _fmt(arg["lbx"]), _fmt(arg["ubx"])))["optimal_control_problem"]
    ocp = Problem(node, batch=batch, qp_solver=qp_solver)
    ocp.deployConstraintsAndAddCost(); ocp.genSolver()
    assert ocp.generalPath_
    p = np.asarray(arg["p"], float) if npar else np.zeros(1)
    x = ocp.computeOptimalTrajectory(np.zeros((batch, nx)), np.tile(p, (batch, 1)))
    return x[:, nx:], np.asarray(expect, float)


@pytest.mark.parametrize("idx", range(7))
def test_general_path_testcpp_cases_with_oracle_backend(built, idx):
    from tests.support.oracle_backend import OracleCuCaQP
    x, expect = _testcpp_through_builders(idx, qp_solver=OracleCuCaQP(batch=1))
    assert np.abs(x[0] - expect).max() < 5e-3, (x, expect)


class SkipCoupledOCP(DoubleIntegratorOCP):
    """the double-integrator OCP plus a constraint between frames k and k + 2 -- |u_{k+2} - u_k| <= d: not a stage pattern
    (frames further apart than neighbours), so genSolver() takes the general path"""
    d = 0.05

    def deployConstraintsAndAddCost(self):
        super().deployConstraintsAndAddCost()
        cfg = self.OCPConfigPtr_
        for k in range(cfg.getHorizon() - 2):
            self.addInequalityConstraint("skip", [-self.d], cfg.getVariable(k + 2, "input") - cfg.getVariable(k, "input"), [self.d])


def _skip_node():
    node = _node()
    node["discretization_settings"]["dt"] = 0.05; node["discretization_settings"]["horizon"] = 10
    return node


def test_general_path_constraint_coupling_frames_k_and_k_plus_2(built):
    from tests.support.oracle_backend import OracleCuCaQP
    B = 3
    ocp = SkipCoupledOCP(_skip_node(), batch=B, qp_solver=OracleCuCaQP(batch=B))
    ocp.deployConstraintsAndAddCost(); ocp.genSolver()
    assert ocp.generalPath_ and "dynamics defects" in ocp.generalPathReason_
    m = ocp.model_
    assert (m.n, m.ng) == (2 + 30, 18 + 8)
    # exact structure, not dense: a dynamics row touches its two frames, a skip row two inputs; the Hessian is diagonal in X plus the p coupling
    jm = m.am[m.n:]
    assert jm[:18].sum(axis=1).max() <= 4 and (jm[18:].sum(axis=1) == 2).all()
    assert m.hm.sum() == 2 + 20 + 2 * 20 + 10          # p block diagonal, state diagonals, state-p couplings, input diagonals
    frame = np.array([[1.0, 0.0, 0.0], [0.5, -0.2, 0.0], [-1.0, 0.3, 0.0]]); ref = np.zeros((B, 2))
    x = ocp.computeOptimalTrajectory(frame, ref).reshape(B, 10, 3)
    assert np.abs(x[:, 0] - frame).max() < 2e-3
    du2 = x[:, 2:, 2] - x[:, :-2, 2]
    assert du2.max() <= SkipCoupledOCP.d + 5e-3 and du2.min() >= -SkipCoupledOCP.d - 5e-3
    assert np.abs(du2).max() > 0.5 * SkipCoupledOCP.d                  # the constraint binds
    # the same problem without the skip rows is the stage pattern: the general path on IT must agree with the compiled stage model
    plain = DoubleIntegratorOCP(_skip_node(), batch=B, qp_solver=OracleCuCaQP(batch=B))
    plain.deployConstraintsAndAddCost(); plain.genSolver()
    assert not plain.generalPath_
    forced = DoubleIntegratorOCP(_skip_node(), batch=B, qp_solver=OracleCuCaQP(batch=B))
    forced.deployConstraintsAndAddCost(); forced._compile_stage_model = lambda: (_ for _ in ()).throw(NotImplementedError("forced"))
    forced.genSolver()
    assert forced.generalPath_
    xa = plain.computeOptimalTrajectory(frame, ref); xb = forced.computeOptimalTrajectory(frame, ref)
    assert np.abs(xa - xb).max() < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("idx", range(7))
def test_general_path_testcpp_cases_on_gpu(built, idx):
    """the same seven NLPs through the builders with the real engine, against the oracle-backed run"""
    from tests.support.oracle_backend import OracleCuCaQP
    xg, expect = _testcpp_through_builders(idx)
    xo, _ = _testcpp_through_builders(idx, qp_solver=OracleCuCaQP(batch=1))
    assert np.abs(xg[0] - expect).max() < 5e-3 and np.abs(xg - xo).max() < 1e-6


@pytest.mark.gpu
def test_general_path_skip_coupling_on_gpu(built):
    from tests.support.oracle_backend import OracleCuCaQP
    B = 4
    frame = np.array([[1.0, 0.0, 0.0], [0.5, -0.2, 0.0], [-1.0, 0.3, 0.0], [0.2, 0.1, 0.0]]); ref = np.zeros((B, 2))
    out = []
    for qp in (None, OracleCuCaQP(batch=B)):
        ocp = SkipCoupledOCP(_skip_node(), batch=B, qp_solver=qp)
        ocp.deployConstraintsAndAddCost(); ocp.genSolver()
        assert ocp.generalPath_
        out.append(ocp.computeOptimalTrajectory(frame, ref))
    assert np.abs(out[0] - out[1]).max() < 1e-6


def test_general_path_without_a_reference_vector_and_gen_code(built):
    """no setReference(): the parameter vector is empty (np = 0), as with an SX that was never assigned; genCode() on a problem that took the
    general path says why it cannot compile it for the device"""
    from tests.support.oracle_backend import OracleCuCaQP

    class P(OptimalControlProblem):
        def deployConstraintsAndAddCost(self):
            self.addScalarCost(General(lambda X, p: (X[2] - 1.0) ** 2 + (X[3] + 2.0) ** 2 + X[2] * X[3]))
            self.addInequalityConstraint("sum", [-np.inf], General(lambda X, p: [X[2] + X[3]], 1), [0.5])

    node = yaml.safe_load(TESTCPP_YAML % (2, "[-10.0, -10.0]", "[10.0, 10.0]"))["optimal_control_problem"]
    ocp = P(node, batch=1, qp_solver=OracleCuCaQP(batch=1))
    ocp.deployConstraintsAndAddCost(); ocp.genSolver()
    assert ocp.generalPath_ and ocp.model_.np == 0
    x = ocp.computeOptimalTrajectory(np.zeros((1, 2)), np.zeros((1, 0)))
    # min (a - 1)^2 + (b + 2)^2 + a b  s.t. a + b <= 0.5: the unconstrained optimum (8/3, -10/3) is feasible
    assert np.abs(x[0, 2:] - np.array([8.0 / 3, -10.0 / 3])).max() < 5e-3
    with pytest.raises(NotImplementedError, match="general path"):
        ocp.genCode()
    with pytest.raises(ValueError, match="Reference dimension mismatch"):
        ocp.computeOptimalTrajectory(np.zeros((1, 2)), np.zeros((1, 1)))
