"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Tolerances (fp64): the two sides run the same ADMM on the same QP but factorise differently (GPU: block
Cholesky of the reduced matrix with FMA contraction and wave-tree reductions; oracle: scalar sparse LDL' of
the KKT matrix), so iterates agree to rounding amplified by the conditioning of the KKT system.  Stated
bar: identical status and iteration count, |x_gpu - x_oracle| <= 1e-6 * (1 + |x|_inf) and the same for y.
Against KKT-verified optima both must be within the ADMM tolerance the reference configures
(eps_abs = eps_rel = 1e-3, reference src/sqp_solver/SQPOptimizationSolver.cpp:83-84)."""
import numpy as np
import pytest

from optimal_control_problem_amd import models
from tests.support import problems

pytestmark = pytest.mark.gpu

RTOL = 1e-6


def _close(got, ref, key, tol=RTOL):
    a, b = got[key], ref[key]
    ok = np.isfinite(b)
    assert (np.isfinite(a) == ok).all(), "NaN pattern differs for %s" % key
    if ok.any():
        scale = 1.0 + np.abs(b[ok]).max()
        err = np.abs(a[ok] - b[ok]).max()
        assert err <= tol * scale, "%s: err %.3e > %.1e * %.3e" % (key, err, tol, scale)


def _compare(ls, **settings):
    from optimal_control_problem_amd.batch_qp import solve_local_system
    got = solve_local_system(ls, **settings)
    ref = problems.oracle_solve(ls, **settings)
    assert (got["status"] == ref["status"]).all(), (got["status"], ref["status"])
    assert (got["iters"] == ref["iters"]).all(), (got["iters"], ref["iters"])
    for k in ("x", "y", "z"):
        _close(got, ref, k)
    return got, ref


def test_blockops(built):
    """matrix-core 16x16x16 product and the in-LDS Cholesky + inverse, against NumPy (asymmetric operands)."""
    from optimal_control_problem_amd import _lib
    rng = np.random.default_rng(0)
    A = rng.normal(size=(16, 16)); B = rng.normal(size=(16, 16)); Cm = rng.normal(size=(16, 16))
    G = rng.normal(size=(16, 16)); S = G @ G.T + 16 * np.eye(16)
    og = np.zeros((16, 16)); ol = np.zeros((16, 16)); fail = np.zeros(1, np.int32)
    _lib.check(_lib.lib().mpcqp_debug_blockops(A.ctypes.data, B.ctypes.data, Cm.ctypes.data, S.ctypes.data,
                                               og.ctypes.data, ol.ctypes.data, fail.ctypes.data))
    assert fail[0] == 0
    assert np.abs(og - (Cm - A @ B.T)).max() < 1e-12
    Linv = np.linalg.inv(np.linalg.cholesky(S))
    assert np.abs(ol - Linv).max() < 1e-12
    # indefinite block must be rejected
    S2 = S.copy(); S2[5, 5] = -1.0
    _lib.check(_lib.lib().mpcqp_debug_blockops(A.ctypes.data, B.ctypes.data, Cm.ctypes.data, S2.ctypes.data,
                                               og.ctypes.data, ol.ctypes.data, fail.ctypes.data))
    assert fail[0] == 1


@pytest.mark.parametrize("idx", range(7))
def test_reference_test_cpp_cases(built, idx):
    """First-iteration QPs of the reference's test/test.cpp cases 1-7 (test/test.cpp:13-185)."""
    mdl, arg, expected = models.reference_test_cases()[idx]
    ls = problems.toy_local_system(mdl, arg)
    got, _ = _compare(ls)
    assert got["status"][0] == 1
    # QP step from x = 0 equals the NLP optimum for these (quadratic objective, linear constraints)
    assert np.abs(got["x"][0, mdl.np:] - np.asarray(expected)).max() < 5e-3


def test_reference_case8_nonconvex(built):
    """test/test.cpp:187-211: indefinite Hessian must be reported, not solved."""
    from optimal_control_problem_amd.batch_qp import solve_local_system
    mdl, arg, _ = models.reference_test_cases()[7]
    got = solve_local_system(problems.toy_local_system(mdl, arg))
    assert got["status"][0] == 9 and np.isnan(got["x"]).all()


@pytest.mark.parametrize("name,batch", [("double_integrator", 96), ("quadrotor", 48), ("cartpole", 8)])
def test_workloads_vs_oracle(built, name, batch):
    mdl, ls, _ = models.make_workload(name, batch)
    got, ref = _compare(ls)
    assert (got["status"] == 1).all()
    _close(got, ref, "obj", 1e-6)


def test_quadrotor_n50(built):
    mdl, ls, _ = models.make_workload("quadrotor", 6, N=50)
    _compare(ls)


@pytest.mark.parametrize("seed", range(6))
def test_random_dense_qps(built, seed):
    ls = problems.random_qp(11 + 3 * seed, 17 + 5 * seed, seed)
    _compare(ls)


def test_infeasible_and_settings(built):
    got, _ = _compare(problems.random_qp(12, 20, 101, infeasible="primal"))
    assert got["status"][0] == 3
    got, _ = _compare(problems.random_qp(12, 20, 102, infeasible="dual"))
    assert got["status"][0] == 5
    ls = problems.random_qp(14, 22, 7)
    _compare(ls, scaling=0)
    _compare(ls, max_iter=20, adaptive_rho=0)
    _compare(ls, adaptive_rho_interval=25, eps_abs=1e-6, eps_rel=1e-6)
    _compare(ls, scaled_termination=1)


def test_shared_matrices_and_device_pointers(built):
    """P, A shared across the batch (stride 0) and inputs resident in HBM (torch tensors) give the same answers."""
    import torch
    from optimal_control_problem_amd.batch_qp import BatchQP
    mdl, ls, _ = models.make_workload("double_integrator", 32)
    ref = problems.oracle_solve(ls)
    qp = BatchQP(ls.n, ls.m, ls.batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    qp.update(ls.P[0], ls.q, ls.A[0], ls.l, ls.u)          # LTI: every instance has the same P and A
    qp.solve(); got = qp.get()
    assert (got["iters"] == ref["iters"]).all()
    _close(got, ref, "x")
    dev = [torch.from_numpy(a).cuda() for a in (ls.P, ls.q, ls.A, ls.l, ls.u)]
    qp.update(*dev)
    qp.solve(torch.cuda.current_stream().cuda_stream)
    xd = torch.empty(ls.batch, ls.n, dtype=torch.float64, device="cuda")
    qp.get_device(x=xd)
    torch.cuda.synchronize()
    assert np.abs(xd.cpu().numpy() - got["x"]).max() == 0.0
    assert qp.last_kernel_ms() > 0
    qp.close()


def test_warm_start(built):
    from optimal_control_problem_amd.batch_qp import BatchQP
    mdl, ls, _ = models.make_workload("double_integrator", 8)
    cold = problems.oracle_solve(ls)
    from oracle import oracle as orc
    pat = orc.Pattern(ls.n, ls.m, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    ref = pat.solve(ls.P, ls.q, ls.A, ls.l, ls.u, orc.default_settings(warm_start=1), x0=cold["x"], y0=cold["y"])
    qp = BatchQP(ls.n, ls.m, ls.batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai, warm_start=1)
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.warm_start(cold["x"], cold["y"]); qp.solve()
    got = qp.get(); qp.close()
    assert (got["iters"] == ref["iters"]).all() and (got["iters"] <= cold["iters"]).all()
    _close(got, ref, "x")


@pytest.mark.parametrize("name,B,N", [("cartpole", 24, 30), ("quadrotor", 8, 10)])
def test_starting_rho_per_instance(built, name, B, N):
    """mpcqp_set_rho: per-instance starting rho (what a kept OSQP workspace carries over) against the oracle, host and
    device pointers, mixed with entries <= 0 that fall back to settings.rho; NULL restores the default"""
    from optimal_control_problem_amd.batch_qp import BatchQP
    from oracle import oracle as orc
    mdl, ls, _ = models.make_workload(name, B, N=N)
    cold = problems.oracle_solve(ls)
    rho0 = cold["rho"].copy(); rho0[::3] = 0.0; rho0[1::3] *= 7.0
    pat = orc.Pattern(ls.n, ls.m, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    ref = pat.solve(ls.P, ls.q, ls.A, ls.l, ls.u, orc.default_settings(), rho0=rho0)
    assert (ref["iters"][::3] == cold["iters"][::3]).all()
    qp = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.set_rho(rho0); qp.solve(); got = qp.get()
    assert (got["status"] == ref["status"]).all() and (got["iters"] == ref["iters"]).all()
    _close(got, ref, "x"); _close(got, ref, "y")
    import torch
    qp.set_rho(torch.as_tensor(rho0, device="cuda")); qp.solve(); dev = qp.get()
    assert np.array_equal(dev["x"], got["x"]) and np.array_equal(dev["iters"], got["iters"])
    qp.set_rho(None); qp.solve(); again = qp.get(); qp.close()
    assert (again["iters"] == cold["iters"]).all()
    _close(again, cold, "x")


def test_error_behaviour(built):
    """bool/err-code behaviour mirroring CuCaQP's checks (reference src/sqp_solver/CuCaQP.cpp:23-27,49-52,199-203)."""
    from optimal_control_problem_amd import _lib
    from optimal_control_problem_amd.batch_qp import BatchQP
    mdl, ls, _ = models.make_workload("double_integrator", 2)
    with pytest.raises(_lib.MpcqpError) as e:
        BatchQP(0, 3, 1, np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(1, np.int32), np.zeros(0, np.int32))
    assert e.value.code == _lib.ERR_ARG
    qp = BatchQP(ls.n, ls.m, ls.batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    with pytest.raises(_lib.MpcqpError) as e:
        qp.solve()
    assert e.value.code == _lib.ERR_STATE
    with pytest.raises(ValueError):
        qp.update(ls.P[:, :-1], ls.q, ls.A, ls.l, ls.u)
    qp.close()


# ---------------------------------------------------------------------------------------------- golden fixtures
from tests.support import golden  # noqa: E402

_FIX = golden.load()


@pytest.mark.parametrize("name", golden.NAMES)
def test_golden_fixtures(built, name):
    """committed fixtures: same status / iterations / x as the pinned oracle run, and within ADMM accuracy of the
    KKT-verified optimum"""
    from optimal_control_problem_amd.batch_qp import solve_local_system
    f = _FIX[name]
    got = solve_local_system(f["ls"])
    assert (got["status"] == f["oracle"]["status"]).all() and (got["iters"] == f["oracle"]["iters"]).all()
    ok = np.isfinite(f["oracle"]["x"])
    assert (np.isfinite(got["x"]) == ok).all()
    if ok.any():
        assert np.abs(got["x"][ok] - f["oracle"]["x"][ok]).max() <= RTOL * (1 + np.abs(f["oracle"]["x"][ok]).max())
        assert np.abs(got["y"][np.isfinite(f["oracle"]["y"])] - f["oracle"]["y"][np.isfinite(f["oracle"]["y"])]).max() <= RTOL * (1 + np.abs(f["oracle"]["y"][np.isfinite(f["oracle"]["y"])]).max())
    if np.isfinite(f["x_star"]).all():
        tight = solve_local_system(f["ls"], eps_abs=1e-9, eps_rel=1e-9, max_iter=200000)
        assert (tight["status"] == 1).all()
        assert np.abs(tight["x"] - f["x_star"]).max() <= 1e-5 * (1 + np.abs(f["x_star"]).max())


# ---------------------------------------------------------------------------------------------- host API on the GPU
def test_cucaqp_call_sequence_and_errors(built, capsys):
    """setDimension -> settings -> setSystem -> initSolver -> solve -> getSolution (reference
    SQPOptimizationSolver.cpp:80-85,155-167); bool + stderr error behaviour of CuCaQP.cpp:23-27,49-52,199-203"""
    from optimal_control_problem_amd.cucaqp import CuCaQP
    mdl, arg, expected = models.reference_test_cases()[0]
    ls = problems.toy_local_system(mdl, arg)
    qp = CuCaQP()
    assert qp.setDimension(0, 3) is False and "Invalid dimensions" in capsys.readouterr().err
    assert qp.solve() is False and "not initialized" in capsys.readouterr().err
    assert qp.setDimension(ls.n, ls.m) is True
    qp.setVerbosity(False); qp.setWarmStart(True); qp.setAbsoluteTolerance(1e-3); qp.setRelativeTolerance(1e-3); qp.setMaxIteration(10000)
    assert qp.setGradient(np.zeros(ls.n + 1)) is False and "size mismatch" in capsys.readouterr().err
    qp.setSystem([(ls.Pp, ls.Pi, ls.P[0]), ls.q[0], (ls.Ap, ls.Ai, ls.A[0]), ls.l[0], ls.u[0]])
    assert qp.initSolver() is True and qp.solve() is True
    assert np.abs(qp.getSolution()[0] - expected).max() < 5e-3
    assert qp.getStatus()[0] == 1 and qp.getIterations()[0] == 25
    qp.printSolverData()
    assert "scaling c" in capsys.readouterr().out
    # crossed bounds: osqp_setup refuses them, so initSolver returns false and solve() has nothing to solve (CuCaQP.cpp:183-203)
    lo = ls.l[0].copy(); up = ls.u[0].copy(); lo[0], up[0] = 1.0, -1.0
    qp.setSystem([(ls.Pp, ls.Pi, ls.P[0]), ls.q[0], (ls.Ap, ls.Ai, ls.A[0]), lo, up])
    assert qp.initSolver() is False and "lower bound greater than upper bound" in capsys.readouterr().err
    assert qp.solve() is False and "not initialized" in capsys.readouterr().err
    qp.close()


@pytest.mark.parametrize("idx", range(7))
def test_sqp_driver_on_gpu_testcpp(built, idx):
    """the reference's SQP outer loop over the GPU QP: expected optima of test/test.cpp cases 1-7"""
    from optimal_control_problem_amd.sqp import SQPOptimizationSolver
    mdl, arg, expected = models.reference_test_cases()[idx]
    s = SQPOptimizationSolver(mdl, {"max_iter": 3, "alpha": 1.0, "verbose": False})
    res = s.getOptimalSolution({k: np.asarray(v, float) for k, v in arg.items()})
    assert np.abs(res["x"][0] - np.asarray(expected)).max() < 5e-3
    s.qpSolver_.close()


def test_sqp_driver_gpu_equals_oracle_backend(built):
    """batched nonlinear MPC tick (quadrotor, 10 damped SQP steps as the reference defaults): GPU QP backend and
    oracle backend produce the same trajectory"""
    from optimal_control_problem_amd.sqp import SQPOptimizationSolver
    from tests.support.oracle_backend import OracleCuCaQP
    B = 8
    mdl, ls, meta = models.make_workload("quadrotor", B, N=10)
    arg = dict(lbx=meta["lbx"], ubx=meta["ubx"], lbg=meta["lbg"], ubg=meta["ubg"], p=meta["p"])
    a = SQPOptimizationSolver(mdl, {"max_iter": 10, "alpha": 0.5}, batch=B)
    b = SQPOptimizationSolver(mdl, {"max_iter": 10, "alpha": 0.5}, batch=B, qp_solver=OracleCuCaQP(batch=B))
    ra = a.getOptimalSolution(arg); rb = b.getOptimalSolution(arg)
    assert np.abs(ra["x"] - rb["x"]).max() <= 1e-6 * (1 + np.abs(rb["x"]).max())
    assert np.abs(mdl.constraints(ra["x"])).max() < np.abs(mdl.constraints(np.zeros_like(ra["x"]))).max()
    a.qpSolver_.close()


# ---------------------------------------------------------------------------------------------- full size
def test_full_size_quadrotor_properties(built):
    """BASELINE.json north-star size (12-state quadrotor, N = 20, batch 8192): size-independent properties --
    every instance solved, batch-order independence (bitwise), duplicated instances identical, returned (x, y, z)
    satisfy the unscaled termination test recomputed on the host, and a sample equals the oracle."""
    from optimal_control_problem_amd.batch_qp import BatchQP
    B = 8192
    mdl, ls, _ = models.make_workload("quadrotor", B)
    qp = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); got = qp.get()
    assert (got["status"] == 1).all()
    # reversed order + instance 0 duplicated at the end
    idx = np.arange(B)[::-1].copy(); idx[-1] = B - 1; idx[0] = B - 1
    qp.update(ls.P[idx], ls.q[idx], ls.A[idx], ls.l[idx], ls.u[idx]); qp.solve(); rev = qp.get()
    assert np.array_equal(rev["x"], got["x"][idx]) and np.array_equal(rev["iters"], got["iters"][idx])
    qp.close()
    # host recomputation of the residuals from the returned point
    x, y, z = got["x"], got["y"], got["z"]
    Ax = np.zeros((B, ls.m)); Aty = np.zeros((B, ls.n)); Px = np.zeros((B, ls.n))
    for j in range(ls.n):
        for k in range(ls.Ap[j], ls.Ap[j + 1]):
            Ax[:, ls.Ai[k]] += ls.A[:, k] * x[:, j]; Aty[:, j] += ls.A[:, k] * y[:, ls.Ai[k]]
        for k in range(ls.Pp[j], ls.Pp[j + 1]):
            if ls.Pi[k] <= j:
                Px[:, ls.Pi[k]] += ls.P[:, k] * x[:, j]
                if ls.Pi[k] != j:
                    Px[:, j] += ls.P[:, k] * x[:, ls.Pi[k]]
    prim = np.abs(Ax - z).max(axis=1); dual = np.abs(Px + ls.q + Aty).max(axis=1)
    eps_p = 1e-3 + 1e-3 * np.maximum(np.abs(Ax).max(axis=1), np.abs(z).max(axis=1))
    eps_d = 1e-3 + 1e-3 * np.maximum(np.abs(Px).max(axis=1), np.maximum(np.abs(Aty).max(axis=1), np.abs(ls.q).max(axis=1)))
    assert (prim <= eps_p * (1 + 1e-9)).all() and (dual <= eps_d * (1 + 1e-9)).all()
    assert np.abs(prim - got["prim_res"]).max() < 1e-9 and np.abs(dual - got["dual_res"]).max() < 1e-7
    lo = np.maximum(ls.l, -1e30); hi = np.minimum(ls.u, 1e30)
    assert (z >= lo - 1e-9).all() and (z <= hi + 1e-9).all()          # z is a projection onto [l, u]
    sl = slice(0, 64)
    sub = models.LocalSystem(ls.n, ls.m, ls.Pp, ls.Pi, ls.Ap, ls.Ai, ls.P[sl], ls.q[sl], ls.A[sl], ls.l[sl], ls.u[sl])
    ref = problems.oracle_solve(sub)
    assert (ref["iters"] == got["iters"][sl]).all()
    assert np.abs(ref["x"] - got["x"][sl]).max() < 1e-6


# ---------------------------------------------------------------------------------------------- kernel variants
@pytest.mark.parametrize("variant", ["stream", "res1", "res2", "res4", "res8", "gres4", "gres2"])
@pytest.mark.parametrize("name,batch,N", [("double_integrator", 40, 20), ("quadrotor", 24, 20), ("cartpole", 6, 30)])
def test_kernel_variants_vs_oracle(built, monkeypatch, variant, name, batch, N):
    """every kernel family (HBM-streamed factor; LDS-resident block LDL' with 1 / 4 / 8 waves per QP) against the oracle"""
    from optimal_control_problem_amd.batch_qp import BatchQP
    monkeypatch.setenv("MPCQP_VARIANT", variant)
    mdl, ls, _ = models.make_workload(name, batch, N=N)
    qp = BatchQP(ls.n, ls.m, ls.batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    assert qp.plan_info()["variant"] == {"stream": 0, "res1": 1, "res2": 2, "res4": 4, "res8": 8, "gres4": 104, "gres2": 102}[variant]
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); got = qp.get(); qp.close()
    ref = problems.oracle_solve(ls)
    assert (got["status"] == ref["status"]).all() and (got["iters"] == ref["iters"]).all()
    for k in ("x", "y", "z"):
        _close(got, ref, k)


@pytest.mark.parametrize("name,batch,N", [("quadrotor", 24, 20), ("quadrotor", 7, 12), ("cartpole", 6, 30), ("double_integrator", 40, 20),
                                          ("double_integrator", 9, 60), ("cartpole", 5, 50)])
def test_onchip_variant_vs_oracle(built, monkeypatch, name, batch, N):
    """the on-chip mode (factor in LDS + registers, triangular solves on the matrix cores; kernel_onchip.hpp): twisted two-chain
    plans with a hub (quadrotor), single chains whose hub shares the last block (cart-pole), tiny plans, runs long enough for an
    adaptive-rho refactorisation (double integrator) -- same bar as every other family"""
    from optimal_control_problem_amd.batch_qp import BatchQP
    monkeypatch.setenv("MPCQP_VARIANT", "oc4")
    mdl, ls, _ = models.make_workload(name, batch, N=N)
    qp = BatchQP(ls.n, ls.m, ls.batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    info = qp.plan_info()
    assert info["variant"] == 204 and info["lds_bytes"] <= 80 * 1024
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); got = qp.get(); qp.close()
    ref = problems.oracle_solve(ls)
    assert (got["status"] == ref["status"]).all() and (got["iters"] == ref["iters"]).all()
    for k in ("x", "y", "z"):
        _close(got, ref, k)


@pytest.mark.parametrize("name,batch,N", [("quadrotor", 12, 50), ("cartpole", 12, 100), ("quadrotor", 9, 30), ("cartpole", 7, 70), ("double_integrator", 12, 100),
                                          ("quadrotor", 6, 25), ("quadrotor", 5, 40), ("cartpole", 4, 150)])
def test_onchip_long_chains_vs_oracle(built, monkeypatch, name, batch, N):
    """the eight-wave on-chip instances (one workgroup per CU; chain loops of any length; padded twist for chains whose last block is
    partly filled -- cart-pole -- ; every hub block in registers up to 32 chain blocks, beyond that five per wave with z and y in the
    slab): BASELINE configs 3 and 4 and the sizes around them, same bar as every other family"""
    from optimal_control_problem_amd.batch_qp import BatchQP
    monkeypatch.setenv("MPCQP_VARIANT", "oc8")
    mdl, ls, _ = models.make_workload(name, batch, N=N)
    qp = BatchQP(ls.n, ls.m, ls.batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    info = qp.plan_info()
    assert info["variant"] == 208 and info["lds_bytes"] <= 160 * 1024
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); got = qp.get(); qp.close()
    ref = problems.oracle_solve(ls)
    assert (got["status"] == ref["status"]).all() and (got["iters"] == ref["iters"]).all()
    for k in ("x", "y", "z"):
        _close(got, ref, k)


def test_onchip_long_chains_are_the_default_for_configs_3_and_4(built):
    from optimal_control_problem_amd.batch_qp import BatchQP
    for name, N, B in [("quadrotor", 50, 8192), ("cartpole", 100, 16384)]:
        mdl, ls, _ = models.make_workload(name, 2, N=N)
        qp = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai); info = qp.plan_info(); qp.close()
        assert info["variant"] == 208, (name, info)


@pytest.mark.parametrize("name,B,N", [("quadrotor", 9, 20), ("cartpole", 5, 40)])
def test_onchip_variant_without_hub(built, monkeypatch, name, B, N):
    """the reduced form has no parameter block, so its pattern is block tridiagonal without an arrow head: the on-chip instance
    without hub phases (chains, diagonal, chains), against the oracle on the reduced QP"""
    from optimal_control_problem_amd.batch_qp import BatchQP
    monkeypatch.setenv("MPCQP_VARIANT", "oc4")
    mdl, ls, _ = models.make_workload(name, B, N=N)
    rows = list(range(mdl.np))
    qp = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai, fixed_rows=rows)
    assert qp.plan_info()["variant"] == 204
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); got = qp.get(); qp.close()
    red, free, kept, fvars, xfix = problems.reduce_qp(ls, rows)
    ref = problems.oracle_solve(red)
    assert (got["status"] == ref["status"]).all() and (got["iters"] == ref["iters"]).all()
    assert np.abs(got["x"][:, free] - ref["x"]).max() <= 1e-6 * (1.0 + np.abs(ref["x"]).max())
    assert np.abs(got["y"][:, kept] - ref["y"]).max() <= 1e-6 * (1.0 + np.abs(ref["y"]).max())


def test_onchip_is_the_default_for_the_north_star_size(built):
    from optimal_control_problem_amd.batch_qp import BatchQP
    mdl, ls, _ = models.make_workload("quadrotor", 2, N=20)
    qp = BatchQP(ls.n, ls.m, 8192, ls.Pp, ls.Pi, ls.Ap, ls.Ai); info = qp.plan_info(); qp.close()
    assert info["variant"] == 204 and info["lds_bytes"] <= 80 * 1024
    # ... and wherever the alternative is a factor streamed from the slab or the LDS-resident kernel at two workgroups per CU (mpcqp.hip
    # selection rule); with three or more resident workgroups per CU the LDS-resident kernels stay
    for name, N, want in (("quadrotor", 10, 204), ("cartpole", 40, 204), ("double_integrator", 60, 204), ("quadrotor", 7, 204),
                          ("quadrotor", 5, 4), ("cartpole", 20, 4), ("double_integrator", 20, 2)):
        mdl, ls, _ = models.make_workload(name, 2, N=N)
        qp = BatchQP(ls.n, ls.m, 8192, ls.Pp, ls.Pi, ls.Ap, ls.Ai); info = qp.plan_info(); qp.close()
        assert info["variant"] == want, (name, N, info["variant"])


@pytest.mark.parametrize("knob", ["MPCQP_LATE", "MPCQP_NO_REMAP", "MPCQP_NO_TOUCH", "MPCQP_OC_PAD4"])
def test_onchip_scheduling_knobs_change_no_result(built, monkeypatch, knob):
    """which wave computes which rows of the right-hand side and when (late rows, ticket), which wave plays which part, the L2 touch: none of
    it may change a bit of the output; the ELL padding only adds zero slots (same sums)"""
    from optimal_control_problem_amd.batch_qp import BatchQP
    monkeypatch.setenv("MPCQP_VARIANT", "oc4")           # (a batch of 64 alone would take the LDS-resident kernel)
    mdl, ls, _ = models.make_workload("quadrotor", 64, N=20)
    def run():
        qp = BatchQP(ls.n, ls.m, 64, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
        assert qp.plan_info()["variant"] == 204
        qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); out = qp.get(); qp.close()
        return out
    ref = run()
    monkeypatch.setenv(knob, "1")
    got = run()
    for key in ("x", "y", "z"):
        assert np.array_equal(ref[key], got[key]), key
    assert np.array_equal(ref["iters"], got["iters"]) and np.array_equal(ref["status"], got["status"])


@pytest.mark.parametrize("variant", ["stream", "res1", "res2", "res4", "gres4", "gres2"])
def test_kernel_variants_hard_cases(built, monkeypatch, variant):
    """adaptive-rho refactorisation, infeasibility certificates, max-iter exit, non-convex rejection per kernel family"""
    monkeypatch.setenv("MPCQP_VARIANT", variant)
    for seed in range(6):
        _compare(problems.random_qp(11 + 3 * seed, 17 + 5 * seed, seed))
    assert _compare(problems.random_qp(12, 20, 101, infeasible="primal"))[0]["status"][0] == 3
    assert _compare(problems.random_qp(12, 20, 102, infeasible="dual"))[0]["status"][0] == 5
    ls = problems.random_qp(14, 22, 7)
    _compare(ls, scaling=0); _compare(ls, max_iter=20, adaptive_rho=0); _compare(ls, adaptive_rho_interval=25, eps_abs=1e-6, eps_rel=1e-6)
    from optimal_control_problem_amd.batch_qp import solve_local_system
    mdl, arg, _ = models.reference_test_cases()[7]
    assert solve_local_system(problems.toy_local_system(mdl, arg))["status"][0] == 9
    mdl, ls, _ = models.make_workload("double_integrator", 8)
    cold = problems.oracle_solve(ls)
    from oracle import oracle as orc
    pat = orc.Pattern(ls.n, ls.m, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    ref = pat.solve(ls.P, ls.q, ls.A, ls.l, ls.u, orc.default_settings(warm_start=1), x0=cold["x"], y0=cold["y"])
    from optimal_control_problem_amd.batch_qp import BatchQP
    qp = BatchQP(ls.n, ls.m, ls.batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai, warm_start=1)
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.warm_start(cold["x"], cold["y"]); qp.solve()
    got = qp.get(); qp.close()
    assert (got["iters"] == ref["iters"]).all()
    _close(got, ref, "x")


# ---------------------------------------------------------------------------------------------- kept workspace
@pytest.mark.parametrize("variant,name,B,N", [(None, "double_integrator", 24, 20), ("res1", "double_integrator", 24, 20),
                                              ("gres4", "double_integrator", 24, 20), ("res4", "cartpole", 12, 30),
                                              ("gres4", "quadrotor", 10, 10), ("oc4", "quadrotor", 10, 20), ("oc4", "double_integrator", 24, 20),
                                              ("oc8", "quadrotor", 6, 30), ("oc8", "cartpole", 6, 100)])
def test_kept_workspace_vectors_vs_oracle(built, monkeypatch, variant, name, B, N):
    """mpcqp_keep_workspace + mpcqp_update_vectors (OSQP's osqp_update_data_vec on a kept workspace: scaling, factor and the
    adapted rho stay) against the oracle's kept workspaces, on every kernel family: a full solve, a q/l/u-only solve, a
    solve after a row changed from equality to inequality (factor rebuilt), and a return to the full path"""
    from optimal_control_problem_amd.batch_qp import BatchQP
    from oracle import oracle as orc
    if variant:
        monkeypatch.setenv("MPCQP_VARIANT", variant)
    mdl, ls, meta = models.make_workload(name, B, N=N)
    pat = orc.Pattern(ls.n, ls.m, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    st = orc.State(pat, B, orc.default_settings())
    qp = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    qp.keep_workspace(True)

    def same(got, ref):
        assert (got["status"] == ref["status"]).all(), (got["status"], ref["status"])
        assert (got["iters"] == ref["iters"]).all(), (got["iters"], ref["iters"])
        for k in ("x", "y", "z"):
            _close(got, ref, k)
        assert np.abs(got["rho"] - ref["rho"]).max() <= 1e-6 * np.abs(ref["rho"]).max()

    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); same(qp.get(), st.solve(ls.P, ls.q, ls.A, ls.l, ls.u))
    rng = np.random.default_rng(11)
    frame0 = meta["frame0"].copy(); frame0[:, :mdl.nx] += rng.normal(0, 0.05, (B, mdl.nx))
    lbx, ubx, lbg, ubg = mdl.stacked_bounds(frame0)
    ls2 = mdl.local_system(meta["p"] + 0.1, meta["x_iterate"], lbx, ubx, lbg, ubg)       # same iterate -> same A, P; new q, l, u
    assert np.array_equal(ls2.A, ls.A)
    qp.update_vectors(ls2.q, ls2.l, ls2.u); qp.solve(); got2 = qp.get()
    same(got2, st.solve_vectors(ls2.q, ls2.l, ls2.u))
    # device pointers, and a class change in one row of every second instance
    import torch
    l3, u3 = ls2.l.copy(), ls2.u.copy()
    l3[::2, mdl.np + 1] -= 0.25; u3[::2, mdl.np + 1] += 0.25
    dq, dl, du = [torch.as_tensor(a, device="cuda") for a in (ls2.q, l3, u3)]
    qp.update_vectors(dq, dl, du); qp.solve(); same(qp.get(), st.solve_vectors(ls2.q, l3, u3))
    # back to a full setup
    qp.update(ls2.P, ls2.q, ls2.A, ls2.l, ls2.u); qp.solve(); same(qp.get(), st.solve(ls2.P, ls2.q, ls2.A, ls2.l, ls2.u))
    qp.close()


def test_kept_workspace_errors_and_nonconvex(built, monkeypatch):
    from optimal_control_problem_amd import _lib
    from optimal_control_problem_amd.batch_qp import BatchQP
    mdl, ls, _ = models.make_workload("double_integrator", 4)
    qp = BatchQP(ls.n, ls.m, 4, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    with pytest.raises(_lib.MpcqpError) as e:
        qp.update_vectors(ls.q, ls.l, ls.u)                      # keep_workspace not enabled
    assert e.value.code == _lib.ERR_STATE
    qp.keep_workspace(True)
    with pytest.raises(_lib.MpcqpError) as e:
        qp.update_vectors(ls.q, ls.l, ls.u)                      # no kept solve yet
    assert e.value.code == _lib.ERR_STATE
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); qp.get()
    with pytest.raises(ValueError, match="dimension mismatch"):
        qp.update_vectors(ls.q[:, :-1], ls.l, ls.u)
    # an instance with an indefinite P stays non-convex through vector updates; the others keep solving
    P = ls.P.copy(); P[1] = -np.abs(P[1]) - 1.0
    qp.update(P, ls.q, ls.A, ls.l, ls.u); qp.solve(); a = qp.get()
    qp.update_vectors(ls.q * 0.5, ls.l, ls.u); qp.solve(); b = qp.get()
    assert a["status"][1] == 9 and b["status"][1] == 9 and np.isnan(b["x"][1]).all()
    assert (np.delete(b["status"], 1) == 1).all()
    qp.close()
    monkeypatch.setenv("MPCQP_VARIANT", "stream")
    qs = BatchQP(ls.n, ls.m, 4, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    with pytest.raises(_lib.MpcqpError) as e:
        qs.keep_workspace(True)
    assert e.value.code == _lib.ERR_LIMIT
    qs.close()


def test_cucaqp_update_members_use_the_kept_workspace(built):
    """the reference's private update* members (CuCaQP.cpp:106-161), working: updateGradient / updateLowerBound / updateUpperBound
    + solve() re-solve without a new setup and agree with the oracle's kept workspace"""
    from optimal_control_problem_amd.cucaqp import CuCaQP
    from oracle import oracle as orc
    B = 6
    mdl, ls, meta = models.make_workload("double_integrator", B)
    qp = CuCaQP(batch=B); qp.setDimension(ls.n, ls.m)
    assert qp.updateGradient(ls.q) is False                      # "Solver not initialized" (CuCaQP.cpp:118-121)
    qp.setSystem(ls); assert qp.initSolver() and qp.solve()
    pat = orc.Pattern(ls.n, ls.m, ls.Pp, ls.Pi, ls.Ap, ls.Ai); st = orc.State(pat, B, orc.default_settings())
    st.solve(ls.P, ls.q, ls.A, ls.l, ls.u)
    q2 = ls.q * 1.3; l2 = ls.l - 0.01; u2 = ls.u + 0.01
    assert qp.updateGradient(q2) and qp.updateLowerBound(l2) and qp.updateUpperBound(u2) and qp.solve()
    ref = st.solve_vectors(q2, l2, u2)
    assert (qp.getIterations() == ref["iters"]).all() and np.abs(qp.getSolution() - ref["x"]).max() < 1e-6
    assert qp.updateGradient(np.zeros(3)) is False               # size mismatch
    qp.close()


def test_cucaqp_updates_reach_the_device_in_every_order(built):
    """the orders the bookkeeping has to survive: a vector update between initSolver() and the first solve(), a matrix update
    (full setup at the next solve), vectors after it (kept workspace again) -- each against a fresh oracle solve of the data that
    must be in effect"""
    from optimal_control_problem_amd.cucaqp import CuCaQP
    B = 4
    mdl, ls, meta = models.make_workload("double_integrator", B)
    mk = lambda P, q, A, l, u: models.LocalSystem(ls.n, ls.m, ls.Pp, ls.Pi, ls.Ap, ls.Ai, P, q, A, l, u)
    qp = CuCaQP(batch=B); qp.setDimension(ls.n, ls.m); qp.setSystem(ls)
    assert qp.initSolver()
    q2 = ls.q * 0.7 + 0.05
    assert qp.updateGradient(q2) and qp.solve()                  # initSolver() -> updateGradient() -> solve(): q2 must be what is solved
    ref = problems.oracle_solve(mk(ls.P, q2, ls.A, ls.l, ls.u))
    assert (qp.getIterations() == ref["iters"]).all() and np.abs(qp.getSolution() - ref["x"]).max() < 1e-6
    P2 = ls.P * 1.5
    assert qp.updateHessianMatrix((ls.Pp, ls.Pi, P2)) and qp.solve()
    ref = problems.oracle_solve(mk(P2, q2, ls.A, ls.l, ls.u))
    assert (qp.getIterations() == ref["iters"]).all() and np.abs(qp.getSolution() - ref["x"]).max() < 1e-6
    A2 = ls.A.copy(); A2[..., np.asarray(ls.Ai) >= ls.n] *= 0.9   # dynamics rows only; the identity rows keep their ones
    assert qp.updateLinearConstraintsMatrix((ls.Ap, ls.Ai, A2)) and qp.updateUpperBound(ls.u + 0.02) and qp.solve()
    ref = problems.oracle_solve(mk(P2, q2, A2, ls.l, ls.u + 0.02))
    assert (qp.getIterations() == ref["iters"]).all() and np.abs(qp.getSolution() - ref["x"]).max() < 1e-6
    qp.close()


def test_dispatch_hint_changes_no_result(built):
    """the longest-first dispatch order (taken from the previous solve's iteration counts) only permutes which workgroup takes
    which instance: bitwise identical outputs with a fresh handle (identity order), with its own history, and with the
    history of a different batch"""
    from optimal_control_problem_amd.batch_qp import BatchQP
    B = 700
    mdl, ls, _ = models.make_workload("double_integrator", B)          # 25 .. 125 iterations: a strongly mixed batch
    qp = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); first = qp.get()
    assert len(np.unique(first["iters"])) >= 3
    qp.solve(); second = qp.get()                                      # now ordered by first["iters"]
    for k in ("x", "y", "z", "status", "iters", "rho"):
        assert np.array_equal(first[k], second[k], equal_nan=True), k
    perm = np.random.default_rng(0).permutation(B)                     # other data: the kept hint is stale, results must not care
    qp.update(ls.P, ls.q[perm], ls.A, ls.l[perm], ls.u[perm]); qp.solve(); third = qp.get()
    assert np.array_equal(third["x"], first["x"][perm]) and np.array_equal(third["iters"], first["iters"][perm])
    qp.close()
    ref = problems.oracle_solve(models.LocalSystem(ls.n, ls.m, ls.Pp, ls.Pi, ls.Ap, ls.Ai, ls.P[:64], ls.q[:64], ls.A[:64], ls.l[:64], ls.u[:64]))
    assert (first["iters"][:64] == ref["iters"]).all()


def test_two_handles_on_two_streams(built):
    """distinct handles are independent (include/mpcqp.h): two batches solved concurrently on two HIP streams give the
    results of solving them one after the other"""
    import torch
    from optimal_control_problem_amd.batch_qp import BatchQP
    mdl, a, _ = models.make_workload("quadrotor", 300, N=10)
    _, b, _ = models.make_workload("cartpole", 500, N=30)
    seq = []
    for ls in (a, b):
        qp = BatchQP(ls.n, ls.m, ls.batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai); qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); seq.append(qp.get()); qp.close()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    qa = BatchQP(a.n, a.m, a.batch, a.Pp, a.Pi, a.Ap, a.Ai); qb = BatchQP(b.n, b.m, b.batch, b.Pp, b.Pi, b.Ap, b.Ai)
    da = [torch.as_tensor(v, device="cuda") for v in (a.P, a.q, a.A, a.l, a.u)]
    db = [torch.as_tensor(v, device="cuda") for v in (b.P, b.q, b.A, b.l, b.u)]
    torch.cuda.synchronize()
    for _ in range(3):                                   # several rounds in flight on both streams
        qa.update(*da); qa.solve(s1.cuda_stream)
        qb.update(*db); qb.solve(s2.cuda_stream)
    ga, gb = qa.get(), qb.get()
    for got, ref in ((ga, seq[0]), (gb, seq[1])):
        assert np.array_equal(got["x"], ref["x"]) and np.array_equal(got["iters"], ref["iters"]) and np.array_equal(got["status"], ref["status"])
    qa.close(); qb.close()


@pytest.mark.parametrize("variant", [None, "res1", "res4", "gres4", "stream"])
def test_random_sparse_patterns(built, monkeypatch, variant):
    """random sparse patterns, sizes and batch sizes through every kernel family (a slice of tools/fuzz_gpu.py, which ran 1351
    such solves: 1329 at the tight bar, 22 tolerance-level on ill-conditioned problems needing hundreds of iterations)"""
    from optimal_control_problem_amd import _lib
    from optimal_control_problem_amd.batch_qp import BatchQP
    if variant:
        monkeypatch.setenv("MPCQP_VARIANT", variant)
    for c in range(12):
        rng = np.random.default_rng(1000 + c)
        n = int(rng.integers(2, 140)); m = int(rng.integers(1, 200)); B = int(rng.integers(1, 9)); dens = float(rng.choice([0.05, 0.15, 0.4, 1.0]))
        ls = problems.sparse_batch(n, m, B, c, dens)
        try:
            qp = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
        except _lib.MpcqpError as e:
            assert e.code == _lib.ERR_LIMIT
            continue
        qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); got = qp.get(); qp.close()
        ref = problems.oracle_solve(ls)
        assert (got["status"] == ref["status"]).all() and (got["iters"] == ref["iters"]).all(), (c, n, m, B, dens)
        _close(got, ref, "x")


def test_pipelined_host_step(built):
    """mpcqp_solve_host (fused setSystem + initSolver + solve + getSolution over pipelined batch slices) returns exactly what
    update + solve + get return, for pinned and pageable inputs, odd slice counts, shared matrices and after other solves"""
    import torch
    from optimal_control_problem_amd.batch_qp import BatchQP
    mdl, ls, _ = models.make_workload("cartpole", 333, N=30)
    qp = BatchQP(ls.n, ls.m, ls.batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); ref = qp.get()
    pin = [torch.from_numpy(a).pin_memory() for a in (ls.P, ls.q, ls.A, ls.l, ls.u)]
    for chunks, args in ((0, pin), (7, pin), (1, pin), (1000, (ls.P, ls.q, ls.A, ls.l, ls.u))):
        got = qp.solve_host(*args, chunks=chunks)
        for k in ("x", "y", "status", "iters"):
            assert np.array_equal(got[k], ref[k], equal_nan=True), (chunks, k)
    after = qp.get()                                     # results also stay on the device
    assert np.array_equal(after["x"], ref["x"]) and np.array_equal(after["z"], ref["z"])
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); again = qp.get()
    assert np.array_equal(again["x"], ref["x"])
    qp.close()
    mdl, ld, _ = models.make_workload("double_integrator", 50)
    qd = BatchQP(ld.n, ld.m, 50, ld.Pp, ld.Pi, ld.Ap, ld.Ai)
    qd.update(ld.P[0], ld.q, ld.A[0], ld.l, ld.u); qd.solve(); r0 = qd.get()
    g0 = qd.solve_host(ld.P[0], ld.q, ld.A[0], ld.l, ld.u, chunks=3)            # matrices shared by the batch (stride 0)
    assert np.array_equal(g0["x"], r0["x"]) and np.array_equal(g0["iters"], r0["iters"])
    with pytest.raises(ValueError, match="dimension mismatch"):
        qd.solve_host(ld.P[0], ld.q[:, :-1], ld.A[0], ld.l, ld.u)
    qd.close()


def test_resident_workgroups_draw_every_instance_once(built, monkeypatch):
    """the eight-wave on-chip instances iterate as resident workgroups that draw instance tickets from one counter (kernel_oc_split.hpp): a batch larger than
    the GPU holds at once, a batch smaller, the pipelined host step's slices (each with its own counter) and a dispatch order all give the results of one
    workgroup per instance -- the single-kernel form of the same build -- bit for bit"""
    from optimal_control_problem_amd.batch_qp import BatchQP
    mdl, ls, _ = models.make_workload("cartpole", 700, N=100)
    monkeypatch.setenv("MPCQP_VARIANT", "oc8")
    qp = BatchQP(ls.n, ls.m, ls.batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    assert qp.plan_info()["variant"] == 208
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); ref = qp.get()
    qp.solve(); hinted = qp.get()                                                   # second solve: longest-first order of the tickets
    host = qp.solve_host(ls.P, ls.q, ls.A, ls.l, ls.u, chunks=5)                    # 140 instances per slice: fewer than resident workgroups
    qp.close()
    monkeypatch.setenv("MPCQP_OC_MONO", "1")
    q1 = BatchQP(ls.n, ls.m, ls.batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    q1.update(ls.P, ls.q, ls.A, ls.l, ls.u); q1.solve(); mono = q1.get(); q1.close()
    assert (ref["status"] == 1).all() and len(np.unique(ref["iters"])) > 2           # (iteration counts differ: what the queue is for)
    for k in ("x", "y", "z", "status", "iters"):
        assert np.array_equal(ref[k], mono[k]) and np.array_equal(hinted[k], ref[k]), k
    for k in ("x", "y", "status", "iters"):
        assert np.array_equal(host[k], ref[k]), k


def test_dissected_order_nonconvex_instance_and_graph_replay(built, monkeypatch):
    """the dissected order of the eight-wave instances (separators of the stage chain in the hub block, several twisted pairs; plan.hpp ordering 4): an instance
    whose P is indefinite is reported non-convex by the pair-by-pair factorisation while its neighbours solve as the oracle does, the padded twist gives the
    same statuses and iteration counts, and a solve captured in a HIP graph (its ticket counters are zeroed by a memset node) replays bit for bit"""
    import torch
    from optimal_control_problem_amd.batch_qp import BatchQP
    mdl, ls, _ = models.make_workload("cartpole", 300, N=100)
    monkeypatch.setenv("MPCQP_VARIANT", "oc8")
    P = np.array(np.broadcast_to(ls.P, (ls.batch, ls.P.shape[-1]))); P[7] = -np.abs(P[7]) - 1.0
    qp = BatchQP(ls.n, ls.m, ls.batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    assert qp.plan_info()["variant"] == 208 and qp.oc_info()["chain_pairs"] == 4 and qp.plan_info()["ordering"] == 4
    qp.update(P, ls.q, ls.A, ls.l, ls.u); qp.solve(); got = qp.get()
    assert got["status"][7] == 9 and np.isnan(got["x"][7]).all() and (np.delete(got["status"], 7) == 1).all()
    ref = problems.oracle_solve(ls)
    keep = np.arange(ls.batch) != 7
    assert (got["iters"][keep] == ref["iters"][keep]).all()
    assert np.abs(got["x"][keep] - ref["x"][keep]).max() <= 1e-6 * max(1.0, np.abs(ref["x"][keep]).max())
    s = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        qp.solve(stream=s.cuda_stream); s.synchronize()
        with torch.cuda.graph(g, stream=s):
            qp.solve(stream=s.cuda_stream)
        g.replay(); s.synchronize()
    again = qp.get()
    for k in ("x", "y", "z", "status", "iters"):
        assert np.array_equal(again[k], got[k], equal_nan=True), k
    qp.close()
    monkeypatch.setenv("MPCQP_NO_DISSECT", "1")
    q0 = BatchQP(ls.n, ls.m, ls.batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    assert q0.oc_info()["chain_pairs"] == 1 and q0.plan_info()["ordering"] == 2      # (the padded twist reports as the twisted order)
    q0.update(P, ls.q, ls.A, ls.l, ls.u); q0.solve(); tw = q0.get(); q0.close()
    assert np.array_equal(tw["status"], got["status"]) and np.array_equal(tw["iters"], got["iters"])
    assert np.nanmax(np.abs(tw["x"] - got["x"])) < 1e-7


@pytest.mark.parametrize("N", [22, 24, 25])
def test_padded_twist_keeps_awkward_horizons_on_chip(built, N):
    """horizons at which the hub variables would share a block with the last frame (cart-pole N=22, 25) or the plain order leaves one long chain (N=24): the
    four-wave on-chip instances take the padded twist (plan.hpp ordering 3) instead of handing the pattern to a slower kernel family -- same bar against the oracle"""
    from optimal_control_problem_amd.batch_qp import BatchQP
    mdl, ls, _ = models.make_workload("cartpole", 6000, N=N)          # (a batch large enough for the throughput rule: the small-batch rule has its own choice)
    qp = BatchQP(ls.n, ls.m, ls.batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    info = qp.plan_info(); oc = qp.oc_info()
    assert info["variant"] == 204 and oc["chain_e"] > 0 and oc["chain_f"] > 0, (info, oc)      # on chip, two chains
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); got = qp.get(); qp.close()
    sub = type(ls)(ls.n, ls.m, ls.Pp, ls.Pi, ls.Ap, ls.Ai, ls.P[:64], ls.q[:64], ls.A[:64], ls.l[:64], ls.u[:64])
    ref = problems.oracle_solve(sub)
    assert (got["status"][:64] == ref["status"]).all() and (got["iters"][:64] == ref["iters"]).all()
    assert np.abs(got["x"][:64] - ref["x"]).max() <= 1e-6 * max(1.0, np.abs(ref["x"]).max())
    assert (got["status"] == 1).all()


@pytest.mark.parametrize("name,N,B", [("quadrotor", 50, 40), ("quadrotor", 100, 12), ("cartpole", 100, 30)])
def test_long_horizons_vs_oracle(built, name, N, B):
    """long horizons run the global-block kernels, the longest ones with z and y in the slab as well (one more workgroup per
    CU): same bar"""
    mdl, ls, _ = models.make_workload(name, B, N=N)
    _compare(ls)
    from optimal_control_problem_amd.batch_qp import BatchQP
    qp = BatchQP(ls.n, ls.m, 4096, ls.Pp, ls.Pi, ls.Ap, ls.Ai); info = qp.plan_info(); qp.close()
    assert info["variant"] >= 100


INF = float("inf")


def _edge_problems():
    """small QPs at the edges of the interface: one variable, no constraints, no quadratic term (LP), a structurally present but
    numerically zero row and column of A, rows loose on both sides, and crossed bounds (l > u: refused like OSQP's setup does)"""
    from optimal_control_problem_amd.models import _csc_from_dense_mask
    out = []

    def add(name, Pd, Ad, q, l, u, Pmask=None, Amask=None):
        n = Pd.shape[0]; m = Ad.shape[0]
        Pm = (Pd != 0) if Pmask is None else Pmask; Am = (Ad != 0) if Amask is None else Amask
        Pp, Pi = _csc_from_dense_mask(Pm)
        Ap, Ai = _csc_from_dense_mask(Am) if m else (np.zeros(n + 1, np.int32), np.zeros(0, np.int32))
        out.append(dict(name=name, n=n, m=m, Pp=Pp, Pi=Pi, Ap=Ap, Ai=Ai, P=np.atleast_2d(Pd.T[Pm.T]), A=np.atleast_2d(Ad.T[Am.T]) if m else np.zeros((1, 0)),
                        q=np.atleast_2d(np.asarray(q, float)), l=np.atleast_2d(np.asarray(l, float)) if m else np.zeros((1, 0)),
                        u=np.atleast_2d(np.asarray(u, float)) if m else np.zeros((1, 0))))

    add("one_variable", np.array([[2.0]]), np.array([[1.0]]), [1.0], [-1.0], [1.0])
    add("no_constraints", np.diag([2.0, 1.0]), np.zeros((0, 2)), [1.0, -1.0], [], [])
    add("lp", np.zeros((2, 2)), np.eye(2), [1.0, -1.0], [-1, -1], [1, 1])
    add("zero_row_and_column", np.eye(2), np.array([[1.0, 0.0], [0.0, 0.0]]), [1.0, -1.0], [-1, -1], [1, 1], Amask=np.ones((2, 2), bool))
    add("loose_rows", np.eye(2), np.eye(2), [1.0, -1.0], [-INF, -INF], [INF, INF])
    add("crossed_bounds", np.eye(2), np.eye(2), [1.0, -1.0], [1.0, -1.0], [-1.0, 1.0])
    return out


@pytest.mark.parametrize("prob", _edge_problems(), ids=lambda p: p["name"])
def test_interface_edge_cases(built, prob):
    from optimal_control_problem_amd.batch_qp import BatchQP
    from oracle import oracle as orc
    n, m = prob["n"], prob["m"]
    pat = orc.Pattern(n, m, prob["Pp"], prob["Pi"], prob["Ap"], prob["Ai"])
    ref = pat.solve(prob["P"], prob["q"], prob["A"], prob["l"], prob["u"], orc.default_settings())
    qp = BatchQP(n, m, 1, prob["Pp"], prob["Pi"], prob["Ap"], prob["Ai"])
    qp.update(prob["P"], prob["q"], prob["A"], prob["l"], prob["u"]); qp.solve(); got = qp.get(); qp.close()
    assert np.array_equal(got["status"], ref["status"]) and np.array_equal(got["iters"], ref["iters"])
    if prob["name"] == "crossed_bounds":
        assert got["status"][0] == 11 and got["iters"][0] == 0 and np.isnan(got["x"]).all() and np.isnan(got["y"]).all() and np.isnan(ref["x"]).all()
    else:
        assert got["status"][0] == 1
        assert np.abs(got["x"] - ref["x"]).max() <= 1e-6 * (1 + np.abs(ref["x"]).max())
        if m:
            assert np.abs(got["y"] - ref["y"]).max() <= 1e-6 * (1 + np.abs(ref["y"]).max())


def test_crossed_bounds_refuse_only_their_instance(built):
    """l > u on one row of some instances of a batch: those report MPCQP_UNSOLVED / NaN (OSQP refuses such data at setup), every other
    instance is solved exactly as without them; the same through the kept workspace (new vectors with crossed bounds, then valid ones)"""
    from optimal_control_problem_amd.batch_qp import BatchQP
    from oracle import oracle as orc
    mdl, ls, _ = models.make_workload("double_integrator", 24)
    l = ls.l.copy(); u = ls.u.copy()
    bad = np.array([3, 10, 23]); row = mdl.n + 5
    l[bad, row] = 1.0; u[bad, row] = -1.0
    pat = orc.Pattern(ls.n, ls.m, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    clean = pat.solve(ls.P, ls.q, ls.A, ls.l, ls.u, orc.default_settings())
    ref = pat.solve(ls.P, ls.q, ls.A, l, u, orc.default_settings())
    qp = BatchQP(ls.n, ls.m, ls.batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    qp.keep_workspace(True)
    qp.update(ls.P, ls.q, ls.A, l, u); qp.solve(); got = qp.get()
    ok = np.setdiff1d(np.arange(ls.batch), bad)
    assert (got["status"][bad] == 11).all() and (got["iters"][bad] == 0).all() and np.isnan(got["x"][bad]).all() and np.isnan(got["z"][bad]).all()
    assert np.array_equal(got["status"], ref["status"]) and np.array_equal(got["iters"], ref["iters"])
    assert np.array_equal(got["status"][ok], clean["status"][ok]) and np.abs(got["x"][ok] - clean["x"][ok]).max() <= 1e-6 * (1 + np.abs(clean["x"]).max())
    # kept workspace: crossed bounds arrive with new vectors on other instances, then everything is valid again
    st = orc.State(pat, ls.batch, orc.default_settings()); st.solve(ls.P, ls.q, ls.A, l, u)
    l2 = ls.l.copy(); u2 = ls.u.copy(); l2[[1, 3], row] = 2.0; u2[[1, 3], row] = -2.0
    r2 = st.solve_vectors(ls.q, l2, u2)
    qp.update_vectors(ls.q, l2, u2); qp.solve(); g2 = qp.get()
    assert np.array_equal(g2["status"], r2["status"]) and np.array_equal(g2["iters"], r2["iters"]) and (g2["status"][[1, 3]] == 11).all()
    good = np.setdiff1d(np.arange(ls.batch), [1, 3])
    assert np.abs(g2["x"][good] - r2["x"][good]).max() <= 1e-6 * (1 + np.abs(r2["x"][good]).max())
    r3 = st.solve_vectors(ls.q, ls.l, ls.u)
    qp.update_vectors(ls.q, ls.l, ls.u); qp.solve(); g3 = qp.get(); qp.close()
    assert np.array_equal(g3["status"], r3["status"]) and np.array_equal(g3["iters"], r3["iters"]) and (g3["status"] == 1).all()
    assert np.abs(g3["x"] - r3["x"]).max() <= 1e-6 * (1 + np.abs(r3["x"]).max())


def test_handle_lifecycle_releases_device_memory(built):
    """create / update / solve / destroy in a loop (QP handles and stage evaluators): free device memory returns to where it was"""
    import torch
    from optimal_control_problem_amd.batch_qp import BatchQP
    from optimal_control_problem_amd.stage_eval import StageEvaluator
    mdl, ls, _ = models.make_workload("quadrotor", 64, N=10)

    def cycle(k):
        for _ in range(k):
            qp = BatchQP(ls.n, ls.m, ls.batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
            qp.keep_workspace(True)
            qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); qp.get()
            qp.update_vectors(ls.q, ls.l, ls.u); qp.solve(); qp.get(); qp.close()
            ev = StageEvaluator(mdl); ev.close()

    cycle(3); torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    cycle(40); torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 << 20, "device memory shrank by %.1f MiB over 40 create/destroy cycles" % ((free0 - free1) / 2**20)


def test_prefix_parity_where_adaptive_rho_is_noise_driven(built):
    """tools/fuzz_gpu.py case 9265: the dual residual converges to rounding noise while the primal one stalls, so the
    adaptive-rho rule at iteration 200 divides by noise and GPU and oracle continue on different trajectories of the same algorithm.  What
    can be pinned is pinned: up to the last check before that update the two agree tightly, and whatever point the GPU calls solved is
    feasible to the tolerance it claims."""
    from optimal_control_problem_amd.batch_qp import BatchQP
    from oracle import oracle as orc
    ls = problems.sparse_batch(11, 35, 7, 9265, 1.0)
    pat = orc.Pattern(ls.n, ls.m, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    for mi in (100, 150, 199):
        ref = pat.solve(ls.P, ls.q, ls.A, ls.l, ls.u, orc.default_settings(max_iter=mi))
        qp = BatchQP(ls.n, ls.m, ls.batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai, max_iter=mi)
        qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); got = qp.get(); qp.close()
        assert np.array_equal(got["status"], ref["status"]) and np.array_equal(got["iters"], ref["iters"])
        fin = np.isfinite(ref["x"])
        assert np.array_equal(np.isfinite(got["x"]), fin) and np.abs(got["x"][fin] - ref["x"][fin]).max() <= 1e-9 * (1 + np.abs(ref["x"][fin]).max())
        assert np.abs(got["rho"] - ref["rho"]).max() <= 1e-5 * ref["rho"].max()
    qp = BatchQP(ls.n, ls.m, ls.batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); got = qp.get(); qp.close()
    for b in np.nonzero(got["status"] == 1)[0]:
        _, A = ls.dense(b)
        Ax = A @ got["x"][b]
        assert np.maximum(np.maximum(ls.l[b] - Ax, Ax - ls.u[b]), 0.0).max() <= 1e-3 + 1e-3 * np.abs(Ax).max()


def test_one_wave_kernel_beyond_eight_per_cu(built):
    """the 128-VGPR instance of the one-wave kernel (LDS footprint below 17.7 KiB: more than eight QPs per CU)"""
    from optimal_control_problem_amd.batch_qp import BatchQP
    mdl, ls, _ = models.make_workload("double_integrator", 4000, N=10)      # past the batch-aware rule's three resident rounds
    qp = BatchQP(ls.n, ls.m, ls.batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    info = qp.plan_info()
    assert info["variant"] == 1 and 160 * 1024 // info["lds_bytes"] > 8
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); got = qp.get(); qp.close()
    ref = problems.oracle_solve(ls)
    assert (got["status"] == ref["status"]).all() and (got["iters"] == ref["iters"]).all()
    for k in ("x", "y", "z"):
        _close(got, ref, k)


@pytest.mark.parametrize("name,N,B,wgs", [("cartpole", 20, 48, 3), ("double_integrator", 30, 64, 3), ("double_integrator", 20, 40, 4)])
def test_resident_kernel_at_three_and_four_workgroups_per_cu(built, monkeypatch, name, N, B, wgs):
    """the LDS-resident 4-wave kernel's 168- and 128-VGPR instances (LDS footprint <= 53 / 40 KiB: three / four workgroups per CU)"""
    from optimal_control_problem_amd.batch_qp import BatchQP
    monkeypatch.setenv("MPCQP_VARIANT", "res4")
    mdl, ls, _ = models.make_workload(name, B, N=N)
    qp = BatchQP(ls.n, ls.m, ls.batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    info = qp.plan_info()
    assert info["variant"] == 4 and (40 * 1024 < info["lds_bytes"] <= 53 * 1024 if wgs == 3 else info["lds_bytes"] <= 40 * 1024)
    qp.keep_workspace(True)
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); got = qp.get()
    ref = problems.oracle_solve(ls)
    assert (got["status"] == ref["status"]).all() and (got["iters"] == ref["iters"]).all()
    for k in ("x", "y", "z"):
        _close(got, ref, k)
    # and the kept-workspace entry of the same instances
    from oracle import oracle as orc
    pat = orc.Pattern(ls.n, ls.m, ls.Pp, ls.Pi, ls.Ap, ls.Ai); st = orc.State(pat, B, orc.default_settings())
    st.solve(ls.P, ls.q, ls.A, ls.l, ls.u)
    q2 = ls.q * 1.05
    r2 = st.solve_vectors(q2, ls.l, ls.u)
    qp.update_vectors(q2, ls.l, ls.u); qp.solve(); g2 = qp.get(); qp.close()
    assert (g2["status"] == r2["status"]).all() and (g2["iters"] == r2["iters"]).all()
    _close(g2, r2, "x")


# ---------------------------------------------------------------------------------------------- reduced form (opt-in)
@pytest.mark.parametrize("name,B,N,first_frame", [("double_integrator", 12, 20, False), ("double_integrator", 12, 20, True), ("quadrotor", 10, 20, False),
                                                  ("quadrotor", 6, 20, True), ("cartpole", 6, 30, True)])
def test_reduced_form(built, name, B, N, first_frame):
    """mpcqp_create_reduced: the parameter rows dp = 0 (reference SQPOptimizationSolver.cpp:117) -- and optionally the pinned first
    frame (OptimalControlProblem.cpp:93-96) -- named as fixed.  (a) against the oracle on the reduced QP (NumPy statement of the
    substitution): status, iteration counts, x, y at the tight bar; (b) against the full form: the same optimum within the
    termination tolerance; (c) the expanded (x, y) satisfy stationarity of the FULL QP as well as the reduced run's dual residual"""
    from optimal_control_problem_amd.batch_qp import BatchQP, solve_local_system
    mdl, ls, meta = models.make_workload(name, B, N=N)
    rows = list(range(mdl.np))
    if first_frame:
        rows += [mdl.np + j for j in range(mdl.nx + mdl.nu) if (ls.l[:, mdl.np + j] == ls.u[:, mdl.np + j]).all()]
        assert len(rows) > mdl.np
    qp = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai, fixed_rows=rows)
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); got = qp.get()
    red, free, kept, fvars, xfix = problems.reduce_qp(ls, rows)
    assert qp.plan_info()["n"] == red.n and qp.plan_info()["m"] == red.m
    ref = problems.oracle_solve(red)
    assert (got["status"] == ref["status"]).all() and (got["iters"] == ref["iters"]).all()
    tol = lambda a: 1e-6 * (1.0 + np.abs(a).max())
    assert np.abs(got["x"][:, free] - ref["x"]).max() <= tol(ref["x"]) and np.abs(got["y"][:, kept] - ref["y"]).max() <= tol(ref["y"])
    assert np.array_equal(got["x"][:, fvars], xfix) and np.array_equal(got["z"][:, rows], ls.l[:, rows])
    # (b) the same optimum as the full form: at eps = 1e-3 two ADMM runs on these problems stop far apart in x (both inside the residual
    # test), so the optimum is compared at a tight tolerance
    tight = dict(eps_abs=1e-9, eps_rel=1e-9)
    qt = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai, fixed_rows=rows, **tight)
    qt.update(ls.P, ls.q, ls.A, ls.l, ls.u); qt.solve(); gt = qt.get(); qt.close()
    full = solve_local_system(ls, **tight)
    assert (full["status"] == 1).all() and (gt["status"] == 1).all()
    assert np.abs(gt["x"] - full["x"]).max() <= 1e-4 * (1.0 + np.abs(full["x"]).max())
    assert np.abs(gt["y"] - full["y"]).max() <= 1e-4 * (1.0 + np.abs(full["y"]).max())
    # (c) the expanded point of the default-tolerance run passes OSQP's residual test on the FULL QP
    for b in range(B):
        Pd, Ad = ls.dense(b); Pd = np.triu(Pd) + np.triu(Pd, 1).T
        x, y, z = got["x"][b], got["y"][b], got["z"][b]
        stat = Pd @ x + ls.q[b] + Ad.T @ y
        assert np.abs(stat[fvars]).max() <= 1e-9 * (1.0 + max(np.abs(Pd @ x).max(), np.abs(ls.q[b]).max(), np.abs(Ad.T @ y).max()))   # eliminated variables: exact
        assert np.abs(stat).max() <= 2 * (1e-3 + 1e-3 * max(np.abs(Pd @ x).max(), np.abs(ls.q[b]).max(), np.abs(Ad.T @ y).max()))
        assert np.abs(Ad @ x - z).max() <= 2 * (1e-3 + 1e-3 * max(np.abs(Ad @ x).max(), np.abs(z).max()))
        assert (z >= np.maximum(ls.l[b], -1e30) - 1e-9).all() and (z <= np.minimum(ls.u[b], 1e30) + 1e-9).all()
    # kept workspace on the reduced handle
    qp.keep_workspace(True)
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); qp.get()
    q2 = ls.q * 1.1
    qp.update_vectors(q2, ls.l, ls.u); qp.solve(); b2 = qp.get()
    red2, *_ = problems.reduce_qp(models.LocalSystem(ls.n, ls.m, ls.Pp, ls.Pi, ls.Ap, ls.Ai, ls.P, q2, ls.A, ls.l, ls.u), rows)
    from oracle import oracle as orc
    st = orc.State(orc.Pattern(red.n, red.m, red.Pp, red.Pi, red.Ap, red.Ai), B, orc.default_settings())
    st.solve(red.P, red.q, red.A, red.l, red.u); r2 = st.solve_vectors(red2.q, red2.l, red2.u)
    assert (b2["iters"] == r2["iters"]).all() and np.abs(b2["x"][:, free] - r2["x"]).max() <= tol(r2["x"])
    qp.close()


@pytest.mark.parametrize("name,B,N,variant", [("quadrotor", 8, 20, 204), ("cartpole", 6, 30, None), ("quadrotor", 4, 50, 208), ("cartpole", 4, 100, None)])
def test_presolved_create_finds_the_rows(built, name, B, N, variant):
    """mpcqp_create_presolved: the rows a caller of the reference's formulation cannot name -- 0 <= dp <= 0 on the parameter block
    (SQPOptimizationSolver.cpp:117) and the first frame pinned through lbx = ubx (OptimalControlProblem.cpp:93-96) -- are found from the bounds of
    the first update.  (a) exactly the singleton rows with l = u in every instance; (b) the handle is the one mpcqp_create_reduced gives for those
    rows: bitwise the same results; (c) oracle parity on the reduced QP at the tight bar; (d) long chains stay on chip without an arrow head
    (variant 208); (e) the same optimum as the full form at a tight tolerance"""
    from optimal_control_problem_amd.batch_qp import BatchQP, solve_local_system
    mdl, ls, meta = models.make_workload(name, B, N=N)
    Pd, Ad = ls.dense(0)
    single = [i for i in range(ls.m) if (Ad[i] != 0).sum() == 1]
    rows = [i for i in single if (ls.l[:, i] == ls.u[:, i]).all()]
    assert len(rows) >= mdl.np + mdl.nx
    qp = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai, presolve_bounds=(ls.l, ls.u))
    assert qp.nfixed == len(rows)
    if variant:
        assert qp.plan_info()["variant"] == variant
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); got = qp.get(); qp.close()
    qn = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai, fixed_rows=rows)
    qn.update(ls.P, ls.q, ls.A, ls.l, ls.u); qn.solve(); named = qn.get(); qn.close()
    for k in ("x", "y", "z", "status", "iters"):
        assert np.array_equal(got[k], named[k], equal_nan=True), k
    red, free, kept, fvars, xfix = problems.reduce_qp(ls, rows)
    ref = problems.oracle_solve(red)
    assert (got["status"] == ref["status"]).all() and (got["iters"] == ref["iters"]).all()
    tol = lambda a: 1e-6 * (1.0 + np.abs(a).max())
    assert np.abs(got["x"][:, free] - ref["x"]).max() <= tol(ref["x"]) and np.abs(got["y"][:, kept] - ref["y"]).max() <= tol(ref["y"])
    assert np.array_equal(got["x"][:, fvars], xfix)
    tight = dict(eps_abs=1e-9, eps_rel=1e-9)
    qt = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai, presolve_bounds=(ls.l, ls.u), **tight)
    qt.update(ls.P, ls.q, ls.A, ls.l, ls.u); qt.solve(); gt = qt.get(); qt.close()
    full = solve_local_system(ls, **tight)
    both = (full["status"] == 1) & (gt["status"] == 1)      # (at 1e-9 the long cart-pole horizon stops at max_iter in one form or the other: compared where both arrive)
    if N <= 50:
        assert both.all()
    if both.any():
        assert np.abs(gt["x"][both] - full["x"][both]).max() <= 1e-4 * (1.0 + np.abs(full["x"][both]).max())


def test_reduced_form_refuses_what_it_must(built):
    from optimal_control_problem_amd import _lib
    from optimal_control_problem_amd.batch_qp import BatchQP
    mdl, ls, meta = models.make_workload("double_integrator", 4)
    with pytest.raises(_lib.MpcqpError) as e:
        BatchQP(ls.n, ls.m, 4, ls.Pp, ls.Pi, ls.Ap, ls.Ai, fixed_rows=[ls.m - 1])       # a dynamics row: more than one entry
    assert e.value.code == _lib.ERR_ARG
    with pytest.raises(_lib.MpcqpError):
        BatchQP(ls.n, ls.m, 4, ls.Pp, ls.Pi, ls.Ap, ls.Ai, fixed_rows=[0, 0])
    qp = BatchQP(ls.n, ls.m, 4, ls.Pp, ls.Pi, ls.Ap, ls.Ai, fixed_rows=list(range(mdl.np)))
    l = ls.l.copy(); l[2, 0] -= 0.5                                                   # instance 2 breaks the promise l = u on a named row
    qp.update(ls.P, ls.q, ls.A, l, ls.u); qp.solve(); got = qp.get()
    assert got["status"][2] == 11 and np.isnan(got["x"][2]).all() and (np.delete(got["status"], 2) == 1).all()
    qp.close()


# ---------------------------------------------------------------------------------------------- kernel family by measurement
@pytest.mark.parametrize("name,N,B", [("double_integrator", 20, 512), ("quadrotor", 10, 256), ("cartpole", 30, 256)])
def test_tuned_create_returns_a_family_handle(built, monkeypatch, name, N, B):
    """mpcqp_create_tuned (opt-in; MPCQP_AUTOTUNE=1 routes mpcqp_create to it): the handle that comes back is a handle of one family --
    results bitwise those of MPCQP_VARIANT=<that family> -- the choice is cached per pattern, and the default create is untouched"""
    from optimal_control_problem_amd.batch_qp import BatchQP
    names = {1: "res1", 2: "res2", 4: "res4", 102: "gres2", 104: "gres4", 204: "oc4", 208: "oc8"}
    mdl, ls, _ = models.make_workload(name, B, N=N)

    def run(**kw):
        qp = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai, **kw)
        v = qp.plan_info()["variant"]
        qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); got = qp.get(); qp.close()
        return v, got

    v_rule, g_rule = run()
    v_tuned, g_tuned = run(tuned=True)
    assert v_tuned in names
    monkeypatch.setenv("MPCQP_VARIANT", names[v_tuned])
    v_forced, g_forced = run()
    monkeypatch.delenv("MPCQP_VARIANT")
    assert v_forced == v_tuned
    for k in ("x", "y", "z", "status", "iters"):
        assert np.array_equal(g_tuned[k], g_forced[k], equal_nan=True), k
    monkeypatch.setenv("MPCQP_AUTOTUNE", "1")                     # the environment switch takes the same (cached) choice
    v_env, g_env = run()
    monkeypatch.delenv("MPCQP_AUTOTUNE")
    assert v_env == v_tuned and np.array_equal(g_env["x"], g_tuned["x"], equal_nan=True)
    assert run()[0] == v_rule                                     # default behaviour unchanged
    ref = problems.oracle_solve(ls)
    assert (g_tuned["status"] == ref["status"]).all() and (g_tuned["iters"] == ref["iters"]).all()
    _close(g_tuned, ref, "x")


def test_two_wave_global_block_kernel_on_a_long_horizon(built, monkeypatch):
    """MPCQP_VARIANT=gres2 forced on a horizon for which the four-wave global-block kernel moves z, y into the slab: the two-wave kernel
    has no such instance and must keep them in LDS (it once took the other layout and reported every QP non-convex)"""
    from optimal_control_problem_amd.batch_qp import BatchQP
    monkeypatch.setenv("MPCQP_VARIANT", "gres2")
    mdl, ls, _ = models.make_workload("quadrotor", 8, N=50)
    qp = BatchQP(ls.n, ls.m, ls.batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    assert qp.plan_info()["variant"] == 102
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); got = qp.get(); qp.close()
    ref = problems.oracle_solve(ls)
    assert (got["status"] == ref["status"]).all() and (got["iters"] == ref["iters"]).all()
    for k in ("x", "y", "z"):
        _close(got, ref, k)


@pytest.mark.parametrize("name,B,N,eps", [("quadrotor", 12, 20, 1e-3), ("quadrotor", 6, 50, 1e-3), ("quadrotor", 8, 20, 1e-7)])
def test_experimental_vector_tile_sweeps_vs_oracle(built, monkeypatch, name, B, N, eps):
    """MPCQP_VTILES=1 (opt-in, experimental, round 4): the two sweeps of the two-kernel form's iteration on ONE row-major copy of A's dense 16 x 16
    blocks, multiplied on the vector ALUs, + remainder ELL layouts -- same scaled numbers as the ELL arrays, another summation order: same bar as
    every family.  The tight-tolerance case runs past the adaptive-rho steps: instances leave the tile kernel for a new factor and come back."""
    from optimal_control_problem_amd.batch_qp import BatchQP
    monkeypatch.setenv("MPCQP_VTILES", "1")
    mdl, ls, _ = models.make_workload(name, B, N=N)
    qp = BatchQP(ls.n, ls.m, ls.batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai, eps_abs=eps, eps_rel=eps)
    info = qp.plan_info()
    assert info["variant"] in (204, 208) and info["tiles"] == N - 1          # one tile per dynamics stage
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); got = qp.get(); qp.close()
    ref = problems.oracle_solve(ls, eps_abs=eps, eps_rel=eps)
    assert (got["status"] == ref["status"]).all() and (got["iters"] == ref["iters"]).all()
    if eps < 1e-3:
        assert ref["iters"].max() > 100
    for k in ("x", "y", "z"):
        _close(got, ref, k)


@pytest.mark.parametrize("name,B,N", [("quadrotor", 12, 20), ("quadrotor", 6, 50), ("quadrotor", 5, 30)])
def test_experimental_tile_sweeps_vs_oracle(built, monkeypatch, name, B, N):
    """MPCQP_TILES=1 (opt-in, experimental): the iteration's two sweeps on dense 16 x 16 tiles of A through the 4-block MFMA + remainder ELL
    layouts (plan.hpp build_tile_plan) -- same numbers as the ELL arrays, another summation order: same bar as every family"""
    from optimal_control_problem_amd.batch_qp import BatchQP
    monkeypatch.setenv("MPCQP_TILES", "1")
    mdl, ls, _ = models.make_workload(name, B, N=N)
    qp = BatchQP(ls.n, ls.m, ls.batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    info = qp.plan_info()
    assert info["variant"] in (204, 208) and info["tiles"] == N - 1          # one tile per dynamics stage
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); got = qp.get(); qp.close()
    ref = problems.oracle_solve(ls)
    assert (got["status"] == ref["status"]).all() and (got["iters"] == ref["iters"]).all()
    for k in ("x", "y", "z"):
        _close(got, ref, k)
    monkeypatch.delenv("MPCQP_TILES")
    qp = BatchQP(ls.n, ls.m, ls.batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai); info = qp.plan_info(); qp.close()
    assert info["tiles"] == 0                                                # default: ELL sweeps


@pytest.mark.parametrize("name,B,N", [("quadrotor", 6, 30), ("cartpole", 6, 100)])
def test_onchip_long_chains_warm_start_rho_and_refused_instances(built, name, B, N):
    """the interface features on the eight-wave on-chip instances (the default for these sizes): warm start, per-instance starting rho, an
    instance with crossed bounds refused while the others solve, an instance made primal infeasible reporting its certificate, the dispatch
    hint changing no result"""
    from optimal_control_problem_amd.batch_qp import BatchQP
    from oracle import oracle as orc
    mdl, ls, _ = models.make_workload(name, B, N=N)
    pat = orc.Pattern(ls.n, ls.m, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    cold = pat.solve(ls.P, ls.q, ls.A, ls.l, ls.u, orc.default_settings())
    qp = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai, warm_start=1)
    assert qp.plan_info()["variant"] == 208
    # warm start + starting rho
    rho0 = cold["rho"].copy(); rho0[::2] = 0.0
    ref = pat.solve(ls.P, ls.q, ls.A, ls.l, ls.u, orc.default_settings(warm_start=1), x0=cold["x"], y0=cold["y"], rho0=rho0)
    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.warm_start(cold["x"], cold["y"]); qp.set_rho(rho0); qp.solve(); got = qp.get()
    assert (got["status"] == ref["status"]).all() and (got["iters"] == ref["iters"]).all()
    _close(got, ref, "x"); _close(got, ref, "y")
    # the same solve again, now dispatched by the hint of the first: bitwise the same
    qp.solve(); again = qp.get()
    assert np.array_equal(again["x"], got["x"]) and np.array_equal(again["iters"], got["iters"])
    qp.close()
    # instance 1: crossed bounds on one row (refused); instance 2: the second frame's first state boxed away from anything the pinned first
    # frame can reach in one step (primal infeasible)
    l2, u2 = ls.l.copy(), ls.u.copy()
    l2[1, mdl.np + 3] = 1.0; u2[1, mdl.np + 3] = -1.0
    row = mdl.np + mdl.f                      # the bound row of frame 1, state 0
    l2[2, row] = 50.0; u2[2, row] = 60.0
    ref2 = pat.solve(ls.P, ls.q, ls.A, l2, u2, orc.default_settings())
    qp = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    qp.update(ls.P, ls.q, ls.A, l2, u2); qp.solve(); got2 = qp.get(); qp.close()
    assert got2["status"][1] == 11 and got2["iters"][1] == 0 and np.isnan(got2["x"][1]).all()          # MPCQP_UNSOLVED: refused
    keep = np.arange(B) != 1
    assert (got2["status"][keep] == ref2["status"][keep]).all(), (got2["status"], ref2["status"])
    assert ref2["status"][2] in (3, 4)                                                                   # the oracle certifies infeasibility
    ok = keep & np.isfinite(ref2["x"]).all(axis=1)
    assert (got2["iters"][ok] == ref2["iters"][ok]).all()
    assert np.abs(got2["x"][ok] - ref2["x"][ok]).max() <= 1e-6 * (1 + np.abs(ref2["x"][ok]).max())
