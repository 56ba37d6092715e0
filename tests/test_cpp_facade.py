"""The C++ CuCaQP facade (optimal_control_problem_amd/cpp/CuCaQP.hpp) over the C ABI, as a compiled program."""
import os
import subprocess

import numpy as np
import pytest

EXE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "support", "cucaqp_cpp_test")


def test_cpp_facade_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; see the gpu-marked test")
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stdout, r.stderr)       # refused loudly, no CPU fallback
    assert "Invalid dimensions" in r.stderr and "not initialized" in r.stderr and "no usable gfx950 GPU" in r.stderr


@pytest.mark.gpu
def test_cpp_facade_on_gpu(built):
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "status 1 iters 25" in r.stdout


CASADI_EXE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "support", "cucaqp_casadi_test")


def test_cpp_facade_casadi_overloads_compile_and_refuse_without_gpu(built):
    """the CasADi overloads of cpp/CuCaQP.hpp (setSystem(DMVector), getSolutionAsDM) compiled against tests/support/casadi_mock
    -- CasADi itself is not installed here -- so the one literal drop-in path (reference SQPOptimizationSolver.cpp unchanged over
    this CuCaQP) is not dead code: exit code 2 would mean the overloads were not compiled at all"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; see the gpu-marked test")
    r = subprocess.run([CASADI_EXE], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
def test_cpp_facade_casadi_call_sequence_on_gpu(built):
    """reference call sequence setDimension -> settings -> [setSystem(DMVector{P,q,A,l,u}) -> initSolver -> solve -> getSolutionAsDM] x 2"""
    r = subprocess.run([CASADI_EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "iteration 1" in r.stdout


EIGEN_EXE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "support", "cucaqp_eigen_test")


def test_cpp_facade_eigen_overloads_compile_and_refuse_without_gpu(built):
    """the Eigen overloads of cpp/CuCaQP.hpp (reference CuCaQP.h:37,49,51,53,55 and the Eigen-returning getSolution :76) compiled against
    tests/support/eigen_mock: exit code 2 would mean they were not compiled at all"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; see the gpu-marked test")
    r = subprocess.run([EIGEN_EXE], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
def test_cpp_facade_eigen_overloads_on_gpu(built):
    """each Eigen overload called on the GPU (float = the reference's OSQPFloat, and double), then refused with a wrong dimension"""
    r = subprocess.run([EIGEN_EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert r.stdout.count("Eigen overloads") == 2 and "mismatch" in r.stderr


SQP_EXE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "support", "stagesqp_cpp_test")


def test_cpp_stage_sqp_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; see the gpu-marked test")
    r = subprocess.run([SQP_EXE], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stdout, r.stderr)       # refused loudly, no CPU fallback


@pytest.mark.gpu
def test_cpp_stage_sqp_on_gpu(built):
    """StageSQP.hpp: the device-resident SQP loop written against the C ABI in C++"""
    r = subprocess.run([SQP_EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "max dynamics violation" in r.stdout


OCP_EXE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "support", "ocp_cpp_test")


def test_cpp_ocp_config_and_errors(built):
    """OptimalControlProblem.hpp without a GPU: YAML subset reader, OCPConfig semantics, validateConfig, error behaviour"""
    r = subprocess.run([OCP_EXE, "config"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "config ok" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_cpp_ocp_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; see the gpu-marked test")
    r = subprocess.run([OCP_EXE, "run"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
def test_cpp_ocp_on_gpu(built):
    """a C++ subclass written like the reference's examples (deployConstraintsAndAddCost / genSolver / computeOptimalTrajectory)"""
    r = subprocess.run([OCP_EXE, "run"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "run ok" in r.stdout, (r.returncode, r.stdout, r.stderr)


def _di_step(s, u):
    return np.stack([s[..., 0] + 0.05 * s[..., 1] + 0.00125 * u[..., 0], s[..., 1] + 0.05 * u[..., 0]], axis=-1)


def _di_stage_cost(s, u, r):
    e = s - r
    return 10.0 * e[..., 0] ** 2 + e[..., 1] ** 2 + e[..., 0] * e[..., 1] + (0.1 + 0.05 * e[..., 1] ** 2) * u[..., 0] ** 2


def _di_terminal_cost(s, u, r):
    e = s - r
    return 200.0 * e[..., 0] ** 2 + 20.0 * e[..., 1] ** 2 + 0.1 * u[..., 0] ** 2


@pytest.mark.gpu
def test_cpp_ocp_general_cost_equals_python_facade(built):
    """C++ OptimalControlProblem with StageCost terms carried by a generated library (dynamics + stage cost + terminal cost) against
    the Python facade on the same problem, both device-resident"""
    import yaml
    from optimal_control_problem_amd import codegen
    from optimal_control_problem_amd.ocp import Dynamics, OptimalControlProblem, StageCost
    lib = codegen.build_device_library(codegen.trace(_di_step, 2, 1, lcost=_di_stage_cost, lterm=_di_terminal_cost))
    r = subprocess.run([OCP_EXE, "cost", lib], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "cost ok" in r.stdout, (r.returncode, r.stdout, r.stderr)
    got = np.array([[float(v) for v in line.split()[1:]] for line in r.stdout.splitlines() if line.startswith("traj")])
    text = """
      discretization_settings: {dt: 0.05, horizon: 20}
      solver_settings: {verbose: false, gen_code: true, load_lib: false, max_iter: 1000, warm_start: true, solve_method: CUDA_SQP,
                        SQP_settings: {alpha: 1.0, step_num: 2}}
      OCP_variables:
        - {name: state, size: 2, lower_bound: [-.inf, -2.0], upper_bound: [.inf, 2.0]}
        - {name: input, size: 1, lower_bound: [-1.0], upper_bound: [1.0]}
    """

    class DI(OptimalControlProblem):
        def deployConstraintsAndAddCost(self):
            cfg = self.OCPConfigPtr_; ref = self.setReference(2); N = cfg.getHorizon()
            for k in range(N):
                st, inp = cfg.getVariable(k, "state"), cfg.getVariable(k, "input")
                self.addScalarCost(StageCost(_di_stage_cost if k < N - 1 else _di_terminal_cost, st, inp, ref))
                if k < N - 1:
                    self.addEquationConstraint("dynamics", cfg.getVariable(k + 1, "state"), Dynamics(_di_step, st, inp))

    B = 4
    ocp = DI(yaml.safe_load(text), batch=B); ocp.deployConstraintsAndAddCost(); ocp.genSolver()
    frame = np.array([[-0.8 + 0.5 * b, 0.2, 0.0] for b in range(B)]); ref = np.array([[0.1 * b, 0.0] for b in range(B)])
    want = ocp.computeOptimalTrajectory(frame, ref)
    assert got.shape == want.shape and np.abs(got - want).max() <= 1e-9 * (1 + np.abs(want).max())


def _di_rate(s, u, sn, un):
    return np.stack([un[..., 0] - u[..., 0]], axis=-1)


@pytest.mark.gpu
def test_cpp_ocp_link_constraint_equals_python_facade(built):
    """C++ OptimalControlProblem::addInequalityConstraint with a Link (rate limit between consecutive frames, carried by the generated
    library) against the Python facade on the same problem, both device-resident: the form of SX expressions over several frames
    (reference src/OptimalControlProblem.cpp:448-489) this facade takes"""
    import yaml
    from optimal_control_problem_amd import codegen
    from optimal_control_problem_amd.ocp import Dynamics, Link, OptimalControlProblem
    lib = codegen.build_device_library(codegen.trace(_di_step, 2, 1, kfun=_di_rate, nk=1, k_lo=[-0.15], k_hi=[0.15]))
    r = subprocess.run([OCP_EXE, "link", lib], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "link ok" in r.stdout, (r.returncode, r.stdout, r.stderr)
    got = np.array([[float(v) for v in line.split()[1:]] for line in r.stdout.splitlines() if line.startswith("traj")])
    text = """
      discretization_settings: {dt: 0.05, horizon: 20}
      solver_settings: {verbose: false, gen_code: true, load_lib: false, max_iter: 1000, warm_start: true, solve_method: CUDA_SQP,
                        SQP_settings: {alpha: 1.0, step_num: 2}}
      OCP_variables:
        - {name: state, size: 2, lower_bound: [-.inf, -2.0], upper_bound: [.inf, 2.0]}
        - {name: input, size: 1, lower_bound: [-1.0], upper_bound: [1.0]}
    """

    class DI(OptimalControlProblem):
        def deployConstraintsAndAddCost(self):
            cfg = self.OCPConfigPtr_; ref = self.setReference(2); N = cfg.getHorizon()
            for k in range(N):
                st, inp = cfg.getVariable(k, "state"), cfg.getVariable(k, "input")
                self.addVectorCost([10.0, 1.0], st - ref); self.addVectorCost([0.1], inp)
                if k < N - 1:
                    self.addEquationConstraint("dynamics", cfg.getVariable(k + 1, "state"), Dynamics(_di_step, st, inp))
                    self.addInequalityConstraint("rate", [-0.15], Link(_di_rate, st, inp, cfg.getVariable(k + 1, "state"), cfg.getVariable(k + 1, "input"), 1), [0.15])

    B = 6
    ocp = DI(yaml.safe_load(text), batch=B); ocp.deployConstraintsAndAddCost(); ocp.genSolver()
    frame = np.array([[-1.5 + 0.6 * b, 0.4 - 0.15 * b, 0.0] for b in range(B)]); ref = np.zeros((B, 2))
    want = ocp.computeOptimalTrajectory(frame, ref)
    assert got.shape == want.shape and np.abs(got - want).max() <= 1e-9 * (1 + np.abs(want).max())
