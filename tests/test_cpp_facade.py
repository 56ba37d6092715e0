"""The C++ CuCaQP facade (optimal_control_problem_amd/cpp/CuCaQP.hpp) over the C ABI, as a compiled program."""
import os
import subprocess

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
