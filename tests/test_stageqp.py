"""Structured stage form, host side (include/mpcqp.h "Structured stage form", csrc/stageqp.hip): the CSC pattern it stands for.  No GPU."""
import ctypes as C

import numpy as np
import pytest

from optimal_control_problem_amd import _lib, models, stage_qp
from tests.support import stage_blocks as sb


@pytest.mark.parametrize("name,N", [("quadrotor", 20), ("quadrotor", 50), ("cartpole", 100), ("double_integrator", 20)])
def test_stage_pattern_is_the_workloads_pattern(built, name, N):
    """with the masks of the reference's formulation (diagonal tracking cost coupled to p, the Jacobian's structure) the stage form's
    pattern is entry for entry the CSC pattern of the BASELINE workloads -- the arrays mpcqp_create gets from the generic path"""
    mdl, ls, _ = models.make_workload(name, 2, N=N)
    cm, dm = sb.masks_of(ls, N, mdl.nx, mdl.nu, mdl.nx)
    n, m, Pp, Pi, Ap, Ai = stage_qp.stage_pattern(N, mdl.nx, mdl.nu, mdl.nx, cm, dm)
    assert (n, m) == (ls.n, ls.m)
    for got, want in ((Pp, ls.Pp), (Pi, ls.Pi), (Ap, ls.Ap), (Ai, ls.Ai)):
        assert np.array_equal(got, want)


@pytest.mark.parametrize("seed", range(6))
def test_stage_pattern_random_masks_vs_numpy(built, seed):
    rng = np.random.default_rng(seed)
    N = int(rng.integers(2, 9)); nx = int(rng.integers(1, 7)); nu = int(rng.integers(0, 4)); npar = int(rng.choice([0, 0, nx, 3]))
    f = nx + nu; nl = f + npar
    cm = rng.random((nl, nl)) < 0.4; cm = cm | cm.T | np.eye(nl, dtype=bool)
    dm = rng.random((nx, f)) < 0.5
    for masks in ((cm, dm), (None, None)):
        n, m, Pp, Pi, Ap, Ai = stage_qp.stage_pattern(N, nx, nu, npar, *masks)
        n2, m2, Pp2, Pi2, Ap2, Ai2, _, _ = sb.numpy_pattern(N, nx, nu, npar, *masks)
        assert (n, m) == (n2, m2)
        for got, want in ((Pp, Pp2), (Pi, Pi2), (Ap, Ap2), (Ai, Ai2)):
            assert np.array_equal(got, want)


def test_stage_pattern_refuses_bad_dimensions_and_masks(built):
    with pytest.raises(_lib.MpcqpError) as e:
        stage_qp.stage_pattern(1, 2, 1)
    assert e.value.code == _lib.ERR_ARG
    with pytest.raises(_lib.MpcqpError):
        stage_qp.stage_pattern(4, 0, 1)
    cm = np.eye(3, dtype=bool); cm[0, 1] = True                     # not symmetric
    with pytest.raises(_lib.MpcqpError) as e:
        stage_qp.stage_pattern(4, 2, 1, 0, cm, None)
    assert "symmetric" in str(e.value)
    cm = np.ones((3, 3), bool); cm[2, 2] = False                    # a structurally zero diagonal entry
    with pytest.raises(_lib.MpcqpError) as e:
        stage_qp.stage_pattern(4, 2, 1, 0, cm, None)
    assert "diagonal" in str(e.value)
    with pytest.raises(ValueError):
        stage_qp.stage_pattern(4, 2, 1, 0, np.ones((4, 4), bool), None)      # mask of the wrong size
    L = _lib.lib()
    assert L.mpcqp_stageqp_pattern(None, None, None, None, None, None) == _lib.ERR_ARG
    sq = C.c_void_p()
    d = stage_qp.StageDims(4, 2, 1, 0, None, None)
    assert L.mpcqp_stageqp_create(C.byref(d), 0, None, C.byref(sq)) == _lib.ERR_ARG and not sq.value


def test_blocks_from_dense_round_trip(built):
    """blocks cut out of dense matrices, scattered back through the pattern, give the matrices again"""
    H, AB, q, l, u, Pd, Ad = sb.random_ltv(5, 3, 2, 2, 0)
    H2, Hp2, Hpp2, AB2 = stage_qp.blocks_from_dense(Pd, Ad, 5, 3, 2, 0)
    assert np.array_equal(H, H2) and np.array_equal(AB, AB2) and Hp2.shape == (2, 5, 0, 5) and Hpp2.shape == (2, 0, 0)


C_EXE = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "support", "stageqp_c_test")


def test_stage_form_from_plain_c_without_gpu(built):
    """include/mpcqp.h compiles as C99 (gcc -std=c99 -Wall in __graft_entry__.build), the host-only calls work from C, and without a GPU
    the create call refuses loudly"""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; see the gpu-marked test")
    r = subprocess.run([C_EXE], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stdout, r.stderr)
    assert "pattern ok" in r.stdout and "no usable gfx950 GPU" in r.stderr


@pytest.mark.gpu
def test_stage_form_from_plain_c_on_gpu(built):
    import subprocess
    r = subprocess.run([C_EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "5 instances solved" in r.stdout
