"""CPU: the C-ABI library loads, exports every symbol include/mpcqp.h declares, and fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from optimal_control_problem_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header(built):
    hdr = open(os.path.join(ROOT, "include", "mpcqp.h")).read()
    declared = sorted(set(re.findall(r"\b(mpcqp_[a-z_0-9]+)\s*\(", hdr)))
    assert declared == sorted(_lib.EXPORTS)
    L = C.CDLL(_lib.SO_PATH)
    for name in declared:
        assert hasattr(L, name), name


def test_default_settings_are_the_reference_values(built):
    s = _lib.default_settings()
    # reference src/sqp_solver/SQPOptimizationSolver.cpp:83-85 + OSQP defaults
    assert (s.eps_abs, s.eps_rel, s.max_iter) == (1e-3, 1e-3, 10000)
    assert (s.rho, s.sigma, s.alpha, s.check_termination, s.scaling, s.adaptive_rho) == (0.1, 1e-6, 1.6, 25, 10, 1)
    assert s.warm_start == 0


def test_settings_struct_layout_matches_oracle(built):
    from oracle import oracle as orc
    a = [f[0] for f in _lib.Settings._fields_][:-1]
    b = [f[0] for f in orc.Settings._fields_][:-1]
    assert a == b                                     # same field order up to the last (device vs linsys)


def test_strerror(built):
    assert _lib.lib().mpcqp_strerror(0) == b"ok"
    assert b"argument" in _lib.lib().mpcqp_strerror(1)


def test_no_gpu_fails_loudly(built):
    """No CPU fallback: without a device every compute entry point reports MPCQP_ERR_NO_GPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from optimal_control_problem_amd.batch_qp import BatchQP
    with pytest.raises(_lib.MpcqpError) as e:
        BatchQP(2, 2, 1, np.array([0, 1, 2], np.int32), np.array([0, 1], np.int32), np.array([0, 1, 2], np.int32), np.array([0, 1], np.int32))
    assert e.value.code == _lib.ERR_NO_GPU
    from optimal_control_problem_amd.cucaqp import CuCaQP
    qp = CuCaQP()
    assert qp.setDimension(0, 1) is False             # reference CuCaQP.cpp:24-27
    assert qp.solve() is False                        # reference CuCaQP.cpp:200-203


def test_product_does_not_import_oracle():
    """the product path may not route through the oracle (or any CPU solver)"""
    pk = os.path.join(ROOT, "optimal_control_problem_amd")
    for dirpath, _, files in os.walk(pk):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("oracle/osqp_oracle", "").replace("oracle update_info", "").replace("oracle ", "").replace("(oracle", "").lower() or f.endswith(".hip"), f
                assert "import oracle" not in txt and "from oracle" not in txt and "liborc" not in txt, f
