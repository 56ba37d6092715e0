"""CuCaQP (Python mirror) update* bookkeeping without a GPU: which ABI call the next solve() makes.

The reference's update* members are private and never called (reference src/sqp_solver/CuCaQP.cpp:106-161); here they work and
must behave like the C++ facade (optimal_control_problem_amd/cpp/CuCaQP.hpp): new data always reaches the device before the next
solve -- vectors alone through the kept workspace once a solve has happened on it, a full update otherwise."""
import numpy as np
import pytest

from optimal_control_problem_amd import cucaqp


class FakeBatchQP:
    """records the calls CuCaQP makes; stands in for batch_qp.BatchQP (the C ABI needs a GPU)"""
    log = []
    refuse_keep = False

    def __init__(self, n, m, batch, Pp, Pi, Ap, Ai, **kw):
        self.n, self.m, self.batch = n, m, batch
        FakeBatchQP.log.append(("create",))

    def keep_workspace(self, on):
        if FakeBatchQP.refuse_keep:
            raise cucaqp._lib.MpcqpError(cucaqp._lib.ERR_LIMIT, "streaming variant")
        FakeBatchQP.log.append(("keep", on))

    def update(self, P, q, A, l, u):
        FakeBatchQP.log.append(("update", np.array(P, copy=True), np.array(q, copy=True)))

    def update_vectors(self, q, l, u):
        FakeBatchQP.log.append(("update_vectors", np.array(q, copy=True)))

    def warm_start(self, x0, y0): pass
    def set_rho(self, r): pass
    def solve(self): FakeBatchQP.log.append(("solve",))
    def get(self): return {"x": np.zeros((self.batch, self.n)), "y": np.zeros((self.batch, self.m)), "status": np.ones(self.batch, int), "iters": np.ones(self.batch, int)}
    def close(self): pass


@pytest.fixture
def qp(monkeypatch):
    FakeBatchQP.log = []; FakeBatchQP.refuse_keep = False
    monkeypatch.setattr(cucaqp, "BatchQP", FakeBatchQP)
    q = cucaqp.CuCaQP()
    q.setDimension(2, 3)
    P = (np.array([0, 1, 2]), np.array([0, 1]), np.array([2.0, 2.0]))
    A = (np.array([0, 2, 4]), np.array([0, 2, 1, 2]), np.ones(4))
    q.setSystem([P, np.zeros(2), A, np.array([-50.0, -100.0, 1.0]), np.array([50.0, 100.0, 1.0])])
    return q, P, A


def kinds():
    return [e[0] for e in FakeBatchQP.log]


def test_update_gradient_between_init_and_first_solve_is_not_lost(qp):
    q, P, A = qp
    assert q.initSolver()
    assert q.updateGradient(np.array([-2.0, 0.0]))
    assert q.solve()
    # no solve has happened on the workspace yet: the new q goes in through a full update, before the solve
    assert kinds() == ["create", "keep", "update", "update", "solve"]
    assert np.array_equal(FakeBatchQP.log[3][2], [[-2.0, 0.0]])


def test_vectors_after_a_solve_use_the_kept_workspace(qp):
    q, P, A = qp
    assert q.initSolver() and q.solve()
    assert q.updateGradient(np.array([-2.0, 0.0])) and q.updateLowerBound(np.array([-50.0, -100.0, 2.0])) and q.updateUpperBound(np.array([50.0, 100.0, 2.0]))
    assert q.solve()
    assert kinds() == ["create", "keep", "update", "solve", "update_vectors", "solve"]
    assert q.solve()                                           # nothing changed: no further update
    assert kinds()[-2:] == ["solve", "solve"]


def test_matrix_update_forces_a_full_update(qp):
    q, P, A = qp
    assert q.initSolver() and q.solve()
    assert q.updateHessianMatrix((P[0], P[1], np.array([8.0, 4.0]))) and q.updateGradient(np.array([-2.0, 0.0]))
    assert q.solve()
    assert kinds() == ["create", "keep", "update", "solve", "update", "solve"]
    assert np.array_equal(FakeBatchQP.log[4][1], [8.0, 4.0]) and np.array_equal(FakeBatchQP.log[4][2], [[-2.0, 0.0]])
    assert q.updateLinearConstraintsMatrix((A[0], A[1], 2 * np.ones(4))) and q.solve()
    assert kinds()[-2:] == ["update", "solve"]


def test_without_a_kept_workspace_vectors_go_through_a_full_update(qp):
    q, P, A = qp
    FakeBatchQP.refuse_keep = True                             # what the streaming kernel variant answers
    assert q.initSolver() and q.solve()
    assert q.updateGradient(np.array([-2.0, 0.0])) and q.solve()
    assert kinds() == ["create", "update", "solve", "update", "solve"]


def test_changed_pattern_is_refused_until_init(qp, capsys):
    q, P, A = qp
    assert q.initSolver() and q.solve()
    dense = (np.array([0, 2, 4]), np.array([0, 1, 0, 1]), np.array([2.0, 0.5, 0.5, 2.0]))
    assert q.updateHessianMatrix(dense) is False
    assert "sparsity pattern changed" in capsys.readouterr().err
    assert q.solve() is False                                  # "Solver not initialized"
    assert q.initSolver() and q.solve()                        # a new plan for the new pattern
    assert kinds()[-4:] == ["create", "keep", "update", "solve"]


def test_updates_before_init_are_refused(qp):
    q, P, A = qp
    assert q.updateGradient(np.zeros(2)) is False and q.updateHessianMatrix(P) is False
