"""CuCaQP -- host-side mirror of the reference's QP adapter class, batched.

Same member names, call order, argument meaning and error behaviour as the reference class
(reference include/optimal_control_problem/sqp_solver/CuCaQP.h:27-102, src/sqp_solver/CuCaQP.cpp): setters return
bool and print "Error: ..." to stderr instead of raising; setSystem takes [P, q, A, l, u] in that order
(CuCaQP.cpp:283-287) and clears the previous solver state (CuCaQP.cpp:271-280); initSolver + solve run the QP.
Differences, all additive: one object holds a batch of QPs of one sparsity (leading axis of the value
arrays; the reference's batch is 1), values stay fp64 (the reference's OSQP build is float,
cpu_install.sh:44), and duals/status/iterations are retrievable.

A sparse matrix is passed as a CSC triple (colptr, rowidx, values); values is [nnz] or [batch, nnz].
"""
import sys

import numpy as np

from . import _lib
from .batch_qp import BatchQP


def _err(msg):
    print("Error: " + msg, file=sys.stderr)
    return False


class CuCaQP:
    def __init__(self, batch=1, device=-1):
        self.batch = int(batch)
        self._device = device
        self.numOfVariables_ = 0
        self.numOfConstraints_ = 0
        self.isInitialized_ = False
        self._kw = {}
        self._P = self._A = None            # (colptr, rowidx, values)
        self.gradient = self.lowerBound = self.upperBound = None
        self._qp = None
        self._pattern_key = None
        self._result = None
        self._start = None
        self._rho0 = None
        self._kept = False
        self._vectors_dirty = self._matrices_dirty = self._solved_once = False

    # -- dimensions (CuCaQP.cpp:23-41)
    def setDimension(self, numOfVariables, numOfConstraints):
        if numOfVariables <= 0 or numOfConstraints <= 0:
            return _err("Invalid dimensions.")
        self._clear_solver()
        self.numOfVariables_ = int(numOfVariables)
        self.numOfConstraints_ = int(numOfConstraints)
        return True

    # -- settings pass-through (CuCaQP.cpp:163-181)
    def setVerbosity(self, verbosity):
        self._kw["verbose"] = bool(verbosity)

    def setWarmStart(self, warmStart):
        # kept for API parity; like the reference the flag is moot because setSystem clears the solver
        self._kw["warm_start_flag"] = bool(warmStart)

    def setAbsoluteTolerance(self, tolerance):
        self._kw["eps_abs"] = float(tolerance)

    def setRelativeTolerance(self, tolerance):
        self._kw["eps_rel"] = float(tolerance)

    def setMaxIteration(self, maxIteration):
        self._kw["max_iter"] = int(maxIteration)

    def setPrimalDualStart(self, x0, y0):
        """extension (the reference's private update* fast path made usable, CuCaQP.cpp:106-161): start the next solve's
        ADMM from (x0, y0) instead of zero; pass None to return to cold starts"""
        if x0 is None:
            self._start = None
            self._kw.pop("warm_start", None)
            return
        self._start = (np.ascontiguousarray(x0, dtype=np.float64).reshape(self.batch, -1),
                       np.ascontiguousarray(y0, dtype=np.float64).reshape(self.batch, -1))
        self._kw["warm_start"] = 1

    def setRhoStart(self, rho0):
        """extension: per-instance starting rho for the next solves (a kept OSQP workspace carries its adapted rho over);
        None returns to the configured rho"""
        self._rho0 = None if rho0 is None else np.ascontiguousarray(np.broadcast_to(np.asarray(rho0, np.float64), (self.batch,)))

    def setSolverSetting(self, **kw):
        """extension: any field of mpcqp_settings (rho, sigma, alpha, scaling, adaptive_rho, ...)"""
        self._kw.update(kw)

    # -- data (CuCaQP.cpp:43-103)
    def _vec(self, v, size, what):
        v = np.asarray(v, dtype=np.float64)
        if v.ndim == 1:
            v = np.broadcast_to(v, (self.batch, v.shape[0]))
        if v.shape != (self.batch, size):
            _err("%s vector size mismatch. Expected %d" % (what, size))
            return None
        return np.ascontiguousarray(v)

    def _mat(self, csc, rows, cols, what):
        if hasattr(csc, "tocsc"):          # scipy.sparse
            c = csc.tocsc(); c.sort_indices()
            if c.shape != (rows, cols):
                _err("%s matrix dimensions mismatch. Expected %dx%d" % (what, rows, cols))
                return None
            return (np.asarray(c.indptr, np.int32), np.asarray(c.indices, np.int32), np.asarray(c.data, np.float64))
        colptr, rowidx, values = csc
        colptr = np.asarray(colptr, np.int32); rowidx = np.asarray(rowidx, np.int32); values = np.asarray(values, np.float64)
        if len(colptr) != cols + 1 or (len(rowidx) and (rowidx.min() < 0 or rowidx.max() >= rows)) or values.shape[-1] != len(rowidx):
            _err("%s matrix dimensions mismatch. Expected %dx%d" % (what, rows, cols))
            return None
        return (colptr, rowidx, values)

    def setHessianMatrix(self, hessian):
        m = self._mat(hessian, self.numOfVariables_, self.numOfVariables_, "Hessian")
        if m is None:
            return False
        self._P = m
        return True

    def setGradient(self, q):
        v = self._vec(q, self.numOfVariables_, "Gradient")
        if v is None:
            return False
        self.gradient = v
        return True

    def setLinearConstraintsMatrix(self, A):
        m = self._mat(A, self.numOfConstraints_, self.numOfVariables_, "Constraint")
        if m is None:
            return False
        self._A = m
        return True

    def setLowerBound(self, l):
        v = self._vec(l, self.numOfConstraints_, "Lower bound")
        if v is None:
            return False
        self.lowerBound = v
        return True

    def setUpperBound(self, u):
        v = self._vec(u, self.numOfConstraints_, "Upper bound")
        if v is None:
            return False
        self.upperBound = v
        return True

    # -- the reference's private update* members (CuCaQP.cpp:106-161; never called there).  Here they work, with the same
    # bookkeeping as the C++ facade (cpp/CuCaQP.hpp: vectorsOnly_ / matricesDirty_ / solvedOnce_): the next solve() sends the new
    # data -- vectors alone through the kept workspace (mpcqp_update_vectors: scaling, factorisation and rho stay) when a solve
    # has happened on it and the matrices are unchanged, everything through a full mpcqp_update otherwise.  A matrix with another
    # sparsity pattern needs a new plan: the updater refuses it and asks for initSolver().
    def _same_pattern(self, old, new):
        return old is not None and np.array_equal(old[0], new[0]) and np.array_equal(old[1], new[1])

    def _update_mat(self, setter, attr, value, what):
        if not self.isInitialized_:
            return _err("Solver not initialized. Call initSolver() first.")          # CuCaQP.cpp:107-110 and siblings
        old = getattr(self, attr)
        if not setter(value):
            return False
        if not self._same_pattern(old, getattr(self, attr)):
            self.isInitialized_ = False
            return _err("%s sparsity pattern changed. Call initSolver() again." % what)
        self._matrices_dirty = True
        return True

    def updateHessianMatrix(self, hessian):
        return self._update_mat(self.setHessianMatrix, "_P", hessian, "Hessian")

    def updateLinearConstraintsMatrix(self, A):
        return self._update_mat(self.setLinearConstraintsMatrix, "_A", A, "Constraint")

    def _update_vec(self, setter, v):
        if not self.isInitialized_:
            return _err("Solver not initialized. Call initSolver() first.")          # CuCaQP.cpp:118-121 and siblings
        if not setter(v):
            return False
        self._vectors_dirty = True
        return True

    def updateGradient(self, q):
        return self._update_vec(self.setGradient, q)

    def updateLowerBound(self, l):
        return self._update_vec(self.setLowerBound, l)

    def updateUpperBound(self, u):
        return self._update_vec(self.setUpperBound, u)

    def setSystem(self, localSystem):
        """[P, q, A, l, u] (CuCaQP.cpp:271-288) or a models.LocalSystem; return values are dropped like the reference"""
        self.isInitialized_ = False
        self._result = None
        self._vectors_dirty = self._matrices_dirty = False
        if hasattr(localSystem, "Pp"):
            ls = localSystem
            localSystem = [(ls.Pp, ls.Pi, ls.P), ls.q, (ls.Ap, ls.Ai, ls.A), ls.l, ls.u]
        self.setHessianMatrix(localSystem[0])
        self.setGradient(localSystem[1])
        self.setLinearConstraintsMatrix(localSystem[2])
        self.setLowerBound(localSystem[3])
        self.setUpperBound(localSystem[4])

    # -- solve (CuCaQP.cpp:183-211)
    def _clear_solver(self):
        self.isInitialized_ = False
        self._result = None

    def initSolver(self):
        self._clear_solver()
        if self._P is None or self._A is None or self.gradient is None or self.lowerBound is None or self.upperBound is None:
            return _err("Failed to initialize solver.")
        lo = np.asarray(self.lowerBound, float).reshape(-1, self.numOfConstraints_); up = np.asarray(self.upperBound, float).reshape(-1, self.numOfConstraints_)
        if self.numOfConstraints_ and (lo > up).any(axis=1).all():
            # osqp_setup refuses l_i > u_i, so OsqpEigen's initSolver fails (reference CuCaQP.cpp:183-197); with a batch, instances
            # are refused one by one (status MPCQP_UNSOLVED) and only a batch without any valid instance fails here
            return _err("Failed to initialize solver. (lower bound greater than upper bound)")
        key = (self.numOfVariables_, self.numOfConstraints_, self.batch, self._P[0].tobytes(), self._P[1].tobytes(),
               self._A[0].tobytes(), self._A[1].tobytes(), tuple(sorted((k, v) for k, v in self._kw.items())))
        try:
            if self._qp is None or key != self._pattern_key:
                if self._qp is not None:
                    self._qp.close()
                kw = {k: v for k, v in self._kw.items() if k not in ("verbose", "warm_start_flag")}
                self._qp = BatchQP(self.numOfVariables_, self.numOfConstraints_, self.batch, self._P[0], self._P[1],
                                   self._A[0], self._A[1], device=self._device, **kw)
                self._pattern_key = key
                self._kept = False
                try:
                    self._qp.keep_workspace(True); self._kept = True
                except _lib.MpcqpError:
                    pass                                  # streaming kernel variant: every solve is a full setup
            self._qp.update(self._P[2], self.gradient, self._A[2], self.lowerBound, self.upperBound)
            self._vectors_dirty = self._matrices_dirty = self._solved_once = False
            if self._start is not None:
                self._qp.warm_start(self._start[0], self._start[1])
            self._qp.set_rho(self._rho0)
        except (_lib.MpcqpError, ValueError) as e:
            return _err("Failed to initialize solver. (%s)" % e)
        self.isInitialized_ = True
        return True

    def solve(self):
        if not self.isInitialized_:
            return _err("Solver not initialized. Call initSolver() first.")
        try:
            if self._vectors_dirty and self._kept and self._solved_once and not self._matrices_dirty:
                self._qp.update_vectors(self.gradient, self.lowerBound, self.upperBound)
            elif self._vectors_dirty or self._matrices_dirty:
                self._qp.update(self._P[2], self.gradient, self._A[2], self.lowerBound, self.upperBound)
            self._vectors_dirty = self._matrices_dirty = False
            self._qp.solve()
            self._solved_once = True
            self._result = self._qp.get()
        except _lib.MpcqpError as e:
            return _err("Failed to solve problem. Error code: %d" % e.code)
        if self._kw.get("verbose"):
            r = self._result
            print("mpcqp: status %s iters %s" % (sorted(set(r["status"].tolist())), sorted(set(r["iters"].tolist()))))
        return True

    # -- results (CuCaQP.cpp:213-224)
    def getSolution(self):
        if self._result is None:
            return np.zeros((self.batch, self.numOfVariables_))
        return self._result["x"]

    def getSolutionAsDM(self):
        return self.getSolution()

    def getDual(self):
        return None if self._result is None else self._result["y"]

    def getStatus(self):
        return None if self._result is None else self._result["status"]

    def getIterations(self):
        return None if self._result is None else self._result["iters"]

    def getInfo(self):
        return self._result

    def printSolverData(self):
        """CuCaQP.cpp:226-269: dumps q, l, u, P, A of instance 0 as the solver holds them (plus the scaling)."""
        print("q:", self.gradient[0]); print("l:", self.lowerBound[0]); print("u:", self.upperBound[0])
        for name, mat in (("P", self._P), ("A", self._A)):
            colptr, rowidx, values = mat
            vals = values if values.ndim == 1 else values[0]
            print("%s (nonzeros):" % name)
            for j in range(len(colptr) - 1):
                for k in range(colptr[j], colptr[j + 1]):
                    print("(%d,%d): %g" % (rowidx[k], j, vals[k]))
        if self._qp is not None and self._result is not None:
            D, E, c = self._qp.debug_scaling(0)
            print("scaling c:", c, "D:", D, "E:", E)

    def close(self):
        if self._qp is not None:
            self._qp.close()
            self._qp = None

