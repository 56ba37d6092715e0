"""SQPOptimizationSolver -- batched host-side mirror of the reference's SQP outer loop.

Follows reference src/sqp_solver/SQPOptimizationSolver.cpp:127-216 step for step, for B independent
instances at once: a fixed number of iterations (options["max_iter"], the YAML's SQP_settings.step_num),
each = evaluate the local system at the current iterate (getLocalSystem, :100-120) -> qpSolver_.setSystem /
initSolver / solve (:155-157, return values ignored) -> result.x += alpha * solution[pSize:] (:171-177) ->
objective (:180-181).  Quirks kept on purpose: arg["x0"] is ignored and the iterate persists
across calls (zero-initialised, :88-91); no line search or convergence test; the ||dx|| < 1e-6 early stop
exists only when verbose (:183-197).

The model supplies what CasADi's generated localSystemFunction_ supplies in the reference
(optimal_control_problem_amd.models).  The QP backend is any object with the CuCaQP interface.
"""
import time

import numpy as np

from .cucaqp import CuCaQP


class SQPOptimizationSolver:
    def __init__(self, nlp, options, batch=1, qp_solver=None):
        """nlp: a models.* object (local_system / objective / n, m, np, nx...).  options: max_iter, alpha, verbose."""
        self.model = nlp
        self.stepNum_ = int(options["max_iter"])
        self.alpha_ = float(options["alpha"])
        self.verbose_ = bool(options.get("verbose", False))
        self.batch = int(batch)
        self.qpSolver_ = qp_solver if qp_solver is not None else CuCaQP(batch=self.batch)
        # reference SQPOptimizationSolver.cpp:80-85
        self.qpSolver_.setDimension(nlp.n, nlp.m)
        self.qpSolver_.setVerbosity(False)
        self.qpSolver_.setWarmStart(True)
        self.qpSolver_.setAbsoluteTolerance(1e-3)
        self.qpSolver_.setRelativeTolerance(1e-3)
        self.qpSolver_.setMaxIteration(10000)
        nvar = nlp.n - nlp.np
        self.result_ = {"x": np.zeros((self.batch, nvar)), "f": np.zeros(self.batch)}
        self.timings = {"local_system_ms": 0.0, "qp_ms": 0.0}
        self.last_qp_info = None
        # extension (BASELINE config 4): warm-start each QP's ADMM from the previous SQP
        # iteration's solution.  Off by default: the reference cold-starts every QP (CuCaQP.cpp:271-288).
        self.warm_start_admm = bool(options.get("warm_start_admm", False))
        self.admm_iterations = []

    def setVerbose(self, verbose):
        self.verbose_ = bool(verbose)
        self.qpSolver_.setVerbosity(verbose)

    def getLocalSystem(self, arg):
        B = self.batch
        as2d = lambda a, w: np.broadcast_to(np.asarray(a, float).reshape(-1, w) if w else np.zeros((1, 0)), (B, w))
        p = as2d(arg.get("p", np.zeros(0)), self.model.np)
        nvar = self.model.n - self.model.np; ng = self.model.m - self.model.n
        return self.model.local_system(p, self.result_["x"], as2d(arg["lbx"], nvar), as2d(arg["ubx"], nvar),
                                       as2d(arg["lbg"], ng), as2d(arg["ubg"], ng))

    def getOptimalSolution(self, arg):
        B = self.batch
        pSize = self.model.np
        p = np.broadcast_to(np.asarray(arg.get("p", np.zeros(0)), float).reshape(-1, pSize) if pSize else np.zeros((1, 0)), (B, pSize))
        for i in range(self.stepNum_):
            t0 = time.perf_counter()
            localSystem = self.getLocalSystem(arg)
            t1 = time.perf_counter()
            self.qpSolver_.setSystem(localSystem)
            if self.warm_start_admm and hasattr(self.qpSolver_, "setPrimalDualStart"):
                info = self.last_qp_info
                if info is not None and np.isfinite(info["x"]).all() and np.isfinite(info["y"]).all():
                    # after the damped update x += alpha * dx the remaining step is (1 - alpha) * dx; duals carry over
                    self.qpSolver_.setPrimalDualStart((1.0 - self.alpha_) * info["x"], info["y"])
            self.qpSolver_.initSolver()
            self.qpSolver_.solve()
            t2 = time.perf_counter()
            self.timings["local_system_ms"] += (t1 - t0) * 1e3
            self.timings["qp_ms"] += (t2 - t1) * 1e3
            solution = np.asarray(self.qpSolver_.getSolutionAsDM(), float).reshape(B, -1)
            self.last_qp_info = getattr(self.qpSolver_, "getInfo", lambda: None)()
            if self.last_qp_info is not None and "iters" in self.last_qp_info:
                self.admm_iterations.append(np.asarray(self.last_qp_info["iters"]).copy())
            oldRes = self.result_["x"].copy()
            self.result_["x"] = self.result_["x"] + self.alpha_ * solution[:, pSize:]
            self.result_["f"] = self.model.objective(p, self.result_["x"])
            if self.verbose_:
                normDelta = np.linalg.norm(self.result_["x"] - oldRes, axis=1).max()
                print("SQP iter %d/%d  max|dx| %.3e  f[0] %.6g" % (i + 1, self.stepNum_, normDelta, self.result_["f"][0]))
                if normDelta < 1e-6:
                    break
        return {"x": self.result_["x"].copy(), "f": self.result_["f"].copy()}
