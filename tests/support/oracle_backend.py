"""A CuCaQP-shaped object backed by the CPU oracle -- TEST INFRASTRUCTURE ONLY.  Lets the host-side SQP driver
(optimal_control_problem_amd.sqp) be exercised on a machine without a GPU; never used by the product."""
import numpy as np

from oracle import oracle as orc


class OracleCuCaQP:
    def __init__(self, batch=1, nthreads=1):
        self.batch = batch; self.nthreads = nthreads
        self.kw = {}; self.ls = None; self.res = None; self._pat = None; self._key = None
        self._start = None; self._rho0 = None

    def setDimension(self, n, m):
        self.n, self.m = n, m
        return n > 0 and m > 0

    def setVerbosity(self, v): pass
    def setWarmStart(self, w): pass
    def setAbsoluteTolerance(self, t): self.kw["eps_abs"] = t
    def setRelativeTolerance(self, t): self.kw["eps_rel"] = t
    def setMaxIteration(self, k): self.kw["max_iter"] = k
    def setSystem(self, ls): self.ls = ls; self.res = None

    def setPrimalDualStart(self, x0, y0):
        self._start = None if x0 is None else (np.asarray(x0, float), np.asarray(y0, float))
        self.kw["warm_start"] = 0 if x0 is None else 1

    def setRhoStart(self, rho0):
        self._rho0 = None if rho0 is None else np.asarray(rho0, float)
    def initSolver(self): return self.ls is not None

    def solve(self):
        ls = self.ls
        key = (ls.n, ls.m, ls.Pi.tobytes(), ls.Ai.tobytes())
        if key != self._key:
            self._pat = orc.Pattern(ls.n, ls.m, ls.Pp, ls.Pi, ls.Ap, ls.Ai); self._key = key
        x0, y0 = self._start if self._start is not None else (None, None)
        self.res = self._pat.solve(ls.P, ls.q, ls.A, ls.l, ls.u, orc.default_settings(**self.kw), x0=x0, y0=y0,
                                   nthreads=self.nthreads, rho0=self._rho0)
        return True

    def getSolutionAsDM(self): return self.res["x"]
    def getInfo(self): return self.res
