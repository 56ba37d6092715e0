"""General (slow) local-system evaluation for NLPs that are not stage-structured: what the reference does for ANY problem.

The reference accepts arbitrary CasADi SX expressions over the whole decision vector as cost terms and constraints
(reference src/OptimalControlProblem.cpp:444-497,574-600), assembles {x, f, g, p} (:235-240) and lets CasADi differentiate:
exact Hessian of f and Jacobian of [p; x; g] with their true sparsity (src/sqp_solver/SQPOptimizationSolver.cpp:47-77).  The fast
path of this build compiles the stage pattern (models.StageOCP, device-resident evaluation); everything else lands here:

* the cost f(p, x) and the constraints g(p, x), given as NumPy callables on 1-D arrays, are traced once into SSA tapes with the
  tracer of codegen.py (the same one that records user dynamics);
* the gradient is derived on the tape (reverse sweep), the Hessian's structure is read off a forward sweep over the gradient tape
  (identically-zero entries fold away), the Jacobian's structure off the dependency sets of g's outputs: the QP carries the exact
  sparsity, like CasADi's;
* per SQP iteration the values come from the tapes evaluated in NumPy over the whole batch at once -- gradient directly, Hessian and
  Jacobian columns by complex-step differentiation (exact to rounding for the analytic operations the tracer knows).

Cost: O(n) tape evaluations per iteration on the host.  The QPs are solved on the GPU by the generic path of the engine
(mpcqp_create takes any sparsity pattern).  The object has the interface SQPOptimizationSolver expects of a model (n, m, np,
local_system, objective), formulated exactly like the reference: w = [p; x], rows [p; x; g], shifted bounds.
"""
import numpy as np

from . import codegen
from .models import LocalSystem, _csc_from_dense_mask


def _flat(v):
    """tracer scalars / constants of whatever a traced callable returned: a vector tracer, a scalar tracer, numbers, or nested sequences of those"""
    if isinstance(v, codegen.TV):
        return list(v.items)
    if isinstance(v, codegen.TS):
        return [v]
    if isinstance(v, (list, tuple)):
        return [t for e in v for t in _flat(e)]
    return [float(a) for a in np.atleast_1d(np.asarray(v, float)).ravel()]


def _trace(fn, n_in, what):
    """tape of fn(w) over a traced vector w of n_in scalars; fn returns a vector tracer, a scalar tracer or a (nested) sequence of them"""
    tape = codegen.Tape(n_in)
    w = codegen.TV(tape, [codegen.TS(tape, k) for k in range(n_in)])
    tape.outputs = [v.idx if isinstance(v, codegen.TS) else tape.const(v) for v in _flat(fn(w))]
    if not tape.outputs and what == "cost":
        raise ValueError("the cost function returned nothing")
    return tape


def _depends(tape):
    """[n_out, n_in] structural dependency of every output on every input (reachability on the tape)"""
    n = tape.n_in
    dep = [None] * len(tape.nodes)
    for i, nd in enumerate(tape.nodes):
        k = nd[0]
        if k == "in":
            s = np.zeros(n, bool); s[nd[1]] = True
        elif k == "const":
            s = np.zeros(n, bool)
        elif len(nd) == 2:
            s = dep[nd[1]]
        else:
            s = dep[nd[1]] | dep[nd[2]]
        dep[i] = s
    return np.array([dep[o] for o in tape.outputs]).reshape(len(tape.outputs), n)


class GeneralNLP:
    """min f(p, x)  s.t.  lbx <= x <= ubx, lbg <= g(p, x) <= ubg, with f and g arbitrary traced NumPy callables of w = [p; x]."""

    name = "general_nlp"

    def __init__(self, nvar, npar, cost, constraints=None):
        self.nvar, self.np = int(nvar), int(npar)
        self.n = self.np + self.nvar
        self._ftape = _trace(lambda w: cost(w), self.n, "cost")
        if len(self._ftape.outputs) != 1:
            raise ValueError("the cost must be a scalar")
        self._gtape = codegen.gradient_tape(self._ftape)
        self.hm = codegen.hessian_mask(self._gtape)
        self._ctape = _trace(lambda w: constraints(w), self.n, "constraints") if constraints is not None else None
        self.ng = len(self._ctape.outputs) if self._ctape is not None else 0
        self.m = self.n + self.ng
        jm = _depends(self._ctape) if self.ng else np.zeros((0, self.n), bool)
        self.am = np.vstack([np.eye(self.n, dtype=bool), jm])
        self.Pp, self.Pi = _csc_from_dense_mask(self.hm)
        self.Ap, self.Ai = _csc_from_dense_mask(self.am)
        # columns that carry any derivative at all: the others are never perturbed
        self._hcols = np.nonzero(self.hm.any(axis=0))[0]
        self._jcols = np.nonzero(jm.any(axis=0))[0]

    # -- evaluation, vectorised over the batch: inputs[k] is an array [B]
    @staticmethod
    def _inputs(w):
        return [w[:, k] for k in range(w.shape[1])]

    def objective(self, p, x):
        w = np.concatenate([np.asarray(p, float), np.asarray(x, float)], axis=1)
        (f,) = self._ftape.evaluate(self._inputs(w))
        return np.broadcast_to(np.asarray(f, float), (w.shape[0],)).copy()

    def constraints(self, p, x):
        w = np.concatenate([np.asarray(p, float), np.asarray(x, float)], axis=1)
        return self._eval_vec(self._ctape, w) if self.ng else np.zeros((w.shape[0], 0))

    @staticmethod
    def _eval_vec(tape, w):
        out = tape.evaluate(GeneralNLP._inputs(w))
        return np.stack([np.broadcast_to(np.asarray(o), (w.shape[0],)) for o in out], axis=1)

    def derivatives(self, w):
        """gradient [B, n], Hessian [B, n, n], g [B, ng], Jacobian [B, ng, n] at w [B, n] (complex-step columns)"""
        B = w.shape[0]; h = 1e-30
        grad = self._eval_vec(self._gtape, w).real.astype(float)
        H = np.zeros((B, self.n, self.n)); J = np.zeros((B, self.ng, self.n))
        wc = w.astype(complex)
        for k in self._hcols:
            wc[:, k] += 1j * h
            H[:, :, k] = self._eval_vec(self._gtape, wc).imag / h
            wc[:, k] = w[:, k]
        gv = np.zeros((B, 0))
        if self.ng:
            gv = self._eval_vec(self._ctape, w).real.astype(float)
            for k in self._jcols:
                wc[:, k] += 1j * h
                J[:, :, k] = self._eval_vec(self._ctape, wc).imag / h
                wc[:, k] = w[:, k]
        H = 0.5 * (H + H.transpose(0, 2, 1))
        return grad, H, gv, J

    def local_system(self, p, x, lbx, ubx, lbg, ubg):
        """reference SQPOptimizationSolver.cpp:47-77,100-120: P = Hessian of f wrt w, q = gradient, A = [I; dg/dw], bounds shifted by the
        current [p; x; g]"""
        p = np.asarray(p, float); x = np.asarray(x, float)
        B = x.shape[0]
        w = np.concatenate([np.broadcast_to(p, (B, self.np)), x], axis=1)
        grad, H, gv, J = self.derivatives(w)
        Afull = np.concatenate([np.broadcast_to(np.eye(self.n), (B, self.n, self.n)), J], axis=1)
        P = H.transpose(0, 2, 1)[:, self.hm.T]          # column-major order of the masked entries
        A = Afull.transpose(0, 2, 1)[:, self.am.T]
        c = np.concatenate([w, gv], axis=1)
        lo = np.concatenate([w[:, :self.np], np.broadcast_to(lbx, (B, self.nvar)), np.broadcast_to(lbg, (B, self.ng))], axis=1) - c
        hi = np.concatenate([w[:, :self.np], np.broadcast_to(ubx, (B, self.nvar)), np.broadcast_to(ubg, (B, self.ng))], axis=1) - c
        return LocalSystem(self.n, self.m, self.Pp, self.Pi, self.Ap, self.Ai, P, grad, A, lo, hi, self.np)
