"""CPU restatement of the reference's local-system formulation -- TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module; the product (optimal_control_problem_amd/) never does.  It is the checker for SURVEY.md section 8 rows
a1 / a2 / f1: what `SQPOptimizationSolver` builds symbolically with CasADi and evaluates in `getLocalSystem`, restated with nothing but
plain callables and central finite differences -- no shared code with the product's host path (models.py: complex step, CSC patterns)
or device path (stage_eval.hip: forward-mode duals):

  reference src/sqp_solver/SQPOptimizationSolver.cpp:47-60    w = [p; x], the constraint vector c(w) = [p; x; g(p, x)]
  :55-60, AutoDifferentiator.cpp:16-28                         P = Hessian of f wrt w, q = gradient of f wrt w
  :61-66, AutoDifferentiator.cpp:132-136                       A = Jacobian of c wrt w = [I; dg/dw]
  :67-73, :100-120                                            l = [p; lbx; lbg] - c(w),  u = [p; ubx; ubg] - c(w)   (bounds shifted by the current point)

Parity is pinned on the reference's own known answers through the QP oracle (tests/test_oracle_golden.py, test/test.cpp:13-185); this file
adds no arithmetic of its own beyond differencing.  Accuracy: O(h^2) truncation + O(eps / h) rounding; the tests compare at 1e-5 relative,
far below any structural mistake (a missing row, a wrong sign, an unshifted bound).
"""
import numpy as np


def gradient(f, w, h=1e-5):
    g = np.zeros(len(w))
    for k in range(len(w)):
        e = np.zeros(len(w)); e[k] = h
        g[k] = (f(w + e) - f(w - e)) / (2 * h)
    return g


def hessian(f, w, h=1e-4):
    n = len(w); H = np.zeros((n, n)); f0 = f(w)
    E = np.eye(n) * h
    for i in range(n):
        H[i, i] = (f(w + E[i]) - 2 * f0 + f(w - E[i])) / (h * h)
        for j in range(i):
            H[i, j] = H[j, i] = (f(w + E[i] + E[j]) - f(w + E[i] - E[j]) - f(w - E[i] + E[j]) + f(w - E[i] - E[j])) / (4 * h * h)
    return H


def jacobian(g, w, h=1e-6):
    g0 = np.atleast_1d(g(w)); J = np.zeros((len(g0), len(w)))
    for k in range(len(w)):
        e = np.zeros(len(w)); e[k] = h
        J[:, k] = (np.atleast_1d(g(w + e)) - np.atleast_1d(g(w - e))) / (2 * h)
    return J


def local_system_dense(f, g, p, x, lbx, ubx, lbg, ubg):
    """dense (P, q, A, l, u) of the QP the reference hands to CuCaQP at the point (p, x); f(w) scalar, g(w) vector, w = [p; x]"""
    p = np.asarray(p, float); x = np.asarray(x, float)
    w = np.concatenate([p, x]); n = len(w)
    gv = np.atleast_1d(g(w)) if g is not None else np.zeros(0)
    P = hessian(f, w); q = gradient(f, w)
    A = np.vstack([np.eye(n), jacobian(g, w)]) if len(gv) else np.eye(n)
    c = np.concatenate([w, gv])
    l = np.concatenate([p, lbx, lbg]) - c
    u = np.concatenate([p, ubx, ubg]) - c
    return P, q, A, l, u
