"""Probe for a real OSQP on this machine -- TEST INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg, tests/).

The reference's CPU path is OSQP v1.0.0.beta1 behind osqp-eigen 0.9.0 (reference cpu_install.sh:4-6); neither is under
/root/reference and this image has no network, so the oracle (oracle/osqp_oracle.c) is a restatement of the published
algorithm.  SURVEY.md section 8(d) asks to look for an installed OSQP on the GPU box at bench time and to use it if
found: as the true CPU baseline, and as a pin for the oracle.  find() reports what exists; solve_batch() runs the Python
binding the way the reference drives OSQP: fresh setup per QP (CuCaQP::setSystem clears the solver, reference
src/sqp_solver/CuCaQP.cpp:271-288), settings of reference src/sqp_solver/SQPOptimizationSolver.cpp:81-85 (verbose off,
warm start flag on but moot, eps_abs = eps_rel = 1e-3, max_iter = 10000), everything else OSQP's defaults.
"""
import ctypes.util
import glob
import time

import numpy as np


def find():
    """{'python': module or None, 'version': str or None, 'libs': [paths of libosqp*]}"""
    mod = None; ver = None
    try:
        import osqp as mod          # noqa: F401
        ver = getattr(mod, "__version__", None)
    except Exception:
        mod = None
    libs = []
    hit = ctypes.util.find_library("osqp")
    if hit:
        libs.append(hit)
    for pat in ("/usr/lib/**/libosqp*", "/usr/local/lib/**/libosqp*", "/opt/**/libosqp*", "/usr/lib/python3*/**/libosqp*"):
        try:
            libs += glob.glob(pat, recursive=True)[:4]
        except OSError:
            pass
    return {"python": mod, "version": ver, "libs": sorted(set(libs))}


def solve_batch(mod, n, m, Pp, Pi, Ap, Ai, P, q, A, l, u, count, eps=1e-3, max_iter=10000):
    """Solves the first `count` instances one at a time with a fresh OSQP setup each; returns x, y, status strings, iterations
    and the wall time.  P is handed over as its upper triangle (OSQP's convention)."""
    import scipy.sparse as sp
    Pp = np.asarray(Pp); Pi = np.asarray(Pi); Ap = np.asarray(Ap); Ai = np.asarray(Ai)
    cols = np.repeat(np.arange(n), np.diff(Pp))
    up = Pi <= cols
    xs = np.empty((count, n)); ys = np.empty((count, m)); its = np.empty(count, np.int64); sts = []
    t0 = time.perf_counter()
    for b in range(count):
        Pv = P if P.ndim == 1 else P[b]; Av = A if A.ndim == 1 else A[b]
        Pm = sp.csc_matrix((Pv[up], (Pi[up], cols[up])), shape=(n, n))
        Am = sp.csc_matrix((Av, Ai, Ap), shape=(m, n))
        prob = mod.OSQP()
        prob.setup(Pm, q[b], Am, np.maximum(l[b], -1e30), np.minimum(u[b], 1e30), verbose=False, warm_starting=True, eps_abs=eps, eps_rel=eps, max_iter=max_iter)
        r = prob.solve()
        xs[b] = r.x; ys[b] = r.y; its[b] = r.info.iter; sts.append(str(r.info.status))
    return {"x": xs, "y": ys, "iters": its, "status": sts, "seconds": time.perf_counter() - t0}
