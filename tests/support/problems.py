"""Shared problem builders for the test-suite (test infrastructure)."""
import numpy as np

from optimal_control_problem_amd import models


def toy_local_system(mdl, arg, x=None):
    p = np.asarray(arg["p"], float).reshape(1, -1)
    x = np.zeros((1, mdl.nx)) if x is None else np.asarray(x, float).reshape(1, -1)
    return mdl.local_system(p, x, np.asarray(arg["lbx"], float)[None], np.asarray(arg["ubx"], float)[None],
                            np.asarray(arg["lbg"], float).reshape(1, -1), np.asarray(arg["ubg"], float).reshape(1, -1))


def oracle_solve(ls, settings=None, nthreads=1, **kw):
    from oracle import oracle as orc
    pat = orc.Pattern(ls.n, ls.m, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
    s = settings or orc.default_settings(**kw)
    return pat.solve(ls.P, ls.q, ls.A, ls.l, ls.u, s, nthreads=nthreads)


def random_qp(n, m, seed, density=0.3, infeasible=None):
    """Random strictly convex QP with mixed equality / inequality / one-sided / free rows, dense patterns."""
    rng = np.random.default_rng(seed)
    Mx = rng.normal(size=(n, n)) * (rng.random((n, n)) < density)
    P = Mx @ Mx.T + 0.1 * np.eye(n)
    A = rng.normal(size=(m, n)) * (rng.random((m, n)) < density)
    for i in range(m):
        if not A[i].any():
            A[i, rng.integers(n)] = 1.0
    xf = rng.normal(size=n)
    ax = A @ xf
    l = ax - rng.random(m); u = ax + rng.random(m)
    kind = rng.integers(0, 4, size=m)
    l[kind == 1] = -np.inf; u[kind == 2] = np.inf
    eq = kind == 3
    u[eq] = l[eq] = ax[eq]
    q = rng.normal(size=n)
    if infeasible == "primal":      # two contradictory rows
        A[0] = 0; A[0, 0] = 1.0; A[1] = 0; A[1, 0] = 1.0
        l[0], u[0] = 1.0, 2.0; l[1], u[1] = -2.0, -1.0
    if infeasible == "dual":        # unbounded direction: zero curvature + free along e0
        P[0, :] = 0; P[:, 0] = 0; A[:, 0] = 0; q[0] = -1.0
    hm = np.ones((n, n), bool); am = np.ones((m, n), bool)
    Pp, Pi = models._csc_from_dense_mask(hm); Ap, Ai = models._csc_from_dense_mask(am)
    return models.LocalSystem(n, m, Pp, Pi, Ap, Ai, P.T[hm.T][None].copy(), q[None].copy(), A.T[am.T][None].copy(), l[None].copy(), u[None].copy())


def sparse_batch(n, m, B, seed, dens):
    """Batch of B convex QPs sharing a random SPARSE pattern (symmetric P mask with full diagonal, A mask with no empty row),
    mixed equality / inequality / one-sided rows."""
    rng = np.random.default_rng(seed)
    hm = np.eye(n, dtype=bool) | (rng.random((n, n)) < dens)
    hm = hm | hm.T
    am = rng.random((m, n)) < dens
    for i in range(m):
        if not am[i].any():
            am[i, rng.integers(n)] = True
    Pp, Pi = models._csc_from_dense_mask(hm); Ap, Ai = models._csc_from_dense_mask(am)
    Ps, As, qs, ls, us = [], [], [], [], []
    for b in range(B):
        Mx = rng.normal(size=(n, n)) * hm * (rng.random((n, n)) < 0.7)
        L = np.tril(Mx); P = L @ L.T
        P = P * hm + np.diag(np.abs(P).sum(axis=1) * (1 - hm).sum(axis=1) + 0.1)      # masked + diagonally dominant => PSD
        P = 0.5 * (P + P.T)
        A = rng.normal(size=(m, n)) * am
        xf = rng.normal(size=n); ax = A @ xf
        l = ax - rng.random(m); u = ax + rng.random(m)
        kind = rng.integers(0, 4, size=m)
        l[kind == 1] = -np.inf; u[kind == 2] = np.inf
        eq = kind == 3; u[eq] = l[eq] = ax[eq]
        Ps.append(P.T[hm.T]); As.append(A.T[am.T]); qs.append(rng.normal(size=n)); ls.append(l); us.append(u)
    return models.LocalSystem(n, m, Pp, Pi, Ap, Ai, np.array(Ps), np.array(qs), np.array(As), np.array(ls), np.array(us))




def reduce_qp(ls, fixed_rows):
    """NumPy statement of the reduced form (mpcqp_create_reduced): substitute the variables of the named equality singleton rows.
    Returns (reduced LocalSystem, free variable indices, kept row indices, fixed variable indices, x_fixed [B, nfix])."""
    import scipy.sparse as sp
    from optimal_control_problem_amd import models
    n, m, B = ls.n, ls.m, ls.batch
    fixed_rows = np.asarray(fixed_rows, int)
    Apat = sp.csc_matrix((np.arange(1, len(ls.Ai) + 1), ls.Ai, ls.Ap), shape=(m, n)).tocsr()
    fvars = np.array([Apat[i].indices[0] for i in fixed_rows]); asrc = np.array([Apat[i].data[0] - 1 for i in fixed_rows])
    free = np.setdiff1d(np.arange(n), fvars); kept = np.setdiff1d(np.arange(m), fixed_rows)
    Av = np.broadcast_to(ls.A, (B, len(ls.Ai))); Pv = np.broadcast_to(ls.P, (B, len(ls.Pi)))
    xfix = ls.l[:, fixed_rows] / Av[:, asrc]
    cols = np.repeat(np.arange(n), np.diff(ls.Pp)); colsA = np.repeat(np.arange(n), np.diff(ls.Ap))
    isfree = np.zeros(n, bool); isfree[free] = True; iskept = np.zeros(m, bool); iskept[kept] = True
    pk = isfree[ls.Pi] & isfree[cols]; ak = iskept[ls.Ai] & isfree[colsA]
    newv = -np.ones(n, int); newv[free] = np.arange(len(free)); newr = -np.ones(m, int); newr[kept] = np.arange(len(kept))
    def csc(rows, cs, ncol):
        ptr = np.zeros(ncol + 1, np.int64); np.add.at(ptr, cs + 1, 1); return np.cumsum(ptr).astype(np.int32), rows.astype(np.int32)
    Ppr, Pir = csc(newv[ls.Pi[pk]], newv[cols[pk]], len(free)); Apr, Air = csc(newr[ls.Ai[ak]], newv[colsA[ak]], len(free))
    qr = ls.q[:, free].copy(); lr = ls.l[:, kept].copy(); ur = ls.u[:, kept].copy()
    for b in range(B):
        Pd, Ad = ls.dense(b)
        Pd = np.triu(Pd) + np.triu(Pd, 1).T                      # only entries with row <= col count
        qr[b] += Pd[np.ix_(free, fvars)] @ xfix[b]
        shift = Ad[np.ix_(kept, fvars)] @ xfix[b]
        lr[b] = np.where(lr[b] <= -1e30, lr[b], lr[b] - shift); ur[b] = np.where(ur[b] >= 1e30, ur[b], ur[b] - shift)
    red = models.LocalSystem(len(free), len(kept), Ppr, Pir, Apr, Air, np.ascontiguousarray(Pv[:, pk]), qr, np.ascontiguousarray(Av[:, ak]), lr, ur)
    return red, free, kept, fvars, xfix


def random_stage_ocp(seed, family="oc4"):
    """One random stage-OCP local system for the on-chip kernel families (the generator of tools/fuzz_oc.py): random state / input sizes, horizon,
    weights, nonlinear dynamics and iterate -- block tridiagonal + arrow patterns with single and twisted chains, phantom slots, hubs that share
    their block with the last frame.  family "oc8": horizons drawn so that the chain part is 21 ... 56 blocks of 16 variables.  Returns
    (ls, dims string, rng); the problem the reference would hand to CuCaQP::setSystem at that iterate (src/sqp_solver/SQPOptimizationSolver.cpp:100-120)."""
    from optimal_control_problem_amd import models
    rng = np.random.default_rng(5000 + seed)
    nx = int(rng.integers(2, 13)); nu = int(rng.integers(1, 5)); N = int(rng.integers(4, 26)); B = int(rng.integers(1, 6))
    if family == "oc8":
        N = int(rng.integers((21 * 16) // (nx + nu) + 1, (56 * 16) // (nx + nu) + 1)); B = int(rng.integers(1, 4))
    Am = np.eye(nx) + 0.1 * rng.normal(size=(nx, nx)); Bm = 0.3 * rng.normal(size=(nx, nu)); w = rng.normal(size=nx)

    class M(models.StageOCP):
        name = "fuzz"

        def F(self, s, u):
            return s @ Am.T + u @ Bm.T + 0.05 * np.sin(s * w)

        def frame_bounds(self):
            return np.concatenate([np.full(nx, -5.0), np.full(nu, -1.0)]), np.concatenate([np.full(nx, 5.0), np.full(nu, 1.0)])
    M.nx, M.nu = nx, nu
    mdl = M(N, 0.05, rng.uniform(0.1, 10.0, nx), rng.uniform(0.01, 1.0, nu))
    x = rng.normal(0, 0.3, (B, mdl.nvar)); p = rng.normal(0, 0.2, (B, nx))
    lbx, ubx, lbg, ubg = mdl.stacked_bounds(x[:, :mdl.f].copy())
    ls = mdl.local_system(p, x, lbx, ubx, lbg, ubg)
    return ls, "nx=%d nu=%d N=%d B=%d n=%d m=%d" % (nx, nu, N, B, ls.n, ls.m), rng
