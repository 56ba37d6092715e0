"""Test infrastructure: the stage form's view of a models.LocalSystem (masks from its pattern, blocks from its values) and the CSC
pattern of a stage QP restated in NumPy."""
import numpy as np
import scipy.sparse as sp


def structure(ls):
    Ps = sp.csc_matrix((np.ones(len(ls.Pi)), ls.Pi, ls.Pp), shape=(ls.n, ls.n)).toarray() != 0
    As = sp.csc_matrix((np.ones(len(ls.Ai)), ls.Ai, ls.Ap), shape=(ls.m, ls.n)).toarray() != 0
    return Ps, As


def masks_of(ls, N, nx, nu, npar):
    """cost_mask over [s; u; p] and dyn_mask over [A_k B_k]: the union over the frames of what the CSC pattern holds"""
    Ps, As = structure(ls)
    f = nx + nu; nl = f + npar
    cm = np.zeros((nl, nl), bool)
    for k in range(N):
        sl = slice(npar + k * f, npar + (k + 1) * f)
        cm[:f, :f] |= Ps[sl, sl]; cm[f:, :f] |= Ps[:npar, sl]; cm[:f, f:] |= Ps[sl, :npar]
    cm[f:, f:] |= Ps[:npar, :npar]
    dm = np.zeros((nx, f), bool)
    for k in range(N - 1):
        dm |= As[ls.n + k * nx:ls.n + (k + 1) * nx, npar + k * f:npar + (k + 1) * f]
    return cm, dm


def dense_batch(ls):
    Pd = np.zeros((ls.batch, ls.n, ls.n)); Ad = np.zeros((ls.batch, ls.m, ls.n))
    for b in range(ls.batch):
        P, A = ls.dense(b)
        Pd[b] = np.triu(P) + np.triu(P, 1).T; Ad[b] = A
    return Pd, Ad


def numpy_pattern(N, nx, nu, npar, cm=None, dm=None):
    """the pattern include/mpcqp.h describes for the stage form, from dense boolean matrices (an independent restatement)"""
    f = nx + nu; n = npar + N * f; m = n + (N - 1) * nx
    cm = np.ones((f + npar, f + npar), bool) if cm is None else np.asarray(cm, bool)
    dm = np.ones((nx, f), bool) if dm is None else np.asarray(dm, bool)
    Pm = np.zeros((n, n), bool); Am = np.zeros((m, n), bool)
    Pm[:npar, :npar] = cm[f:, f:]
    Am[np.arange(n), np.arange(n)] = True
    for k in range(N):
        sl = slice(npar + k * f, npar + (k + 1) * f)
        Pm[sl, sl] = cm[:f, :f]; Pm[:npar, sl] = cm[f:, :f]; Pm[sl, :npar] = cm[:f, f:]
        if k >= 1:
            Am[n + (k - 1) * nx + np.arange(nx), npar + k * f + np.arange(nx)] = True
        if k < N - 1:
            Am[n + k * nx:n + (k + 1) * nx, sl] = dm
    from optimal_control_problem_amd import models
    Pp, Pi = models._csc_from_dense_mask(Pm); Ap, Ai = models._csc_from_dense_mask(Am)
    return n, m, Pp, Pi, Ap, Ai, Pm, Am


def random_ltv(N, nx, nu, B, seed):
    """a batch of LQ-structured QPs without a parameter block: dense SPD stage Hessians, stable-ish dynamics, the first state pinned,
    box bounds on the rest -> H, AB, q, l, u and the dense P, A they stand for"""
    rng = np.random.default_rng(seed)
    f = nx + nu; n = N * f; m = n + (N - 1) * nx
    H = np.zeros((B, N, f, f)); AB = np.zeros((B, N - 1, nx, f))
    for b in range(B):
        for k in range(N):
            M = rng.normal(size=(f, f)); H[b, k] = M @ M.T / f + 0.5 * np.eye(f)
        for k in range(N - 1):
            AB[b, k, :, :nx] = np.eye(nx) + 0.1 * rng.normal(size=(nx, nx)); AB[b, k, :, nx:] = 0.3 * rng.normal(size=(nx, nu))
    q = rng.normal(size=(B, n))
    l = np.full((B, m), -3.0); u = np.full((B, m), 3.0)
    x0 = rng.uniform(-1, 1, size=(B, nx))
    l[:, :nx] = x0; u[:, :nx] = x0                              # first state pinned (reference src/OptimalControlProblem.cpp:93-96)
    c = 0.05 * rng.normal(size=(B, (N - 1) * nx))
    l[:, n:] = c; u[:, n:] = c                                  # s_{k+1} - A s - B u = c
    Pd = np.zeros((B, n, n)); Ad = np.zeros((B, m, n))
    for b in range(B):
        Ad[b, :n, :n] = np.eye(n)
        for k in range(N):
            sl = slice(k * f, (k + 1) * f)
            Pd[b, sl, sl] = H[b, k]
            if k < N - 1:
                Ad[b, n + k * nx:n + (k + 1) * nx, sl] = -AB[b, k]
                Ad[b, n + k * nx + np.arange(nx), (k + 1) * f + np.arange(nx)] = 1.0
    return H, AB, q, l, u, Pd, Ad


def csc_values(D, colptr, rowidx):
    """values of dense matrices D [B, r, c] at a CSC pattern -> [B, nnz]"""
    cols = np.repeat(np.arange(len(colptr) - 1), np.diff(colptr))
    return np.ascontiguousarray(D[:, rowidx, cols])
