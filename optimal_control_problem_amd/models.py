"""Model zoo + batched local-system (QP) builder in the reference's formulation.

The reference builds one CasADi function (p, x, l, u) -> (H, grad, J, l - c, u - c) with augmented
variables w = [p; x] and augmented rows c = [p; x; g(p, x)]
(reference src/sqp_solver/SQPOptimizationSolver.cpp:47-77) and evaluates it at the current SQP iterate
with l = [p; lbx; lbg], u = [p; ubx; ubg] (SQPOptimizationSolver.cpp:100-120).  CasADi is not available
here, so every model below carries hand-derived (or complex-step) derivatives and a fixed structural
sparsity, batched over instances with NumPy.  Outputs are exactly what CuCaQP::setSystem receives
(reference src/sqp_solver/CuCaQP.cpp:271-288): P (both triangles, CSC), q, A (CSC), l, u.

Variable layout of stage problems follows OCPConfig: X is horizon x frameSize, stage-interleaved
(reference src/OCP_config/OCPConfig.cpp:29-46,102); the whole first frame is pinned through lbx/ubx
(reference src/OptimalControlProblem.cpp:93-96).
"""
import numpy as np

INF = float("inf")


class LocalSystem:
    """Batch of QPs sharing one sparsity: P [B,nnzP] (both triangles), q [B,n], A [B,nnzA], l,u [B,m]."""

    def __init__(self, n, m, Pp, Pi, Ap, Ai, P, q, A, l, u, np_=0):
        self.n, self.m, self.np = n, m, np_
        self.Pp, self.Pi, self.Ap, self.Ai = Pp, Pi, Ap, Ai
        self.P, self.q, self.A, self.l, self.u = P, q, A, l, u

    @property
    def batch(self):
        return self.q.shape[0]

    def dense(self, b=0):
        """Dense (P, A) of instance b, for small tests."""
        P = np.zeros((self.n, self.n)); A = np.zeros((self.m, self.n))
        Pv = self.P if self.P.ndim == 1 else self.P[b]
        Av = self.A if self.A.ndim == 1 else self.A[b]
        for j in range(self.n):
            for k in range(self.Pp[j], self.Pp[j + 1]):
                P[self.Pi[k], j] = Pv[k]
            for k in range(self.Ap[j], self.Ap[j + 1]):
                A[self.Ai[k], j] = Av[k]
        return P, A


def _csc_from_dense_mask(mask):
    """CSC (colptr, rowidx) of a boolean structure matrix."""
    rows, cols = mask.shape
    p = [0]; idx = []
    for j in range(cols):
        r = np.nonzero(mask[:, j])[0]
        idx.extend(r.tolist()); p.append(len(idx))
    return np.asarray(p, dtype=np.int32), np.asarray(idx, dtype=np.int32)


class DenseNLP:
    """Small NLP given by callables; structure is taken dense-by-mask.  Used for the reference's
    test/test.cpp cases (test/test.cpp:13-211).  f, grad, hess, g, jac act on w = [p; x] (1-D)."""

    def __init__(self, nx, np_, f, grad, hess, g, jac, hess_mask=None, jac_mask=None, name=""):
        self.nx, self.np, self.name = nx, np_, name
        self.n = nx + np_
        self.f, self.grad, self.hess, self.g, self.jac = f, grad, hess, g, jac
        w0 = np.zeros(self.n)
        self.ng = len(np.atleast_1d(g(w0))) if g is not None else 0
        self.m = self.n + self.ng
        hm = np.ones((self.n, self.n), bool) if hess_mask is None else hess_mask
        jm = np.ones((self.ng, self.n), bool) if jac_mask is None else jac_mask
        am = np.vstack([np.eye(self.n, dtype=bool), jm])
        self.hm, self.am = hm, am
        self.Pp, self.Pi = _csc_from_dense_mask(hm)
        self.Ap, self.Ai = _csc_from_dense_mask(am)

    def objective(self, p, x):
        return np.array([self.f(np.concatenate([p[b], x[b]])) for b in range(x.shape[0])])

    def local_system(self, p, x, lbx, ubx, lbg, ubg):
        B = x.shape[0]
        P = np.zeros((B, len(self.Pi))); A = np.zeros((B, len(self.Ai)))
        q = np.zeros((B, self.n)); l = np.zeros((B, self.m)); u = np.zeros((B, self.m))
        for b in range(B):
            w = np.concatenate([p[b], x[b]])
            H = np.atleast_2d(self.hess(w)); J = np.zeros((self.ng, self.n))
            gv = np.zeros(0)
            if self.ng:
                J = np.atleast_2d(self.jac(w)); gv = np.atleast_1d(self.g(w))
            Afull = np.vstack([np.eye(self.n), J])
            P[b] = H.T[self.hm.T]       # column-major order of masked entries
            A[b] = Afull.T[self.am.T]
            q[b] = self.grad(w)
            c = np.concatenate([w, gv])
            l[b] = np.concatenate([p[b], lbx[b], lbg[b]]) - c
            u[b] = np.concatenate([p[b], ubx[b], ubg[b]]) - c
        return LocalSystem(self.n, self.m, self.Pp, self.Pi, self.Ap, self.Ai, P, q, A, l, u, self.np)


def reference_test_cases():
    """The 8 NLPs of the reference's test/test.cpp (cases 1-8), as (model, arg, expected-or-None).
    INF there is float infinity (test/test.cpp:11)."""
    cases = []

    def quad(c):  # f = sum (x_i - c_i)^2
        c = np.asarray(c, float)
        return (lambda w: float(np.sum((w - c) ** 2)), lambda w: 2 * (w - c), lambda w: 2 * np.eye(len(c)))

    f, gr, he = quad([0, 0])
    m = DenseNLP(2, 0, f, gr, he, lambda w: np.array([w[0] + w[1] - 1]), lambda w: np.array([[1.0, 1.0]]), np.eye(2, dtype=bool), name="case1")
    cases.append((m, dict(lbx=[-50, -100], ubx=[50, 100], lbg=[0.0], ubg=[0.0], p=[]), [0.5, 0.5]))
    f, gr, he = quad([3, -2])
    m = DenseNLP(2, 0, f, gr, he, None, None, np.eye(2, dtype=bool), name="case2")
    cases.append((m, dict(lbx=[-50, -100], ubx=[50, 100], lbg=[], ubg=[], p=[]), [3, -2]))
    f, gr, he = quad([2, 3])
    m = DenseNLP(2, 0, f, gr, he, lambda w: np.array([w[0] + w[1] - 1]), lambda w: np.array([[1.0, 1.0]]), np.eye(2, dtype=bool), name="case3")
    cases.append((m, dict(lbx=[-100, -100], ubx=[100, 100], lbg=[1.0], ubg=[INF], p=[]), [2, 3]))
    f, gr, he = quad([0, 0])
    m = DenseNLP(2, 0, f, gr, he, lambda w: np.array([w[0], w[1]]), lambda w: np.eye(2), np.eye(2, dtype=bool), np.eye(2, dtype=bool), name="case4")
    cases.append((m, dict(lbx=[-100, -100], ubx=[100, 100], lbg=[1.0, 2.0], ubg=[INF, INF], p=[]), [1, 2]))
    f, gr, he = quad([1, 2, 3])
    m = DenseNLP(3, 0, f, gr, he, lambda w: np.array([w.sum() - 5]), lambda w: np.ones((1, 3)), np.eye(3, dtype=bool), name="case5")
    cases.append((m, dict(lbx=[0, 0, 0], ubx=[INF, INF, INF], lbg=[0.0], ubg=[0.0], p=[]), [2.0 / 3, 5.0 / 3, 8.0 / 3]))
    # case 6: w = [p; x1; x2], f = (x1 - p)^2 + x2^2
    f6 = lambda w: float((w[1] - w[0]) ** 2 + w[2] ** 2)
    g6 = lambda w: np.array([-2 * (w[1] - w[0]), 2 * (w[1] - w[0]), 2 * w[2]])
    h6 = lambda w: np.array([[2.0, -2, 0], [-2, 2, 0], [0, 0, 2]])
    hm6 = np.array([[1, 1, 0], [1, 1, 0], [0, 0, 1]], bool)
    m = DenseNLP(2, 1, f6, g6, h6, None, None, hm6, name="case6")
    cases.append((m, dict(lbx=[-100, -100], ubx=[100, 100], lbg=[], ubg=[], p=[5.0]), [5, 0]))
    f, gr, he = quad([3, 4])
    m = DenseNLP(2, 0, f, gr, he, None, None, np.eye(2, dtype=bool), name="case7")
    cases.append((m, dict(lbx=[0, 0], ubx=[2, 3], lbg=[], ubg=[], p=[]), [2, 3]))
    # case 8: non-convex objective, no pinned answer (test/test.cpp:187-211)
    f8 = lambda w: float(w[0] ** 2 - w[1] ** 2)
    m = DenseNLP(2, 0, f8, lambda w: np.array([2 * w[0], -2 * w[1]]), lambda w: np.diag([2.0, -2.0]),
                 lambda w: np.array([w[0] ** 2 + w[1] ** 2 - 1]), lambda w: np.array([[2 * w[0], 2 * w[1]]]), np.eye(2, dtype=bool), name="case8")
    cases.append((m, dict(lbx=[-100, -100], ubx=[100, 100], lbg=[-INF], ubg=[1.0], p=[]), None))
    return cases


# --------------------------------------------------------------------------------------------- stage OCPs
class StageOCP:
    """min sum_k (s_k - p)' Q (s_k - p) + u_k' R u_k   s.t.  s_{k+1} = F(s_k, u_k),  frame bounds,
    frame_k = [s_k; u_k], N frames, p = reference state (np = nx).  Q, R diagonal (addVectorCost,
    reference src/OptimalControlProblem.cpp:574-600, sums w_i e_i^2 -- no 1/2, so the Hessian is 2Q).
    Dynamics enter as (N-1)*nx equality rows g_k = s_{k+1} - F(s_k, u_k) in [0, 0]
    (addEquationConstraint, reference src/OptimalControlProblem.cpp:448-470)."""

    nx = 0; nu = 0; name = "ocp"
    # optional per-stage path constraint lo <= h(s_k, u_k) <= hi, nh rows per frame (addInequalityConstraint, reference
    # src/OptimalControlProblem.cpp:448-470); rows are stacked behind the dynamics rows: c = [p; x; g; h].  h_lo, h_hi: [nh] for
    # every frame, or [N, nh] when they differ by frame (a terminal constraint is loose, -inf / +inf, on every frame but the last)
    nh = 0; h_lo = None; h_hi = None
    # optional link constraint between consecutive frames, k_lo <= kfun(s_k, u_k, s_{k+1}, u_{k+1}) <= k_hi, nk rows per stage
    # k = 0 .. N-2 (rate limits u_{k+1} - u_k and the like; in the reference any SX over the whole decision vector can be a
    # constraint, src/OptimalControlProblem.cpp:448-489); rows behind the path rows: c = [p; x; g; h; r]
    nk = 0; k_lo = None; k_hi = None

    def kfun(self, s, u, sn, un):
        """[..., nk] link-constraint values; must accept complex input (override together with nk, k_lo, k_hi)"""
        raise NotImplementedError

    def link_bounds(self):
        """([N-1, nk], [N-1, nk]) bounds of the link constraint per stage (the same on every stage)"""
        return (np.broadcast_to(np.asarray(self.k_lo, float), (self.N - 1, self.nk)).copy(),
                np.broadcast_to(np.asarray(self.k_hi, float), (self.N - 1, self.nk)).copy())

    def dk(self, s, u, sn, un):
        """[..., nk, 2 f] Jacobian of kfun wrt [s; u; s_next; u_next] by complex-step differentiation"""
        eps = 1e-30
        args = [s.astype(complex), u.astype(complex), sn.astype(complex), un.astype(complex)]
        out = np.empty(s.shape[:-1] + (self.nk, 2 * self.f))
        col = 0
        for a, w in enumerate((self.nx, self.nu, self.nx, self.nu)):
            for c in range(w):
                pert = [v.copy() if i == a else v for i, v in enumerate(args)]
                pert[a][..., c] += 1j * eps
                out[..., :, col] = np.asarray(self.kfun(*pert)).imag / eps
                col += 1
        return out

    def link_values(self, x):
        s, u = self.frames(x)
        return np.asarray(self.kfun(s[:, :-1], u[:, :-1], s[:, 1:], u[:, 1:])).reshape(x.shape[0], -1)

    def path_bounds(self):
        """([N, nh], [N, nh]) bounds of the path constraint per frame"""
        return (np.broadcast_to(np.asarray(self.h_lo, float), (self.N, self.nh)).copy(),
                np.broadcast_to(np.asarray(self.h_hi, float), (self.N, self.nh)).copy())

    def hfun(self, s, u):
        """[..., nh] path-constraint values; must accept complex input (override together with nh, h_lo, h_hi)"""
        raise NotImplementedError

    # optional general stage cost: lcost(s, u, r) -> [...] replaces the diagonal tracking terms, f = sum_k lcost(s_k, u_k, p),
    # lterm the same for the last frame only.  Any SX expression can be a cost term in the reference (addScalarCost,
    # src/OptimalControlProblem.cpp:491-497) and its QP uses the exact Hessian (SQPOptimizationSolver.cpp:55-60); here the
    # callables are traced (codegen.trace_cost), the gradient is derived on the tape and the Hessian is its derivative.
    lcost = None; lterm = None

    def __init__(self, N, dt, Q, R):
        self.N, self.dt = int(N), float(dt)
        # diagonal weights, the same for every frame ([nx], [nu]) or one row per frame ([N, nx], [N, nu]: terminal costs,
        # ramps) -- addVectorCost is called per step in the reference (readme.md:121-128), so weights may differ by step
        self.Q = np.asarray(Q, float); self.R = np.asarray(R, float)
        self.varying_weights = self.Q.ndim == 2 or self.R.ndim == 2
        self.Qk = np.broadcast_to(self.Q, (int(N), self.nx)).copy(); self.Rk = np.broadcast_to(self.R, (int(N), self.nu)).copy()
        self.f = self.nx + self.nu
        self.np = self.nx
        self.nvar = self.N * self.f
        self.n = self.np + self.nvar
        self.ngd = (self.N - 1) * self.nx                 # dynamics rows
        self.ng = self.ngd + self.N * self.nh + (self.N - 1) * self.nk    # all general rows: dynamics, nh path rows per frame, nk link rows per stage
        self.m = self.n + self.ng
        self.general_cost = self.lcost is not None
        if self.general_cost:
            from . import codegen
            self._ltape, self._gtape = codegen.trace_cost(self.lcost, self.nx, self.nu, self.nx)
            mask = codegen.hessian_mask(self._gtape)
            self._lttape = self._gttape = None
            if self.lterm is not None:
                self._lttape, self._gttape = codegen.trace_cost(self.lterm, self.nx, self.nu, self.nx)
                mask = mask | codegen.hessian_mask(self._gttape)
            self.cost_mask = mask | np.eye(mask.shape[0], dtype=bool)
        self._build_pattern()
        if self.general_cost:
            self._build_cost_pattern()

    # -- structure -------------------------------------------------------------------------------
    def _build_pattern(self):
        nx, nu, f, N, npp, n = self.nx, self.nu, self.f, self.N, self.np, self.n
        # P (both triangles): diag everywhere, p_i <-> s_k[i] couplings
        Pp = [0]; Pi = []
        self._P_pp = np.zeros(npp, np.int64); self._P_ps = np.zeros((N, nx), np.int64)  # value slots
        self._P_sp = np.zeros((N, nx), np.int64); self._P_ss = np.zeros((N, nx), np.int64); self._P_uu = np.zeros((N, nu), np.int64)
        for i in range(npp):                       # column p_i
            self._P_pp[i] = len(Pi); Pi.append(i)
            for k in range(N):
                self._P_sp[k, i] = len(Pi); Pi.append(npp + k * f + i)   # row s_k[i], col p_i
            Pp.append(len(Pi))
        for k in range(N):
            for i in range(nx):                    # column s_k[i]
                self._P_ps[k, i] = len(Pi); Pi.append(i)
                self._P_ss[k, i] = len(Pi); Pi.append(npp + k * f + i)
                Pp.append(len(Pi))
            for i in range(nu):
                self._P_uu[k, i] = len(Pi); Pi.append(npp + k * f + nx + i)
                Pp.append(len(Pi))
        self.Pp = np.asarray(Pp, np.int32); self.Pi = np.asarray(Pi, np.int32)
        # A = [I_n; dg/dw]; dg_k/dframe_k dense nx x f, dg_k/ds_{k+1} = I
        Ap = [0]; Ai = []
        self._A_id = np.zeros(n, np.int64)
        self._A_next = np.zeros((N, nx), np.int64)          # slot of +1 in row g_{k-1}[i], col s_k[i] (k>=1)
        self._A_blk = np.zeros((N - 1, nx, f), np.int64)    # slot of -dF[r, c] for stage k
        nh = self.nh; nk = self.nk
        self._A_h = np.zeros((N, nh, f), np.int64)          # slot of +dh[r, c] for frame k
        self._A_k0 = np.zeros((max(N - 1, 0), nk, f), np.int64)   # slot of d r_k / d frame_k [r, c]
        self._A_k1 = np.zeros((max(N - 1, 0), nk, f), np.int64)   # slot of d r_k / d frame_{k+1} [r, c]
        krow0 = n + (N - 1) * nx + N * nh
        for j in range(npp):
            self._A_id[j] = len(Ai); Ai.append(j); Ap.append(len(Ai))
        for k in range(N):
            for c in range(f):
                j = npp + k * f + c
                self._A_id[j] = len(Ai); Ai.append(j)
                if k >= 1 and c < nx:
                    self._A_next[k, c] = len(Ai); Ai.append(n + (k - 1) * nx + c)
                if k < N - 1:
                    for r in range(nx):
                        self._A_blk[k, r, c] = len(Ai); Ai.append(n + k * nx + r)
                for r in range(nh):
                    self._A_h[k, r, c] = len(Ai); Ai.append(n + (N - 1) * nx + k * nh + r)
                if k >= 1:
                    for r in range(nk):
                        self._A_k1[k - 1, r, c] = len(Ai); Ai.append(krow0 + (k - 1) * nk + r)
                if k < N - 1:
                    for r in range(nk):
                        self._A_k0[k, r, c] = len(Ai); Ai.append(krow0 + k * nk + r)
                Ap.append(len(Ai))
        self.Ap = np.asarray(Ap, np.int32); self.Ai = np.asarray(Ai, np.int32)

    def _build_cost_pattern(self):
        """P structure of a general stage cost (csrc/stage_models.hpp sm_build_cost_pattern): both triangles, rows ascending;
        per entry the Hessian element it takes: (frame k or -1 = summed over the frames, local row, local column)"""
        nx, f, N, npp = self.nx, self.f, self.N, self.np
        mk = self.cost_mask
        Pp = [0]; Pi = []; src = []
        for i in range(npp):
            for r in range(npp):
                if mk[f + r, f + i]: Pi.append(r); src.append((-1, f + r, f + i))
            for k in range(N):
                for r in range(f):
                    if mk[r, f + i]: Pi.append(npp + k * f + r); src.append((k, r, f + i))
            Pp.append(len(Pi))
        for k in range(N):
            for c in range(f):
                for i in range(npp):
                    if mk[f + i, c]: Pi.append(i); src.append((k, f + i, c))
                for r in range(f):
                    if mk[r, c]: Pi.append(npp + k * f + r); src.append((k, r, c))
                Pp.append(len(Pi))
        self.Pp = np.asarray(Pp, np.int32); self.Pi = np.asarray(Pi, np.int32)
        self._P_src = np.asarray(src, np.int64)

    def _cost_inputs(self, p, x):
        s, u = self.frames(x)
        r = np.broadcast_to(p[:, None, :], s.shape)
        return [s[..., i] for i in range(self.nx)] + [u[..., i] for i in range(self.nu)] + [r[..., i] for i in range(self.nx)]

    def _cost_eval(self, tape, ttape, inputs):
        """outputs of tape on every frame [B, N], the last frame taken from ttape when there is a terminal cost"""
        shape = np.shape(inputs[0]); dt = np.result_type(*inputs)
        out = [np.broadcast_to(np.asarray(v), shape).astype(dt) for v in tape.evaluate(inputs)]
        if ttape is not None:
            for o, v in zip(out, ttape.evaluate(inputs)):
                o[:, -1] = np.broadcast_to(np.asarray(v), shape)[:, -1]
        return out

    def cost_derivatives(self, p, x):
        """(grad [B, N, nl], hess [B, N, nl, nl]) of the stage cost on every frame over [s; u; r]: gradient tape evaluated
        directly, Hessian columns by complex-step differentiation of the gradient tape (exact to rounding)"""
        inputs = self._cost_inputs(p, x)
        nl = len(inputs); B, N = inputs[0].shape
        grad = np.stack(self._cost_eval(self._gtape, self._gttape, inputs), axis=-1)
        hess = np.zeros((B, N, nl, nl)); eps = 1e-30
        for c in range(nl):
            ins = [v.astype(complex) for v in inputs]
            ins[c] = ins[c] + 1j * eps
            hess[..., :, c] = np.stack(self._cost_eval(self._gtape, self._gttape, ins), axis=-1).imag / eps
        return grad, hess

    # -- dynamics (override) ---------------------------------------------------------------------
    def cdyn(self, s, u):
        """continuous dynamics ds/dt, arrays [..., nx], [..., nu]; must accept complex input"""
        raise NotImplementedError

    def F(self, s, u):
        """discrete map, RK4 over dt"""
        h = self.dt
        k1 = self.cdyn(s, u); k2 = self.cdyn(s + 0.5 * h * k1, u)
        k3 = self.cdyn(s + 0.5 * h * k2, u); k4 = self.cdyn(s + h * k3, u)
        return s + (h / 6.0) * (k1 + 2 * k2 + 2 * k3 + k4)

    def dF(self, s, u):
        """[..., nx, f] Jacobian of F wrt [s; u] by complex-step differentiation (exact to rounding)."""
        eps = 1e-30
        out = np.empty(s.shape[:-1] + (self.nx, self.f))
        sc = s.astype(complex); uc = u.astype(complex)
        for c in range(self.f):
            if c < self.nx:
                sp = sc.copy(); sp[..., c] += 1j * eps
                out[..., :, c] = self.F(sp, uc).imag / eps
            else:
                up = uc.copy(); up[..., c - self.nx] += 1j * eps
                out[..., :, c] = self.F(sc, up).imag / eps
        return out

    # -- evaluation ------------------------------------------------------------------------------
    def frames(self, x):
        X = x.reshape(x.shape[0], self.N, self.f)
        return X[:, :, :self.nx], X[:, :, self.nx:]

    def objective(self, p, x):
        if self.general_cost:
            return self._cost_eval(self._ltape, self._lttape, self._cost_inputs(p, x))[0].sum(axis=1)
        s, u = self.frames(x)
        e = s - p[:, None, :]
        if self.varying_weights:
            return np.einsum("bki,ki->b", e * e, self.Qk) + np.einsum("bki,ki->b", u * u, self.Rk)
        return np.einsum("bki,i->b", e * e, self.Q) + np.einsum("bki,i->b", u * u, self.R)

    def dh(self, s, u):
        """[..., nh, f] Jacobian of hfun wrt [s; u] by complex-step differentiation"""
        eps = 1e-30
        out = np.empty(s.shape[:-1] + (self.nh, self.f))
        sc = s.astype(complex); uc = u.astype(complex)
        for c in range(self.f):
            if c < self.nx:
                sp = sc.copy(); sp[..., c] += 1j * eps
                out[..., :, c] = np.asarray(self.hfun(sp, uc)).imag / eps
            else:
                up = uc.copy(); up[..., c - self.nx] += 1j * eps
                out[..., :, c] = np.asarray(self.hfun(sc, up)).imag / eps
        return out

    def constraints(self, x):
        """dynamics defects only (the equality rows)"""
        s, u = self.frames(x)
        return (s[:, 1:, :] - self.F(s[:, :-1, :], u[:, :-1, :])).reshape(x.shape[0], -1)

    def path_values(self, x):
        s, u = self.frames(x)
        return np.asarray(self.hfun(s, u)).reshape(x.shape[0], -1)

    def local_system(self, p, x, lbx, ubx, lbg, ubg):
        B = x.shape[0]; N, nx, nu, f, npp, n = self.N, self.nx, self.nu, self.f, self.np, self.n
        s, u = self.frames(x)
        e = s - p[:, None, :]
        q = np.zeros((B, n))
        if self.general_cost:
            grad, hess = self.cost_derivatives(p, x)
            src = self._P_src; summed = src[:, 0] < 0
            P = np.zeros((B, len(self.Pi)))
            P[:, summed] = hess[:, :, src[summed, 1], src[summed, 2]].sum(axis=1)
            P[:, ~summed] = hess[:, src[~summed, 0], src[~summed, 1], src[~summed, 2]]
            q[:, :npp] = grad[:, :, f:].sum(axis=1)
            q[:, npp:] = grad[:, :, :f].reshape(B, -1)
        else:
            # Hessian values are constant
            Pv = np.zeros(len(self.Pi))
            Pv[self._P_pp] = 2.0 * self.Qk.sum(axis=0) if self.varying_weights else 2.0 * N * self.Q
            Pv[self._P_sp] = -2.0 * self.Qk; Pv[self._P_ps] = -2.0 * self.Qk
            Pv[self._P_ss] = 2.0 * self.Qk; Pv[self._P_uu] = 2.0 * self.Rk
            P = np.broadcast_to(Pv, (B, len(Pv))).copy()
            q[:, :npp] = -2.0 * (np.einsum("bki,ki->bi", e, self.Qk) if self.varying_weights else np.einsum("bki,i->bi", e, self.Q))
            qf = q[:, npp:].reshape(B, N, f)
            qf[:, :, :nx] = 2.0 * e * self.Qk; qf[:, :, nx:] = 2.0 * u * self.Rk
        J = self.dF(s[:, :-1, :], u[:, :-1, :])                 # [B, N-1, nx, f]
        A = np.zeros((B, len(self.Ai)))
        A[:, self._A_id] = 1.0
        A[:, self._A_next[1:].ravel()] = 1.0
        A[:, self._A_blk.ravel()] = -J.reshape(B, -1)
        g = (s[:, 1:, :] - self.F(s[:, :-1, :], u[:, :-1, :])).reshape(B, -1)
        if self.nh:
            A[:, self._A_h.ravel()] = self.dh(s, u).reshape(B, -1)
            g = np.concatenate([g, np.asarray(self.hfun(s, u)).reshape(B, -1)], axis=1)
        if self.nk:
            Jk = self.dk(s[:, :-1], u[:, :-1], s[:, 1:], u[:, 1:])          # [B, N-1, nk, 2 f]
            A[:, self._A_k0.ravel()] = Jk[..., :f].reshape(B, -1)
            A[:, self._A_k1.ravel()] = Jk[..., f:].reshape(B, -1)
            g = np.concatenate([g, self.link_values(x)], axis=1)
        c = np.concatenate([p, x, g], axis=1)
        l = np.concatenate([p, lbx, lbg], axis=1) - c
        uu = np.concatenate([p, ubx, ubg], axis=1) - c
        return LocalSystem(n, self.m, self.Pp, self.Pi, self.Ap, self.Ai, P, q, A, l, uu, npp)

    # -- bounds as computeOptimalTrajectory stacks them (OptimalControlProblem.cpp:93-99) ----------
    def frame_bounds(self):
        """per-frame (lower, upper) of length f; override"""
        return np.full(self.f, -INF), np.full(self.f, INF)

    def stacked_bounds(self, frame0):
        B = frame0.shape[0]
        lo, hi = self.frame_bounds()
        lbx = np.tile(lo, (B, self.N)); ubx = np.tile(hi, (B, self.N))
        lbx[:, :self.f] = frame0; ubx[:, :self.f] = frame0
        lbg = np.zeros((B, self.ng)); ubg = np.zeros((B, self.ng))
        if self.nh:
            lo, hi = self.path_bounds()
            lbg[:, self.ngd:self.ngd + self.N * self.nh] = lo.ravel(); ubg[:, self.ngd:self.ngd + self.N * self.nh] = hi.ravel()
        if self.nk:
            lo, hi = self.link_bounds()
            lbg[:, self.ngd + self.N * self.nh:] = lo.ravel(); ubg[:, self.ngd + self.N * self.nh:] = hi.ravel()
        return lbx, ubx, lbg, ubg


class DoubleIntegrator(StageOCP):
    """nx=2, nu=1 LQ-MPC (BASELINE.json configs[1]):
    A_d = [[1, dt], [0, 1]], B_d = [dt^2/2, dt], Q = diag(10, 1), R = 0.1, |u| <= 1, |v| <= 2."""
    nx = 2; nu = 1; name = "double_integrator"

    def __init__(self, N=20, dt=0.05):
        super().__init__(N, dt, [10.0, 1.0], [0.1])

    def F(self, s, u):
        h = self.dt
        pos = s[..., 0] + h * s[..., 1] + 0.5 * h * h * u[..., 0]
        vel = s[..., 1] + h * u[..., 0]
        return np.stack([pos, vel], axis=-1)

    def frame_bounds(self):
        return np.array([-INF, -2.0, -1.0]), np.array([INF, 2.0, 1.0])


class Quadrotor(StageOCP):
    """12-state quadrotor (pos, Euler roll/pitch/yaw, world velocity, body rates; 4 rotor thrusts),
    BASELINE.json configs[2] and the north-star N=20 size:
    Q = diag(10*1_3, 1*1_3, 1*1_3, 0.1*1_3), R = 0.1 I_4, 0 <= u_i <= 2 m g / 4, dt = 0.02."""
    nx = 12; nu = 4; name = "quadrotor"
    mass = 1.0; grav = 9.81; arm = 0.2; kappa = 0.05
    inertia = np.array([0.01, 0.01, 0.02])

    def __init__(self, N=20, dt=0.02):
        super().__init__(N, dt, [10.0] * 3 + [1.0] * 3 + [1.0] * 3 + [0.1] * 3, [0.1] * 4)

    @property
    def hover_thrust(self):
        return self.mass * self.grav / 4.0

    def cdyn(self, s, u):
        phi, th, psi = s[..., 3], s[..., 4], s[..., 5]
        v = s[..., 6:9]; w = s[..., 9:12]
        cph, sph, cth, sth, cps, sps = np.cos(phi), np.sin(phi), np.cos(th), np.sin(th), np.cos(psi), np.sin(psi)
        tth = sth / cth
        T = u[..., 0] + u[..., 1] + u[..., 2] + u[..., 3]
        a = T / self.mass
        # third column of R = Rz(psi) Ry(th) Rx(phi)
        ax = a * (cps * sth * cph + sps * sph)
        ay = a * (sps * sth * cph - cps * sph)
        az = a * (cth * cph) - self.grav
        p_, q_, r_ = w[..., 0], w[..., 1], w[..., 2]
        dphi = p_ + sph * tth * q_ + cph * tth * r_
        dth = cph * q_ - sph * r_
        dpsi = (sph * q_ + cph * r_) / cth
        Jx, Jy, Jz = self.inertia
        tx = self.arm * (u[..., 3] - u[..., 1]); ty = self.arm * (u[..., 2] - u[..., 0])
        tz = self.kappa * (u[..., 0] - u[..., 1] + u[..., 2] - u[..., 3])
        dp = (tx - (Jz - Jy) * q_ * r_) / Jx
        dq = (ty - (Jx - Jz) * p_ * r_) / Jy
        dr = (tz - (Jy - Jx) * p_ * q_) / Jz
        return np.stack([v[..., 0], v[..., 1], v[..., 2], dphi, dth, dpsi, ax, ay, az, dp, dq, dr], axis=-1)

    def frame_bounds(self):
        lo = np.full(self.f, -INF); hi = np.full(self.f, INF)
        lo[self.nx:] = 0.0; hi[self.nx:] = 2.0 * self.mass * self.grav / 4.0
        return lo, hi


class CartPole(StageOCP):
    """Cart-pole swing-up (BASELINE.json configs[3]): s = [x, theta, xdot,
    thetadot] with theta = 0 upright, |u| <= 20 N, track +-2.4 m, dt = 0.02; Gauss-Newton Hessian = 2Q."""
    nx = 4; nu = 1; name = "cartpole"
    mc = 1.0; mp = 0.1; length = 0.5; grav = 9.81

    def __init__(self, N=100, dt=0.02):
        super().__init__(N, dt, [1.0, 10.0, 0.1, 0.1], [0.01])

    def cdyn(self, s, u):
        th, xd, thd = s[..., 1], s[..., 2], s[..., 3]
        F_ = u[..., 0]
        sn, cs = np.sin(th), np.cos(th)
        tot = self.mc + self.mp
        tmp = (F_ + self.mp * self.length * thd * thd * sn) / tot
        thdd = (self.grav * sn - cs * tmp) / (self.length * (4.0 / 3.0 - self.mp * cs * cs / tot))
        xdd = tmp - self.mp * self.length * thdd * cs / tot
        return np.stack([xd, thd, xdd, thdd], axis=-1)

    def frame_bounds(self):
        return np.array([-2.4, -INF, -INF, -INF, -20.0]), np.array([2.4, INF, INF, INF, 20.0])


# --------------------------------------------------------------------------------------------- workloads
def make_workload(name, batch, seed=None, N=None):
    """Seeded synthetic QP batches of SURVEY.md section 8(d).  Returns (model, LocalSystem, meta)."""
    if name == "double_integrator":
        mdl = DoubleIntegrator(N or 20, 0.05); rng = np.random.default_rng(1234 if seed is None else seed)
        x0 = rng.uniform([-3.0, -1.5], [3.0, 1.5], size=(batch, 2))
        frame0 = np.concatenate([x0, np.zeros((batch, 1))], axis=1)
        p = np.zeros((batch, 2))
        xit = np.zeros((batch, mdl.nvar))                      # the reference starts SQP at x = 0
    elif name == "quadrotor":
        mdl = Quadrotor(N or 20, 0.02); rng = np.random.default_rng(2024 if seed is None else seed)
        x0 = np.zeros((batch, 12))
        x0[:, 0:3] = rng.normal(0.0, 0.5, size=(batch, 3)); x0[:, 3:6] = rng.normal(0.0, 0.1, size=(batch, 3))
        hov = mdl.hover_thrust
        frame0 = np.concatenate([x0, np.full((batch, 4), hov)], axis=1)
        p = np.zeros((batch, 12))
        # per-instance perturbed-hover iterate: every instance has its own Jacobian blocks
        X = np.zeros((batch, mdl.N, mdl.f))
        X[:, :, :12] = x0[:, None, :] + rng.normal(0.0, 0.05, size=(batch, mdl.N, 12))
        X[:, :, 12:] = hov + rng.normal(0.0, 0.3, size=(batch, mdl.N, 4))
        X[:, 0, :] = frame0
        xit = X.reshape(batch, -1)
    elif name == "cartpole":
        mdl = CartPole(N or 100, 0.02); rng = np.random.default_rng(7 if seed is None else seed)
        x0 = np.zeros((batch, 4)); x0[:, 1] = np.pi + rng.normal(0.0, 0.05, size=batch)
        frame0 = np.concatenate([x0, np.zeros((batch, 1))], axis=1)
        p = np.zeros((batch, 4))
        X = np.zeros((batch, mdl.N, mdl.f))
        X[:, :, :4] = x0[:, None, :] * np.linspace(1.0, 0.0, mdl.N)[None, :, None] + rng.normal(0.0, 0.02, size=(batch, mdl.N, 4))
        X[:, :, 4:] = rng.normal(0.0, 1.0, size=(batch, mdl.N, 1))
        X[:, 0, :] = frame0
        xit = X.reshape(batch, -1)
    else:
        raise ValueError("unknown workload %r" % name)
    lbx, ubx, lbg, ubg = mdl.stacked_bounds(frame0)
    ls = mdl.local_system(p, xit, lbx, ubx, lbg, ubg)
    meta = dict(frame0=frame0, p=p, x_iterate=xit, lbx=lbx, ubx=ubx, lbg=lbg, ubg=ubg)
    return mdl, ls, meta
