// kernel_stream.hpp -- mpcqp_admm_kernel<PD>: the first-generation streaming kernel, one QP per wavefront
// Part of the single translation unit mpcqp.hip (included there in order; not a stand-alone header).
#pragma once

// ------------------------------------------------------------------------------------------ the kernel
template <int PD>
__global__ void __launch_bounds__(WAVE) mpcqp_admm_kernel(const DevPlan pl, const mpcqp_settings st, const DevIO io) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int b = io.order ? io.order[blockIdx.x] : blockIdx.x, lane = threadIdx.x;
  Ctx cx;
  cx.pl = &pl; cx.st = &st;
  cx.X = lds; cx.Q = cx.X + pl.npad; cx.R = cx.Q + pl.npad;
  cx.Z = cx.R + pl.npad; cx.Y = cx.Z + pl.mpad; cx.W = cx.Y + pl.mpad;
  // the two 16x17 Cholesky tiles alias the rhs vector R when it is large enough: R only carries the singleton
  // diagonal during assembly and is rebuilt at the start of every ADMM iteration (one more QP per CU for N = 50)
  if (pl.npad >= 2 * BS * 17) { cx.S0 = cx.R; cx.S1 = cx.R + BS * 17; }
  else { cx.S0 = cx.W + pl.mpad; cx.S1 = cx.S0 + BS * 17; }
  double *ws = io.ws + (long)b * pl.ws_stride; cx.ws = ws;
  double *valA = ws + pl.o_ellA, *valAt = ws + pl.o_ellAt, *valP = ws + pl.o_ellP;
  double *lb = ws + pl.o_l, *ub = ws + pl.o_u, *Dg = ws + pl.o_D, *Eg = ws + pl.o_E;
  const double *inP = io.P + (long)b * io.sP, *inA = io.A + (long)b * io.sA, *inq = io.q + (long)b * io.sq;
  const double *inl = io.l + (long)b * io.sl, *inu = io.u + (long)b * io.su;
  const int n = pl.n, m = pl.m, npad = pl.npad, mpad = pl.mpad;
  cx.unscale = st.scaling && !st.scaled_termination;

  // ---- load: caller's CSC values -> ELL slabs (CuCaQP::setSystem, reference CuCaQP.cpp:271-288)
  for (long e = lane; e < pl.A.entries; e += WAVE) { const int s = pl.A.src[e]; valA[e] = s >= 0 ? inA[s] : 0.0; }
  for (long e = lane; e < pl.At.entries; e += WAVE) { const int s = pl.At.src[e]; valAt[e] = s >= 0 ? inA[s] : 0.0; }
  for (long e = lane; e < pl.P.entries; e += WAVE) { const int s = pl.P.src[e]; valP[e] = s >= 0 ? inP[s] : 0.0; }
  for (int t = lane; t < npad; t += WAVE) { cx.Q[t] = 0.0; cx.R[t] = 1.0; }
  for (int i = lane; i < mpad; i += WAVE) cx.W[i] = 1.0;
  wsync();
  for (int j = lane; j < n; j += WAVE) cx.Q[pl.pos[j]] = inq[j];
  wsync();

  // ---- modified Ruiz equilibration (oracle scale_data): D in R, E in W, temporaries in X / Z
  double c = 1.0;
  for (int it = 0; it < st.scaling; it++) {
    {
      const int lane_ = lane;
      for (int ch = 0; ch < pl.At.nchunks; ch++) {
        const int t = ch * WAVE + lane_;
        double nA = 0.0, nP = 0.0;
        for (int s = pl.At.chunk_off[ch]; s < pl.At.chunk_off[ch + 1]; s++) { const long e = (long)s * WAVE + lane_; nA = fmax(nA, fabs(valAt[e]) * cx.W[pl.At.idx[e]]); }
        for (int s = pl.P.chunk_off[ch]; s < pl.P.chunk_off[ch + 1]; s++) { const long e = (long)s * WAVE + lane_; nP = fmax(nP, fabs(valP[e]) * cx.R[pl.P.idx[e]]); }
        if (t < npad) { const double dj = cx.R[t]; cx.X[t] = 1.0 / sqrt(limit_scaling(fmax(c * dj * nP, dj * nA))); }
      }
    }
    ell_rowmax(pl.A, valA, cx.R, [&](int i, double v) { if (i < mpad) cx.Z[i] = 1.0 / sqrt(limit_scaling(cx.W[i] * v)); });
    wsync();
    for (int t = lane; t < npad; t += WAVE) cx.R[t] *= cx.X[t];
    for (int i = lane; i < mpad; i += WAVE) cx.W[i] *= cx.Z[i];
    wsync();
    double sum = 0.0, qn = 0.0;
    ell_rowmax(pl.P, valP, cx.R, [&](int t, double v) { if (t < npad) { sum += c * cx.R[t] * v; qn = fmax(qn, fabs(c * cx.R[t] * cx.Q[t])); } });
    sum = wave_sum(sum); qn = wave_max(qn);
    const double ct = 1.0 / limit_scaling(fmax(sum / (double)n, limit_scaling(qn)));
    c *= ct;
    wsync();
  }
  cx.c = c; cx.cinv = 1.0 / c;
  // apply scaling: A <- E A D, P <- c D P D, q <- c D q, l,u <- E l, E u (bounds clipped to +-1e30 first)
  for (int ch = 0; ch < pl.A.nchunks; ch++) {
    const int i = ch * WAVE + lane; const double ei = cx.W[i];
    for (int s = pl.A.chunk_off[ch]; s < pl.A.chunk_off[ch + 1]; s++) { const unsigned e = (unsigned)s * WAVE + lane; valA[e] *= ei * cx.R[pl.A.idx[e]]; }
  }
  for (int ch = 0; ch < pl.At.nchunks; ch++) {
    const int t = ch * WAVE + lane; const double dj = t < npad ? cx.R[t] : 0.0;
    for (int s = pl.At.chunk_off[ch]; s < pl.At.chunk_off[ch + 1]; s++) { const unsigned e = (unsigned)s * WAVE + lane; valAt[e] *= dj * cx.W[pl.At.idx[e]]; }
    for (int s = pl.P.chunk_off[ch]; s < pl.P.chunk_off[ch + 1]; s++) { const unsigned e = (unsigned)s * WAVE + lane; valP[e] *= c * dj * cx.R[pl.P.idx[e]]; }
  }
  for (int t = lane; t < npad; t += WAVE) { cx.Q[t] *= c * cx.R[t]; Dg[t] = cx.R[t]; }
  for (int i = lane; i < mpad; i += WAVE) {
    const double ei = cx.W[i];
    Eg[i] = ei;
    lb[i] = i < m ? ei * fmax(inl[i], -Q_INFTY) : 0.0;
    ub[i] = i < m ? ei * fmin(inu[i], Q_INFTY) : 0.0;
  }
  wsync();

  // ---- start point (cold: x = z = y = 0; warm: oracle solve_one / osqp_warm_start)
  for (int t = lane; t < npad; t += WAVE) cx.X[t] = 0.0;
  for (int i = lane; i < mpad; i += WAVE) { cx.Z[i] = 0.0; cx.Y[i] = 0.0; }
  wsync();
  if (st.warm_start && io.x0 && io.y0) {
    for (int j = lane; j < n; j += WAVE) { const int t = pl.pos[j]; cx.X[t] = io.x0[(long)b * n + j] * (1.0 / Dg[t]); }
    for (int i = lane; i < m; i += WAVE) cx.Y[i] = io.y0[(long)b * m + i] * (1.0 / Eg[i]) * c;
    wsync();
    ell_rows(pl.A, valA, cx.X, [&](int i, double ax) { if (i < m) cx.Z[i] = ax; });
    wsync();
  }
  cx.rho = fmin(fmax(io.rho0 && io.rho0[b] > 0.0 ? io.rho0[b] : st.rho, Q_RHO_MIN), Q_RHO_MAX);
  int status = MPCQP_UNSOLVED, iter_done = 0;
  Info in; memset(&in, 0, sizeof(in));
  bool ok = factorize(cx);
  if (!ok) status = MPCQP_NON_CVX;

  // ---- ADMM loop (OSQP Algorithm 1; oracle solve_one)
  int interval = st.adaptive_rho_interval;
  if (st.adaptive_rho && interval == 0) interval = st.check_termination ? 4 * st.check_termination : 100;
  const double alpha = st.alpha, sigma = st.sigma;
  const double *Lf = ws + pl.o_Lf, *Lbk = ws + pl.o_Lb;
  double *dxg = ws + pl.o_dx, *dyg = ws + pl.o_dy;
  int can_check = 0;
  if (ok) {
    int iter;
    for (iter = 1; iter <= st.max_iter; iter++) {
      // rhs = sigma x - q + A'(rho z - y)
      ell_rows(pl.At, valAt, cx.W, [&](int t, double v) { if (t < npad) cx.R[t] = sigma * cx.X[t] - cx.Q[t] + v; });
      wsync();
      // xtilde = M^-1 rhs
      run_stream<PD>(Lf, pl.fwd_ops, pl.nblk, cx.R);
      run_stream<PD>(Lbk, pl.bwd_ops, pl.nblk, cx.R);
      can_check = st.check_termination && (iter % st.check_termination == 0);
      const int do_rho = st.adaptive_rho && interval && (iter % interval == 0);
      const int save = can_check || do_rho;
      // ztilde = A xtilde, fused with the z / y updates (projection onto [l, u]) and w = rho z - y
      ell_rows(pl.A, valA, cx.R, [&](int i, double zt) {
        if (i < m) {
          const double lo = lb[i], up = ub[i], rh = rho_of(lo, up, cx.rho), rinv = 1.0 / rh;
          const double zr = alpha * zt + (1.0 - alpha) * cx.Z[i], yo = cx.Y[i];
          const double zn = fmin(fmax(zr + rinv * yo, lo), up);
          const double dy = rh * (zr - zn), yn = yo + dy;
          cx.Z[i] = zn; cx.Y[i] = yn; cx.W[i] = rh * zn - yn;
          if (save) dyg[i] = dy;
        }
      });
      for (int t = lane; t < npad; t += WAVE) {
        const double xo = cx.X[t], xn = alpha * cx.R[t] + (1.0 - alpha) * xo;
        cx.X[t] = xn;
        if (save) dxg[t] = xn - xo;
      }
      wsync();
      iter_done = iter;
      if (can_check) {
        update_info(cx, in);
        status = check_termination(cx, in, 0);
        if (status != MPCQP_UNSOLVED) break;
      }
      if (do_rho) {
        if (!can_check) update_info(cx, in);
        const double pr = in.prs / (fmax(in.nzs, in.naxs) + Q_DIV_TOL);
        const double dr = in.drs / (fmax(in.nqs, fmax(in.natys, in.npxs)) + Q_DIV_TOL);
        double rn = cx.rho * sqrt(pr / (dr + Q_DIV_TOL));
        rn = fmin(fmax(rn, Q_RHO_MIN), Q_RHO_MAX);
        if (rn > cx.rho * st.adaptive_rho_tolerance || rn < cx.rho / st.adaptive_rho_tolerance) {
          cx.rho = rn;
          if (!factorize(cx)) { status = MPCQP_NON_CVX; break; }
        }
      }
    }
    if (iter > st.max_iter) iter_done = st.max_iter;
    if (status == MPCQP_UNSOLVED) {
      if (!can_check) { update_info(cx, in); status = check_termination(cx, in, 0); }
      if (status == MPCQP_UNSOLVED) { status = check_termination(cx, in, 1); if (status == MPCQP_UNSOLVED) status = MPCQP_MAX_ITER_REACHED; }
    }
  }

  // ---- store_solution: x = D x, y = E y / c, z = z / E; NaN where no solution is defined
  const bool bad = status == MPCQP_PRIMAL_INFEASIBLE || status == MPCQP_PRIMAL_INFEASIBLE_INACCURATE ||
                   status == MPCQP_DUAL_INFEASIBLE || status == MPCQP_DUAL_INFEASIBLE_INACCURATE || status == MPCQP_NON_CVX;
  for (int j = lane; j < n; j += WAVE) { const int t = pl.pos[j]; io.x[(long)b * n + j] = bad ? NAN : Dg[t] * cx.X[t]; }
  for (int i = lane; i < m; i += WAVE) {
    io.y[(long)b * m + i] = bad ? NAN : cx.cinv * Eg[i] * cx.Y[i];
    io.z[(long)b * m + i] = bad ? NAN : (1.0 / Eg[i]) * cx.Z[i];
  }
  if (lane == 0) {
    io.status[b] = status; io.iters[b] = iter_done;
    io.info[4L * b] = in.obj; io.info[4L * b + 1] = in.prim_res; io.info[4L * b + 2] = in.dual_res; io.info[4L * b + 3] = cx.rho;
    io.cscale[b] = c;
  }
}
