// stage_kernels.hpp -- device templates of the local-system evaluation (see stage_eval.hip for the design notes).
// Included by stage_eval.hip (the built-in zoo) and by the translation units optimal_control_problem_amd/codegen.py
// generates for user-defined dynamics (the native analogue of the reference's gen_code / load_lib flow,
// reference src/OptimalControlProblem.cpp:263-287,602-640).  STAGE_ABI_VERSION guards the StageDev layout shared between
// libmpcqp.so and a generated library.
#pragma once
#include <hip/hip_runtime.h>
#include "stage_models.hpp"

#define STAGE_ABI_VERSION 6

struct StageDev {
  int model, N, nx, nu, f, np, n, m, ng, nvar, nnzP, nnzA;   // ng = all general rows: (N-1)*nx dynamics rows, then N*nh path rows, then (N-1)*nk link rows
  int nh, ngd, nk;
  double h_lo[SM_MAXNH], h_hi[SM_MAXNH];   // path-constraint bounds, for the merit kernel's violation measure
  double k_lo[SM_MAXNK], k_hi[SM_MAXNK];   // link-constraint bounds (the same on every stage), likewise
  double dt;
  double Q[SM_MAXNX], R[SM_MAXNU], par[SM_NPAR];
  const int *Pp, *Ap;   // device copies of the column pointers
  const double *Qk, *Rk;   // optional per-frame diagonal weights [N * nx], [N * nu] (device; NULL = Q, R for every frame)
  const unsigned char *hmask;   // general stage cost (M::has_cost): Hessian structure over [s; u; r], (f + nx)^2 bytes (device)
  const double *h_lok, *h_hik;  // optional per-frame path-constraint bounds [N * nh] (device; NULL = h_lo, h_hi on every frame)
};

template <class M>
__global__ void __launch_bounds__(256) stage_eval_kernel(StageDev sd, int batch, const double *__restrict__ p, const double *__restrict__ x,
                                                         const double *__restrict__ lbx, const double *__restrict__ ubx,
                                                         const double *__restrict__ lbg, const double *__restrict__ ubg,
                                                         double *__restrict__ P, double *__restrict__ q, double *__restrict__ A,
                                                         double *__restrict__ l, double *__restrict__ u) {
  constexpr int nx = M::nx, nu = M::nu, f = nx + nu;
  // cooperative functors (sm_has_coop): the f lanes of a stage sit in one wave at a multiple of f -- the parameter columns get a group of f thread
  // slots of their own (nx of them used), every frame the next f; threads per instance f (N + 1) instead of n
  constexpr bool COOP = sm_has_coop<M>::value && (64 % f == 0) && nx <= f;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int n = sd.n, N = sd.N;
  const int per = COOP ? f * (N + 1) : n;
  if (gid >= (long)batch * per) return;
  const int b = (int)(gid / per), jt = (int)(gid - (long)b * per);
  if (COOP && jt < f && jt >= nx) return;       // (padding lanes of the parameter group)
  const int j = COOP ? (jt < f ? jt : jt - f + nx) : jt;
  const double *pb = p + (long)b * nx, *xb = x + (long)b * sd.nvar;
  double *Pc = P + (long)b * sd.nnzP + sd.Pp[j], *Ac = A + (long)b * sd.nnzA + sd.Ap[j];
  double *qb = q + (long)b * n, *lb = l + (long)b * sd.m, *ub = u + (long)b * sd.m;
  if (j < nx) {
    const double pi = pb[j];
    if constexpr (M::has_cost) {
      // general stage cost: column p_i of the Hessian = sum over the frames of d(grad l_k)/dr_i (dual numbers through the
      // generated gradient); the rows p_r accumulate over k, the rows of frame k are written as they come
      constexpr int nl = f + nx;
      const unsigned char *mk = sd.hmask + f + j;
      int e = 0;
#pragma unroll
      for (int r = 0; r < nx; r++) e += mk[(f + r) * nl] ? 1 : 0;
      double acc[nx], qa = 0.0;
#pragma unroll
      for (int r = 0; r < nx; r++) acc[r] = 0.0;
      Dual s[nx], uu[nu], rr[nx], g[nl];
#pragma unroll
      for (int i = 0; i < nx; i++) rr[i] = {pb[i], i == j ? 1.0 : 0.0};
      for (int k = 0; k < N; k++) {
        const double *fr = xb + k * f;
#pragma unroll
        for (int i = 0; i < nx; i++) s[i] = {fr[i], 0.0};
#pragma unroll
        for (int i = 0; i < nu; i++) uu[i] = {fr[nx + i], 0.0};
        if (M::has_term && k == N - 1) M::template LTG<Dual>(s, uu, rr, g); else M::template LG<Dual>(s, uu, rr, g);
        double gv = 0.0;
#pragma unroll
        for (int r = 0; r < nx; r++) { acc[r] += g[f + r].d; gv = r == j ? g[f + r].v : gv; }
        qa += gv;
#pragma unroll
        for (int r = 0; r < f; r++) if (mk[r * nl]) Pc[e++] = g[r].d;
      }
      e = 0;
#pragma unroll
      for (int r = 0; r < nx; r++) if (mk[(f + r) * nl]) Pc[e++] = acc[r];
      qb[j] = qa;
    } else if (sd.Qk) {   // per-frame weights: d2f/dp_i2 = 2 sum_k Q_k,i
      // column p_i: H = d2f/dp_i2 = 2 N Q_i, d2f/dp_i ds_k[i] = -2 Q_i; grad = -2 Q_i sum_k (s_k[i] - p_i); rows l = u = p - p
      double e = 0.0, qs = 0.0;
      for (int k = 0; k < N; k++) { const double Qi = sd.Qk[k * nx + j]; qs += Qi; Pc[1 + k] = -2.0 * Qi; e += (xb[k * f + j] - pi) * Qi; }
      Pc[0] = 2.0 * qs;
      qb[j] = -2.0 * e;
    } else {
      double e = 0.0;
      const double Qi = sd.Q[j];
      Pc[0] = 2.0 * N * Qi;
      for (int k = 0; k < N; k++) { Pc[1 + k] = -2.0 * Qi; e += (xb[k * f + j] - pi) * Qi; }
      qb[j] = -2.0 * e;
    }
    Ac[0] = 1.0;
    lb[j] = pi - pi; ub[j] = pi - pi;
    return;
  }
  const int jj = j - nx, k = jj / f, c = jj - k * f;
  const double *fr = xb + k * f;
  const double xv = fr[c];
  Dual s[nx], uu[nu];
#pragma unroll
  for (int i = 0; i < nx; i++) s[i] = {fr[i], i == c ? 1.0 : 0.0};
#pragma unroll
  for (int i = 0; i < nu; i++) uu[i] = {fr[nx + i], nx + i == c ? 1.0 : 0.0};
  if constexpr (M::has_cost) {
    // column frame_k[c] of the Hessian of l_k: the dual parts of the generated gradient, rows p first, then the frame's
    constexpr int nl = f + nx;
    const unsigned char *mk = sd.hmask + c;
    Dual rr[nx], g[nl];
#pragma unroll
    for (int i = 0; i < nx; i++) rr[i] = {pb[i], 0.0};
    if (M::has_term && k == N - 1) M::template LTG<Dual>(s, uu, rr, g); else M::template LG<Dual>(s, uu, rr, g);
    int e = 0;
    double gv = 0.0;
#pragma unroll
    for (int i = 0; i < nx; i++) if (mk[(f + i) * nl]) Pc[e++] = g[f + i].d;
#pragma unroll
    for (int r = 0; r < f; r++) { if (mk[r * nl]) Pc[e++] = g[r].d; gv = r == c ? g[r].v : gv; }
    qb[j] = gv;
  } else if (c < nx) {
    const double Qc = sd.Qk ? sd.Qk[k * nx + c] : sd.Q[c];
    Pc[0] = -2.0 * Qc; Pc[1] = 2.0 * Qc;
    qb[j] = 2.0 * (xv - pb[c]) * Qc;
  } else {
    const double Rc = sd.Rk ? sd.Rk[k * nu + c - nx] : sd.R[c - nx];
    Pc[0] = 2.0 * Rc;
    qb[j] = 2.0 * xv * Rc;
  }
  lb[j] = lbx[(long)b * sd.nvar + jj] - xv; ub[j] = ubx[(long)b * sd.nvar + jj] - xv;
  int a = 0;
  Ac[a++] = 1.0;
  if (k >= 1 && c < nx) Ac[a++] = 1.0;
  if (k < N - 1) {
    Dual out[nx];
    if constexpr (COOP) M::Fc(sd.par, sd.dt, s, uu, out, c); else M::template F<Dual>(sd.par, sd.dt, s, uu, out);
#pragma unroll
    for (int r = 0; r < nx; r++) Ac[a + r] = -out[r].d;
    a += nx;
    if (c < nx) {
      double Fc = 0.0;
#pragma unroll
      for (int r = 0; r < nx; r++) Fc = r == c ? out[r].v : Fc;
      const double g = fr[f + c] - Fc;
      const int row = n + k * nx + c; const long gi = (long)b * sd.ng + k * nx + c;
      lb[row] = lbg[gi] - g; ub[row] = ubg[gi] - g;
    }
  }
  if constexpr (M::nh > 0) {
    // path constraint rows of this frame: column c of +dh/d[s; u]; lane c < nh also owns the shifted bounds of row h_k[c]
    constexpr int nh = M::nh;
    Dual hv[nh];
    M::template H<Dual>(s, uu, hv);
#pragma unroll
    for (int r = 0; r < nh; r++) Ac[a + r] = hv[r].d;
    a += nh;
    for (int r0 = c; r0 < nh; r0 += f) {      // nh may exceed the frame size: lane c takes rows c, c + f, ...
      double hc = 0.0;
#pragma unroll
      for (int r = 0; r < nh; r++) hc = r == r0 ? hv[r].v : hc;
      const int row = n + sd.ngd + k * nh + r0; const long gi = (long)b * sd.ng + sd.ngd + k * nh + r0;
      lb[row] = lbg[gi] - hc; ub[row] = ubg[gi] - hc;
    }
  }
  if constexpr (M::nk > 0) {
    // link constraint r_k = K(frame_k, frame_{k+1}) (rate limits u_{k+1} - u_k and the like): this column takes part in r_{k-1} as
    // the second frame and in r_k as the first; lane c < nk of frame k < N - 1 also owns the shifted bounds of row r_k[c]
    constexpr int nk = M::nk;
    const int g0 = sd.ngd + N * sd.nh;
    Dual os[nx], ou[nu], kv[nk];
    if (k >= 1) {
      const double *pf = fr - f;
#pragma unroll
      for (int i = 0; i < nx; i++) os[i] = {pf[i], 0.0};
#pragma unroll
      for (int i = 0; i < nu; i++) ou[i] = {pf[nx + i], 0.0};
      M::template K<Dual>(os, ou, s, uu, kv);
#pragma unroll
      for (int r = 0; r < nk; r++) Ac[a + r] = kv[r].d;
      a += nk;
    }
    if (k < N - 1) {
      const double *nf = fr + f;
#pragma unroll
      for (int i = 0; i < nx; i++) os[i] = {nf[i], 0.0};
#pragma unroll
      for (int i = 0; i < nu; i++) ou[i] = {nf[nx + i], 0.0};
      M::template K<Dual>(s, uu, os, ou, kv);
#pragma unroll
      for (int r = 0; r < nk; r++) Ac[a + r] = kv[r].d;
      for (int r0 = c; r0 < nk; r0 += f) {
        double kc = 0.0;
#pragma unroll
        for (int r = 0; r < nk; r++) kc = r == r0 ? kv[r].v : kc;
        const int row = n + g0 + k * nk + r0; const long gi = (long)b * sd.ng + g0 + k * nk + r0;
        lb[row] = lbg[gi] - kc; ub[row] = ubg[gi] - kc;
      }
    }
  }
}

// one wave per instance: lanes stride over the frames, butterfly reduction (fixed order => deterministic)
template <class M>
__global__ void __launch_bounds__(256) stage_merit_kernel(StageDev sd, int batch, const double *__restrict__ p, const double *__restrict__ x,
                                                          double *__restrict__ fout, double *__restrict__ gout) {
  constexpr int nx = M::nx, nu = M::nu, f = nx + nu;
  const int lane = threadIdx.x & 63, b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= batch) return;
  const double *pb = p + (long)b * nx, *xb = x + (long)b * sd.nvar;
  double cost = 0.0, gmax = 0.0;
  for (int k = lane; k < sd.N; k += 64) {
    const double *fr = xb + k * f;
    double s[nx], uu[nu];
#pragma unroll
    for (int i = 0; i < nx; i++) s[i] = fr[i];
#pragma unroll
    for (int i = 0; i < nu; i++) uu[i] = fr[nx + i];
    if constexpr (M::has_cost) {
      double rr[nx], lv[1];
#pragma unroll
      for (int i = 0; i < nx; i++) rr[i] = pb[i];
      if (M::has_term && k == sd.N - 1) M::template LT<double>(s, uu, rr, lv); else M::template L<double>(s, uu, rr, lv);
      cost += lv[0];
    } else {
#pragma unroll
      for (int i = 0; i < nx; i++) { const double e = s[i] - pb[i]; cost += e * e * (sd.Qk ? sd.Qk[k * nx + i] : sd.Q[i]); }
#pragma unroll
      for (int i = 0; i < nu; i++) cost += uu[i] * uu[i] * (sd.Rk ? sd.Rk[k * nu + i] : sd.R[i]);
    }
    if (k < sd.N - 1) {
      double out[nx];
      M::template F<double>(sd.par, sd.dt, s, uu, out);
#pragma unroll
      for (int i = 0; i < nx; i++) gmax = fmax(gmax, fabs(fr[f + i] - out[i]));
    }
    if constexpr (M::nh > 0) {
      double hv[M::nh];
      M::template H<double>(s, uu, hv);
#pragma unroll
      for (int i = 0; i < M::nh; i++) {
        const double lo = sd.h_lok ? sd.h_lok[k * M::nh + i] : sd.h_lo[i], hi = sd.h_hik ? sd.h_hik[k * M::nh + i] : sd.h_hi[i];
        gmax = fmax(gmax, fmax(lo - hv[i], hv[i] - hi));
      }
    }
    if constexpr (M::nk > 0) {
      if (k < sd.N - 1) {
        double ns[nx], nun[nu], kv[M::nk];
#pragma unroll
        for (int i = 0; i < nx; i++) ns[i] = fr[f + i];
#pragma unroll
        for (int i = 0; i < nu; i++) nun[i] = fr[f + nx + i];
        M::template K<double>(s, uu, ns, nun, kv);
#pragma unroll
        for (int i = 0; i < M::nk; i++) gmax = fmax(gmax, fmax(sd.k_lo[i] - kv[i], kv[i] - sd.k_hi[i]));
      }
    }
  }
  for (int o = 32; o >= 1; o >>= 1) { cost += __shfl_xor(cost, o, 64); gmax = fmax(gmax, __shfl_xor(gmax, o, 64)); }
  if (lane == 0) { if (fout) fout[b] = cost; if (gout) gout[b] = gmax; }
}


// launchers shared by the zoo dispatch and generated libraries
template <class M>
inline hipError_t stage_launch_eval(const StageDev &sd, int batch, const double *p, const double *x, const double *lbx, const double *ubx,
                                    const double *lbg, const double *ubg, double *P, double *q, double *A, double *l, double *u, hipStream_t st) {
  constexpr int f = M::nx + M::nu;
  const long threads = (long)batch * ((sm_has_coop<M>::value && (64 % f == 0) && M::nx <= f) ? f * (sd.N + 1) : sd.n);
  stage_eval_kernel<M><<<(unsigned)((threads + 255) / 256), 256, 0, st>>>(sd, batch, p, x, lbx, ubx, lbg, ubg, P, q, A, l, u);
  return hipGetLastError();
}
template <class M>
inline hipError_t stage_launch_merit(const StageDev &sd, int batch, const double *p, const double *x, double *f, double *gmax, hipStream_t st) {
  stage_merit_kernel<M><<<(unsigned)((batch + 3) / 4), 256, 0, st>>>(sd, batch, p, x, f, gmax);
  return hipGetLastError();
}
