// stage_eval.hip -- local-system evaluation on device for the stage-OCP model zoo (include/mpcqp.h, "Local-system
// evaluation on device").  Replaces SQPOptimizationSolver::getLocalSystem
// (reference src/sqp_solver/SQPOptimizationSolver.cpp:100-120) for a batch of instances.
//
// Mapping: one thread per (instance b, QP column j).  A thread owns everything indexed by its column of w = [p; x]:
// the CSC column of P, the CSC column of A = [I; dg/dw] (its identity entry, the +1 of s_{k+1} in g_k, and column c of
// -dF(s_k, u_k), obtained by running F on dual numbers seeded in direction c), q[j], the identity row's bounds
// l[j], u[j], and -- for state columns -- the shifted bounds of dynamics row g_k[c].  Adjacent lanes own adjacent
// columns, so every store stream is contiguous across a wave; the value part of F is recomputed by the f lanes of a
// stage (cheaper than exchanging it).  HBM-bound: one pass, algorithmic bytes = inputs + outputs.
#include <hip/hip_runtime.h>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/mpcqp.h"
#include "common.hpp"
#include "stage_models.hpp"

struct StageDev {
  int model, N, nx, nu, f, np, n, m, ng, nvar, nnzP, nnzA;
  double dt;
  double Q[SM_MAXNX], R[SM_MAXNU], par[SM_NPAR];
  const int *Pp, *Ap;   // device copies of the column pointers
};

struct mpcqp_stage {
  mpcqp_stage_desc desc;
  StageDev sd;
  int device = 0;
  std::vector<int> Pp, Pi, Ap, Ai;
  int *dPp = nullptr, *dAp = nullptr;
};

template <class M>
__global__ void __launch_bounds__(256) stage_eval_kernel(StageDev sd, int batch, const double *__restrict__ p, const double *__restrict__ x,
                                                         const double *__restrict__ lbx, const double *__restrict__ ubx,
                                                         const double *__restrict__ lbg, const double *__restrict__ ubg,
                                                         double *__restrict__ P, double *__restrict__ q, double *__restrict__ A,
                                                         double *__restrict__ l, double *__restrict__ u) {
  constexpr int nx = M::nx, nu = M::nu, f = nx + nu;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int n = sd.n, N = sd.N;
  if (gid >= (long)batch * n) return;
  const int b = (int)(gid / n), j = (int)(gid - (long)b * n);
  const double *pb = p + (long)b * nx, *xb = x + (long)b * sd.nvar;
  double *Pc = P + (long)b * sd.nnzP + sd.Pp[j], *Ac = A + (long)b * sd.nnzA + sd.Ap[j];
  double *qb = q + (long)b * n, *lb = l + (long)b * sd.m, *ub = u + (long)b * sd.m;
  if (j < nx) {
    // column p_i: H = d2f/dp_i2 = 2 N Q_i, d2f/dp_i ds_k[i] = -2 Q_i; grad = -2 Q_i sum_k (s_k[i] - p_i); rows l = u = p - p
    const double Qi = sd.Q[j], pi = pb[j];
    double e = 0.0;
    Pc[0] = 2.0 * N * Qi;
    for (int k = 0; k < N; k++) { Pc[1 + k] = -2.0 * Qi; e += (xb[k * f + j] - pi) * Qi; }
    qb[j] = -2.0 * e;
    Ac[0] = 1.0;
    lb[j] = pi - pi; ub[j] = pi - pi;
    return;
  }
  const int jj = j - nx, k = jj / f, c = jj - k * f;
  const double *fr = xb + k * f;
  const double xv = fr[c];
  if (c < nx) {
    const double Qc = sd.Q[c];
    Pc[0] = -2.0 * Qc; Pc[1] = 2.0 * Qc;
    qb[j] = 2.0 * (xv - pb[c]) * Qc;
  } else {
    const double Rc = sd.R[c - nx];
    Pc[0] = 2.0 * Rc;
    qb[j] = 2.0 * xv * Rc;
  }
  lb[j] = lbx[(long)b * sd.nvar + jj] - xv; ub[j] = ubx[(long)b * sd.nvar + jj] - xv;
  int a = 0;
  Ac[a++] = 1.0;
  if (k >= 1 && c < nx) Ac[a++] = 1.0;
  if (k < N - 1) {
    Dual s[nx], uu[nu], out[nx];
#pragma unroll
    for (int i = 0; i < nx; i++) s[i] = {fr[i], i == c ? 1.0 : 0.0};
#pragma unroll
    for (int i = 0; i < nu; i++) uu[i] = {fr[nx + i], nx + i == c ? 1.0 : 0.0};
    M::template F<Dual>(sd.par, sd.dt, s, uu, out);
#pragma unroll
    for (int r = 0; r < nx; r++) Ac[a + r] = -out[r].d;
    if (c < nx) {
      double Fc = 0.0;
#pragma unroll
      for (int r = 0; r < nx; r++) Fc = r == c ? out[r].v : Fc;
      const double g = fr[f + c] - Fc;
      const int row = n + k * nx + c; const long gi = (long)b * sd.ng + k * nx + c;
      lb[row] = lbg[gi] - g; ub[row] = ubg[gi] - g;
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
    for (int i = 0; i < nx; i++) { s[i] = fr[i]; const double e = s[i] - pb[i]; cost += e * e * sd.Q[i]; }
#pragma unroll
    for (int i = 0; i < nu; i++) { uu[i] = fr[nx + i]; cost += uu[i] * uu[i] * sd.R[i]; }
    if (k < sd.N - 1) {
      double out[nx];
      M::template F<double>(sd.par, sd.dt, s, uu, out);
#pragma unroll
      for (int i = 0; i < nx; i++) gmax = fmax(gmax, fabs(fr[f + i] - out[i]));
    }
  }
  for (int o = 32; o >= 1; o >>= 1) { cost += __shfl_xor(cost, o, 64); gmax = fmax(gmax, __shfl_xor(gmax, o, 64)); }
  if (lane == 0) { if (fout) fout[b] = cost; if (gout) gout[b] = gmax; }
}

__global__ void __launch_bounds__(256) stage_step_kernel(int batch, int nvar, int n, int np, double alpha, const double *__restrict__ dw,
                                                         double *__restrict__ x, double *__restrict__ step_max, const int *__restrict__ status) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= batch) return;
  double mx = 0.0;
  if (status) {
    const int s = status[b];
    if (s != MPCQP_SOLVED && s != MPCQP_SOLVED_INACCURATE && s != MPCQP_MAX_ITER_REACHED) { if (step_max && lane == 0) step_max[b] = 0.0; return; }
  }
  for (int i = lane; i < nvar; i += 64) {
    const double d = alpha * dw[(long)b * n + np + i];
    x[(long)b * nvar + i] += d; mx = fmax(mx, fabs(d));
  }
  for (int o = 32; o >= 1; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
  if (step_max && lane == 0) step_max[b] = mx;
}

// ------------------------------------------------------------------------------------------ host side
extern "C" {

int mpcqp_stage_default(int model, int horizon, mpcqp_stage_desc *d) {
  if (!d) return mpcqp_set_error(MPCQP_ERR_ARG, "desc is null");
  if (model < 0 || model >= SM_NMODELS) return mpcqp_set_error(MPCQP_ERR_ARG, "unknown model");
  *d = mpcqp_stage_desc();
  d->model = model; d->horizon = horizon; d->device = -1;
  switch (model) {
    case SM_DOUBLE_INTEGRATOR:
      d->dt = 0.05; d->Q[0] = 10.0; d->Q[1] = 1.0; d->R[0] = 0.1; break;
    case SM_QUADROTOR: {
      d->dt = 0.02;
      const double Qd[12] = {10, 10, 10, 1, 1, 1, 1, 1, 1, 0.1, 0.1, 0.1};
      for (int i = 0; i < 12; i++) d->Q[i] = Qd[i];
      for (int i = 0; i < 4; i++) d->R[i] = 0.1;
      const double pr[7] = {1.0, 9.81, 0.2, 0.05, 0.01, 0.01, 0.02};
      for (int i = 0; i < 7; i++) d->par[i] = pr[i];
      break;
    }
    case SM_CARTPOLE: {
      d->dt = 0.02;
      const double Qd[4] = {1.0, 10.0, 0.1, 0.1};
      for (int i = 0; i < 4; i++) d->Q[i] = Qd[i];
      d->R[0] = 0.01;
      const double pr[4] = {1.0, 0.1, 0.5, 9.81};
      for (int i = 0; i < 4; i++) d->par[i] = pr[i];
      break;
    }
  }
  return MPCQP_OK;
}

int mpcqp_stage_create(const mpcqp_stage_desc *d, mpcqp_stage **out) {
  if (!out) return mpcqp_set_error(MPCQP_ERR_ARG, "out is null");
  *out = nullptr;
  if (!d) return mpcqp_set_error(MPCQP_ERR_ARG, "desc is null");
  if (d->model < 0 || d->model >= SM_NMODELS) return mpcqp_set_error(MPCQP_ERR_ARG, "unknown model");
  if (d->horizon < 2 || !(d->dt > 0.0)) return mpcqp_set_error(MPCQP_ERR_ARG, "horizon must be >= 2 and dt > 0");
  int dev = 0;
  if (int rc = mpcqp_pick_device(d->device, &dev)) return rc;
  mpcqp_stage *s = new mpcqp_stage();
  s->desc = *d; s->device = dev;
  StageDev &sd = s->sd;
  sd.model = d->model; sd.N = d->horizon; sd.dt = d->dt;
  sm_model_dims(d->model, &sd.nx, &sd.nu);
  sd.f = sd.nx + sd.nu; sd.np = sd.nx; sd.nvar = sd.N * sd.f; sd.n = sd.np + sd.nvar;
  sd.ng = (sd.N - 1) * sd.nx; sd.m = sd.n + sd.ng;
  for (int i = 0; i < SM_MAXNX; i++) sd.Q[i] = d->Q[i];
  for (int i = 0; i < SM_MAXNU; i++) sd.R[i] = d->R[i];
  for (int i = 0; i < SM_NPAR; i++) sd.par[i] = d->par[i];
  sm_build_pattern(sd.nx, sd.nu, sd.N, s->Pp, s->Pi, s->Ap, s->Ai);
  sd.nnzP = (int)s->Pi.size(); sd.nnzA = (int)s->Ai.size();
  auto bail = [&](int rc) { mpcqp_stage_destroy(s); return rc; };
  if (hipSetDevice(dev) != hipSuccess) return bail(mpcqp_set_error(MPCQP_ERR_HIP, "hipSetDevice failed"));
  const size_t bytes = (size_t)(sd.n + 1) * sizeof(int);
  if (hipMalloc(&s->dPp, bytes) != hipSuccess || hipMalloc(&s->dAp, bytes) != hipSuccess)
    return bail(mpcqp_set_error(MPCQP_ERR_HIP, "hipMalloc of the column pointers failed"));
  if (hipMemcpy(s->dPp, s->Pp.data(), bytes, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(s->dAp, s->Ap.data(), bytes, hipMemcpyHostToDevice) != hipSuccess)
    return bail(mpcqp_set_error(MPCQP_ERR_HIP, "upload of the column pointers failed"));
  sd.Pp = s->dPp; sd.Ap = s->dAp;
  *out = s;
  return MPCQP_OK;
}

void mpcqp_stage_destroy(mpcqp_stage *s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->dPp) (void)hipFree(s->dPp);
  if (s->dAp) (void)hipFree(s->dAp);
  delete s;
}

int mpcqp_stage_dims(const mpcqp_stage *s, int *o) {
  if (!s || !o) return mpcqp_set_error(MPCQP_ERR_ARG, "null argument");
  const StageDev &d = s->sd;
  o[0] = d.nx; o[1] = d.nu; o[2] = d.np; o[3] = d.n; o[4] = d.m; o[5] = d.nnzP; o[6] = d.nnzA; o[7] = d.nvar;
  return MPCQP_OK;
}

int mpcqp_stage_pattern(const mpcqp_stage *s, int *Pp, int *Pi, int *Ap, int *Ai) {
  if (!s || !Pp || !Pi || !Ap || !Ai) return mpcqp_set_error(MPCQP_ERR_ARG, "null argument");
  std::copy(s->Pp.begin(), s->Pp.end(), Pp); std::copy(s->Pi.begin(), s->Pi.end(), Pi);
  std::copy(s->Ap.begin(), s->Ap.end(), Ap); std::copy(s->Ai.begin(), s->Ai.end(), Ai);
  return MPCQP_OK;
}

#define STAGE_DISPATCH(KERNEL, GRID, ...)                                                                               \
  switch (s->sd.model) {                                                                                                \
    case SM_DOUBLE_INTEGRATOR: KERNEL<SmDoubleIntegrator><<<GRID, 256, 0, st>>>(__VA_ARGS__); break;                    \
    case SM_QUADROTOR: KERNEL<SmQuadrotor><<<GRID, 256, 0, st>>>(__VA_ARGS__); break;                                   \
    case SM_CARTPOLE: KERNEL<SmCartPole><<<GRID, 256, 0, st>>>(__VA_ARGS__); break;                                     \
  }

int mpcqp_stage_eval(mpcqp_stage *s, int batch, const double *p, const double *x, const double *lbx, const double *ubx,
                     const double *lbg, const double *ubg, double *P, double *q, double *A, double *l, double *u, void *stream) {
  if (!s) return mpcqp_set_error(MPCQP_ERR_ARG, "stage handle is null");
  if (batch <= 0) return mpcqp_set_error(MPCQP_ERR_ARG, "batch must be positive");
  if (!p || !x || !lbx || !ubx || !lbg || !ubg || !P || !q || !A || !l || !u) return mpcqp_set_error(MPCQP_ERR_ARG, "null data pointer");
  MPCQP_HIPCHK(hipSetDevice(s->device));
  hipStream_t st = (hipStream_t)stream;
  const long threads = (long)batch * s->sd.n;
  const unsigned grid = (unsigned)((threads + 255) / 256);
  STAGE_DISPATCH(stage_eval_kernel, grid, s->sd, batch, p, x, lbx, ubx, lbg, ubg, P, q, A, l, u)
  MPCQP_HIPCHK(hipGetLastError());
  return MPCQP_OK;
}

int mpcqp_stage_merit(mpcqp_stage *s, int batch, const double *p, const double *x, double *f, double *gmax, void *stream) {
  if (!s) return mpcqp_set_error(MPCQP_ERR_ARG, "stage handle is null");
  if (batch <= 0 || !p || !x) return mpcqp_set_error(MPCQP_ERR_ARG, "bad batch or null data pointer");
  MPCQP_HIPCHK(hipSetDevice(s->device));
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)((batch + 3) / 4);
  STAGE_DISPATCH(stage_merit_kernel, grid, s->sd, batch, p, x, f, gmax)
  MPCQP_HIPCHK(hipGetLastError());
  return MPCQP_OK;
}

int mpcqp_stage_step(mpcqp_stage *s, int batch, double alpha, const double *dw, double *x, double *step_max, const int *status, void *stream) {
  if (!s) return mpcqp_set_error(MPCQP_ERR_ARG, "stage handle is null");
  if (batch <= 0 || !dw || !x) return mpcqp_set_error(MPCQP_ERR_ARG, "bad batch or null data pointer");
  MPCQP_HIPCHK(hipSetDevice(s->device));
  hipStream_t st = (hipStream_t)stream;
  stage_step_kernel<<<(unsigned)((batch + 3) / 4), 256, 0, st>>>(batch, s->sd.nvar, s->sd.n, s->sd.np, alpha, dw, x, step_max, status);
  MPCQP_HIPCHK(hipGetLastError());
  return MPCQP_OK;
}

}  // extern "C"
