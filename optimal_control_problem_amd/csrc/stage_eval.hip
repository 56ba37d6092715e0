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
#include <dlfcn.h>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/mpcqp.h"
#include "common.hpp"
#include "stage_kernels.hpp"

typedef int (*user_eval_fn)(const StageDev *, int, const double *, const double *, const double *, const double *, const double *,
                            const double *, double *, double *, double *, double *, double *, void *);
typedef int (*user_merit_fn)(const StageDev *, int, const double *, const double *, double *, double *, void *);

struct mpcqp_stage {
  mpcqp_stage_desc desc;
  StageDev sd;
  int device = 0;
  std::vector<int> Pp, Pi, Ap, Ai;
  int *dPp = nullptr, *dAp = nullptr;
  double *dQk = nullptr, *dRk = nullptr;
  double *dhlo = nullptr, *dhhi = nullptr;   // per-frame path bounds (mpcqp_stage_set_path_bounds)
  unsigned char *dmask = nullptr;     // Hessian structure of a generated general stage cost
  void *user_lib = nullptr;           // dlopen handle of a generated dynamics library (model == MPCQP_MODEL_USER)
  user_eval_fn user_eval = nullptr;
  user_merit_fn user_merit = nullptr;
  bool general_cost = false;          // the library carries its own stage cost: Q, R and mpcqp_stage_set_weights do not apply
};

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

static int stage_create_common(const mpcqp_stage_desc *d, int nx, int nu, int nh, const double *h_lo, const double *h_hi, mpcqp_stage *s,
                               const unsigned char *cost_mask = nullptr, int nk = 0, const double *k_lo = nullptr, const double *k_hi = nullptr) {
  int dev = 0;
  if (int rc = mpcqp_pick_device(d->device, &dev)) return rc;
  s->desc = *d; s->device = dev;
  StageDev &sd = s->sd;
  sd.model = d->model; sd.N = d->horizon; sd.dt = d->dt; sd.nx = nx; sd.nu = nu;
  sd.f = sd.nx + sd.nu; sd.np = sd.nx; sd.nvar = sd.N * sd.f; sd.n = sd.np + sd.nvar;
  sd.nh = nh; sd.nk = nk; sd.ngd = (sd.N - 1) * sd.nx; sd.ng = sd.ngd + sd.N * nh + (sd.N - 1) * nk; sd.m = sd.n + sd.ng;
  for (int i = 0; i < SM_MAXNK; i++) { sd.k_lo[i] = (k_lo && i < nk) ? k_lo[i] : -INFINITY; sd.k_hi[i] = (k_hi && i < nk) ? k_hi[i] : INFINITY; }
  for (int i = 0; i < SM_MAXNH; i++) { sd.h_lo[i] = (h_lo && i < nh) ? h_lo[i] : -INFINITY; sd.h_hi[i] = (h_hi && i < nh) ? h_hi[i] : INFINITY; }
  for (int i = 0; i < SM_MAXNX; i++) sd.Q[i] = d->Q[i];
  for (int i = 0; i < SM_MAXNU; i++) sd.R[i] = d->R[i];
  for (int i = 0; i < SM_NPAR; i++) sd.par[i] = d->par[i];
  sm_build_pattern(sd.nx, sd.nu, sd.N, sd.nh, sd.nk, s->Pp, s->Pi, s->Ap, s->Ai);
  if (cost_mask) sm_build_cost_pattern(sd.nx, sd.nu, sd.N, cost_mask, s->Pp, s->Pi);
  sd.nnzP = (int)s->Pi.size(); sd.nnzA = (int)s->Ai.size();
  if (hipSetDevice(dev) != hipSuccess) return mpcqp_set_error(MPCQP_ERR_HIP, "hipSetDevice failed");
  const size_t bytes = (size_t)(sd.n + 1) * sizeof(int);
  if (hipMalloc(&s->dPp, bytes) != hipSuccess || hipMalloc(&s->dAp, bytes) != hipSuccess)
    return mpcqp_set_error(MPCQP_ERR_HIP, "hipMalloc of the column pointers failed");
  if (hipMemcpy(s->dPp, s->Pp.data(), bytes, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(s->dAp, s->Ap.data(), bytes, hipMemcpyHostToDevice) != hipSuccess)
    return mpcqp_set_error(MPCQP_ERR_HIP, "upload of the column pointers failed");
  sd.Pp = s->dPp; sd.Ap = s->dAp; sd.Qk = nullptr; sd.Rk = nullptr; sd.hmask = nullptr; sd.h_lok = nullptr; sd.h_hik = nullptr;
  if (cost_mask) {
    const size_t mb = (size_t)(sd.f + sd.nx) * (sd.f + sd.nx);
    if (hipMalloc(&s->dmask, mb) != hipSuccess || hipMemcpy(s->dmask, cost_mask, mb, hipMemcpyHostToDevice) != hipSuccess)
      return mpcqp_set_error(MPCQP_ERR_HIP, "upload of the cost structure failed");
    sd.hmask = s->dmask;
  }
  return MPCQP_OK;
}

int mpcqp_stage_create(const mpcqp_stage_desc *d, mpcqp_stage **out) {
  if (!out) return mpcqp_set_error(MPCQP_ERR_ARG, "out is null");
  *out = nullptr;
  if (!d) return mpcqp_set_error(MPCQP_ERR_ARG, "desc is null");
  if (d->model < 0 || d->model >= SM_NMODELS) return mpcqp_set_error(MPCQP_ERR_ARG, "unknown model");
  if (d->horizon < 2 || !(d->dt > 0.0)) return mpcqp_set_error(MPCQP_ERR_ARG, "horizon must be >= 2 and dt > 0");
  int nx, nu;
  sm_model_dims(d->model, &nx, &nu);
  mpcqp_stage *s = new mpcqp_stage();
  if (int rc = stage_create_common(d, nx, nu, 0, nullptr, nullptr, s)) { mpcqp_stage_destroy(s); return rc; }
  *out = s;
  return MPCQP_OK;
}

int mpcqp_stage_create_user(const mpcqp_stage_desc *d, const char *library_path, mpcqp_stage **out) {
  if (!out) return mpcqp_set_error(MPCQP_ERR_ARG, "out is null");
  *out = nullptr;
  if (!d || !library_path) return mpcqp_set_error(MPCQP_ERR_ARG, "null argument");
  if (d->horizon < 2 || !(d->dt > 0.0)) return mpcqp_set_error(MPCQP_ERR_ARG, "horizon must be >= 2 and dt > 0");
  void *lib = dlopen(library_path, RTLD_NOW | RTLD_LOCAL);
  if (!lib) return mpcqp_set_error(MPCQP_ERR_ARG, std::string("cannot load the dynamics library: ") + dlerror());
  auto abi = (int (*)())dlsym(lib, "mpcqp_user_abi");
  auto dims = (void (*)(int *, int *))dlsym(lib, "mpcqp_user_dims");
  auto ev = (user_eval_fn)dlsym(lib, "mpcqp_user_eval");
  auto me = (user_merit_fn)dlsym(lib, "mpcqp_user_merit");
  if (!abi || !dims || !ev || !me) { dlclose(lib); return mpcqp_set_error(MPCQP_ERR_ARG, "the library does not export mpcqp_user_abi/dims/eval/merit"); }
  if (abi() != STAGE_ABI_VERSION) { dlclose(lib); return mpcqp_set_error(MPCQP_ERR_ARG, "the library was generated for another version of the stage kernels; regenerate it"); }
  int nx = 0, nu = 0, nh = 0;
  dims(&nx, &nu);
  auto nhf = (int (*)())dlsym(lib, "mpcqp_user_nh");
  auto hb = (void (*)(double *, double *))dlsym(lib, "mpcqp_user_path_bounds");
  if (nhf) nh = nhf();
  if (nx <= 0 || nx > SM_MAXNX || nu <= 0 || nu > SM_MAXNU || nh < 0 || nh > SM_MAXNH || (nh > 0 && !hb)) {
    dlclose(lib); return mpcqp_set_error(MPCQP_ERR_LIMIT, "nx must be in 1..16, nu in 1..8 and the path constraint in 0..16 rows");
  }
  double h_lo[SM_MAXNH], h_hi[SM_MAXNH];
  if (nh > 0) hb(h_lo, h_hi);
  int nk = 0;
  auto nkf = (int (*)())dlsym(lib, "mpcqp_user_nk");
  auto kb = (void (*)(double *, double *))dlsym(lib, "mpcqp_user_link_bounds");
  if (nkf) nk = nkf();
  if (nk < 0 || nk > SM_MAXNK || (nk > 0 && !kb)) { dlclose(lib); return mpcqp_set_error(MPCQP_ERR_LIMIT, "the link constraint may have 0..8 rows"); }
  double k_lo[SM_MAXNK], k_hi[SM_MAXNK];
  if (nk > 0) kb(k_lo, k_hi);
  mpcqp_stage *s = new mpcqp_stage();
  s->user_lib = lib; s->user_eval = ev; s->user_merit = me;
  mpcqp_stage_desc dd = *d; dd.model = MPCQP_MODEL_USER;
  std::vector<unsigned char> mask((size_t)(2 * nx + nu) * (2 * nx + nu));
  auto cf = (int (*)(unsigned char *))dlsym(lib, "mpcqp_user_cost");
  s->general_cost = cf && cf(mask.data());
  if (int rc = stage_create_common(&dd, nx, nu, nh, h_lo, h_hi, s, s->general_cost ? mask.data() : nullptr, nk, k_lo, k_hi)) { mpcqp_stage_destroy(s); return rc; }
  *out = s;
  return MPCQP_OK;
}

void mpcqp_stage_destroy(mpcqp_stage *s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->dPp) (void)hipFree(s->dPp);
  if (s->dAp) (void)hipFree(s->dAp);
  if (s->dQk) (void)hipFree(s->dQk);
  if (s->dRk) (void)hipFree(s->dRk);
  if (s->dmask) (void)hipFree(s->dmask);
  if (s->dhlo) (void)hipFree(s->dhlo);
  if (s->dhhi) (void)hipFree(s->dhhi);
  if (s->user_lib) dlclose(s->user_lib);
  delete s;
}

int mpcqp_stage_set_weights(mpcqp_stage *s, const double *Qk, const double *Rk) {
  if (!s) return mpcqp_set_error(MPCQP_ERR_ARG, "stage handle is null");
  if ((Qk == nullptr) != (Rk == nullptr)) return mpcqp_set_error(MPCQP_ERR_ARG, "give both weight arrays or neither");
  if (s->general_cost) return mpcqp_set_error(MPCQP_ERR_ARG, "this evaluator was generated with its own stage cost; diagonal weights do not apply");
  MPCQP_HIPCHK(hipSetDevice(s->device));
  StageDev &sd = s->sd;
  if (!Qk) { sd.Qk = nullptr; sd.Rk = nullptr; return MPCQP_OK; }
  const size_t bq = (size_t)sd.N * sd.nx * sizeof(double), br = (size_t)sd.N * sd.nu * sizeof(double);
  if (!s->dQk) MPCQP_HIPCHK(hipMalloc(&s->dQk, bq));
  if (!s->dRk) MPCQP_HIPCHK(hipMalloc(&s->dRk, br));
  MPCQP_HIPCHK(hipMemcpy(s->dQk, Qk, bq, hipMemcpyHostToDevice));
  MPCQP_HIPCHK(hipMemcpy(s->dRk, Rk, br, hipMemcpyHostToDevice));
  sd.Qk = s->dQk; sd.Rk = s->dRk;
  return MPCQP_OK;
}

int mpcqp_stage_set_path_bounds(mpcqp_stage *s, const double *lo, const double *hi) {
  if (!s) return mpcqp_set_error(MPCQP_ERR_ARG, "stage handle is null");
  if ((lo == nullptr) != (hi == nullptr)) return mpcqp_set_error(MPCQP_ERR_ARG, "give both bound arrays or neither");
  StageDev &sd = s->sd;
  if (sd.nh == 0) return mpcqp_set_error(MPCQP_ERR_ARG, "this evaluator has no path constraint");
  MPCQP_HIPCHK(hipSetDevice(s->device));
  if (!lo) { sd.h_lok = nullptr; sd.h_hik = nullptr; return MPCQP_OK; }
  const size_t bytes = (size_t)sd.N * sd.nh * sizeof(double);
  if (!s->dhlo) MPCQP_HIPCHK(hipMalloc(&s->dhlo, bytes));
  if (!s->dhhi) MPCQP_HIPCHK(hipMalloc(&s->dhhi, bytes));
  MPCQP_HIPCHK(hipMemcpy(s->dhlo, lo, bytes, hipMemcpyHostToDevice));
  MPCQP_HIPCHK(hipMemcpy(s->dhhi, hi, bytes, hipMemcpyHostToDevice));
  sd.h_lok = s->dhlo; sd.h_hik = s->dhhi;
  return MPCQP_OK;
}

int mpcqp_stage_dims(const mpcqp_stage *s, int *o) {
  if (!s || !o) return mpcqp_set_error(MPCQP_ERR_ARG, "null argument");
  const StageDev &d = s->sd;
  o[0] = d.nx; o[1] = d.nu; o[2] = d.np; o[3] = d.n; o[4] = d.m; o[5] = d.nnzP; o[6] = d.nnzA; o[7] = d.nvar;
  return MPCQP_OK;
}

int mpcqp_stage_has_cost(const mpcqp_stage *s) { return s && s->general_cost ? 1 : 0; }

int mpcqp_stage_pattern(const mpcqp_stage *s, int *Pp, int *Pi, int *Ap, int *Ai) {
  if (!s || !Pp || !Pi || !Ap || !Ai) return mpcqp_set_error(MPCQP_ERR_ARG, "null argument");
  std::copy(s->Pp.begin(), s->Pp.end(), Pp); std::copy(s->Pi.begin(), s->Pi.end(), Pi);
  std::copy(s->Ap.begin(), s->Ap.end(), Ap); std::copy(s->Ai.begin(), s->Ai.end(), Ai);
  return MPCQP_OK;
}

int mpcqp_stage_eval(mpcqp_stage *s, int batch, const double *p, const double *x, const double *lbx, const double *ubx,
                     const double *lbg, const double *ubg, double *P, double *q, double *A, double *l, double *u, void *stream) {
  if (!s) return mpcqp_set_error(MPCQP_ERR_ARG, "stage handle is null");
  if (batch <= 0) return mpcqp_set_error(MPCQP_ERR_ARG, "batch must be positive");
  if (!p || !x || !lbx || !ubx || !lbg || !ubg || !P || !q || !A || !l || !u) return mpcqp_set_error(MPCQP_ERR_ARG, "null data pointer");
  MPCQP_HIPCHK(hipSetDevice(s->device));
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipSuccess;
  switch (s->sd.model) {
    case SM_DOUBLE_INTEGRATOR: e = stage_launch_eval<SmDoubleIntegrator>(s->sd, batch, p, x, lbx, ubx, lbg, ubg, P, q, A, l, u, st); break;
    case SM_QUADROTOR: e = stage_launch_eval<SmQuadrotor>(s->sd, batch, p, x, lbx, ubx, lbg, ubg, P, q, A, l, u, st); break;
    case SM_CARTPOLE: e = stage_launch_eval<SmCartPole>(s->sd, batch, p, x, lbx, ubx, lbg, ubg, P, q, A, l, u, st); break;
    case MPCQP_MODEL_USER: e = (hipError_t)s->user_eval(&s->sd, batch, p, x, lbx, ubx, lbg, ubg, P, q, A, l, u, stream); break;
  }
  MPCQP_HIPCHK(e);
  return MPCQP_OK;
}

int mpcqp_stage_merit(mpcqp_stage *s, int batch, const double *p, const double *x, double *f, double *gmax, void *stream) {
  if (!s) return mpcqp_set_error(MPCQP_ERR_ARG, "stage handle is null");
  if (batch <= 0 || !p || !x) return mpcqp_set_error(MPCQP_ERR_ARG, "bad batch or null data pointer");
  MPCQP_HIPCHK(hipSetDevice(s->device));
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipSuccess;
  switch (s->sd.model) {
    case SM_DOUBLE_INTEGRATOR: e = stage_launch_merit<SmDoubleIntegrator>(s->sd, batch, p, x, f, gmax, st); break;
    case SM_QUADROTOR: e = stage_launch_merit<SmQuadrotor>(s->sd, batch, p, x, f, gmax, st); break;
    case SM_CARTPOLE: e = stage_launch_merit<SmCartPole>(s->sd, batch, p, x, f, gmax, st); break;
    case MPCQP_MODEL_USER: e = (hipError_t)s->user_merit(&s->sd, batch, p, x, f, gmax, stream); break;
  }
  MPCQP_HIPCHK(e);
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
