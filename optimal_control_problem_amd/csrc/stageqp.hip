// stageqp.hip -- the structured stage form of the QP (include/mpcqp.h, "Structured stage form"): the caller hands
// over the stage blocks of an OCP-structured QP -- Hessian blocks per frame, their coupling to a leading parameter block, the
// dynamics blocks [A_k B_k] -- instead of CSC value arrays.  Host side: the CSC pattern in the reference's formulation (w = [p; frames],
// rows [p; frames; dynamics], reference src/sqp_solver/SQPOptimizationSolver.cpp:47-77; frames stage-interleaved as in
// src/OCP_config/OCPConfig.cpp:29-46,102) and, per CSC value slot, where its number comes from.  Device side: one gather kernel that
// writes the value arrays an ordinary handle (mpcqp_create on that pattern) borrows.  The ADMM kernels do not know about this form: a QP
// given in blocks and the same QP given in CSC run the same instance on the same numbers.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/mpcqp.h"
#include "common.hpp"

namespace {

// where a CSC value comes from: one of the caller's block arrays (offset inside an instance), or a constant
enum { SRC_H = 0, SRC_HP = 1, SRC_HPP = 2, SRC_AB_NEG = 3, SRC_ONE = 4 };
struct SqSrc { int kind, off; };

struct SqPattern {
  int N = 0, nx = 0, nu = 0, np = 0, f = 0, n = 0, m = 0;
  std::vector<int> Pp, Pi, Ap, Ai;
  std::vector<SqSrc> Ps, As;
  std::string error;
};

// rows ascending inside a column, both triangles of P (as CasADi hands the reference its Hessian, and as mpcqp_stage_pattern does):
//   P column p_i     : rows p_r with cost[f + r][f + i]; then per frame k the rows frame_k[r] with cost[r][f + i]   (value Hp_k[i][r])
//   P column frame_k[c]: rows p_i with cost[f + i][c] (value Hp_k[i][c]); rows frame_k[r] with cost[r][c] (value H_k[r][c])
//   A column p_i     : row p_i (1)
//   A column frame_k[c]: row np + k f + c (1); for k >= 1 and c < nx the +1 of s_k in dynamics row block k - 1; for k < N - 1 the rows
//                        r of block k with dyn[r][c] (value -[A_k B_k][r][c])
SqPattern build_pattern(const mpcqp_stageqp_dims *d) {
  SqPattern q;
  if (!d) { q.error = "dims is null"; return q; }
  q.N = d->N; q.nx = d->nx; q.nu = d->nu; q.np = d->np; q.f = d->nx + d->nu;
  if (d->N < 2 || d->nx < 1 || d->nu < 0 || d->np < 0) { q.error = "stage dimensions: N >= 2, nx >= 1, nu >= 0, np >= 0"; return q; }
  const long nl = q.f + q.np, nlong = (long)q.np + (long)q.N * q.f, mlong = nlong + (long)(q.N - 1) * q.nx;
  if (mlong > (1L << 24) || nl * nl * q.N > (1L << 28)) { q.error = "stage dimensions too large"; return q; }
  q.n = (int)nlong; q.m = (int)mlong;
  const int f = q.f, np = q.np, N = q.N, nx = q.nx;
  auto cost = [&](int r, int c) { return d->cost_mask ? d->cost_mask[(long)r * nl + c] != 0 : true; };
  auto dyn = [&](int r, int c) { return d->dyn_mask ? d->dyn_mask[(long)r * f + c] != 0 : true; };
  if (d->cost_mask) {
    for (int r = 0; r < nl; r++) {
      if (!cost(r, r)) { q.error = "cost_mask: the diagonal must be set"; return q; }
      for (int c = 0; c < r; c++) if (cost(r, c) != cost(c, r)) { q.error = "cost_mask must be symmetric"; return q; }
    }
  }
  q.Pp.assign(1, 0); q.Ap.assign(1, 0);
  for (int i = 0; i < np; i++) {
    for (int r = 0; r < np; r++) if (cost(f + r, f + i)) { q.Pi.push_back(r); q.Ps.push_back({SRC_HPP, r * np + i}); }
    for (int k = 0; k < N; k++)
      for (int r = 0; r < f; r++) if (cost(r, f + i)) { q.Pi.push_back(np + k * f + r); q.Ps.push_back({SRC_HP, (k * np + i) * f + r}); }
    q.Pp.push_back((int)q.Pi.size());
    q.Ai.push_back(i); q.As.push_back({SRC_ONE, 0}); q.Ap.push_back((int)q.Ai.size());
  }
  for (int k = 0; k < N; k++)
    for (int c = 0; c < f; c++) {
      const int j = np + k * f + c;
      for (int i = 0; i < np; i++) if (cost(f + i, c)) { q.Pi.push_back(i); q.Ps.push_back({SRC_HP, (k * np + i) * f + c}); }
      for (int r = 0; r < f; r++) if (cost(r, c)) { q.Pi.push_back(np + k * f + r); q.Ps.push_back({SRC_H, (k * f + r) * f + c}); }
      q.Pp.push_back((int)q.Pi.size());
      q.Ai.push_back(j); q.As.push_back({SRC_ONE, 0});
      if (k >= 1 && c < nx) { q.Ai.push_back(q.n + (k - 1) * nx + c); q.As.push_back({SRC_ONE, 0}); }
      if (k < N - 1) for (int r = 0; r < nx; r++) if (dyn(r, c)) { q.Ai.push_back(q.n + k * nx + r); q.As.push_back({SRC_AB_NEG, (k * nx + r) * f + c}); }
      q.Ap.push_back((int)q.Ai.size());
    }
  return q;
}

// one thread per (instance, CSC value): coalesced stores, gathered loads (the blocks of an instance are a few tens of KB: L2 serves the re-reads)
__global__ void __launch_bounds__(256) stageqp_pack_kernel(int batch, int nnzP, int nnzA, const int2 *__restrict__ srcP, const int2 *__restrict__ srcA,
                                                           const double *__restrict__ H, long sH, const double *__restrict__ Hp, long sHp,
                                                           const double *__restrict__ Hpp, long sHpp, const double *__restrict__ AB, long sAB,
                                                           double *__restrict__ P, double *__restrict__ A) {
  const long per = (long)nnzP + nnzA, gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= per * batch) return;
  const int b = (int)(gid / per), e = (int)(gid - (long)b * per);
  const bool isP = e < nnzP;
  const int2 s = isP ? srcP[e] : srcA[e - nnzP];
  double v;
  switch (s.x) {
    case SRC_H: v = H[(long)b * sH + s.y]; break;
    case SRC_HP: v = Hp[(long)b * sHp + s.y]; break;
    case SRC_HPP: v = Hpp[(long)b * sHpp + s.y]; break;
    case SRC_AB_NEG: v = -AB[(long)b * sAB + s.y]; break;
    default: v = 1.0;
  }
  if (isP) P[(long)b * nnzP + e] = v; else A[(long)b * nnzA + (e - nnzP)] = v;
}

}  // namespace

struct mpcqp_stageqp {
  SqPattern pat;
  int batch = 0, device = 0;
  mpcqp_handle *h = nullptr;
  int2 *dsrcP = nullptr, *dsrcA = nullptr;
  double *dP = nullptr, *dA = nullptr;                                  // the CSC value arrays the handle borrows
  double *dH = nullptr, *dHp = nullptr, *dHpp = nullptr, *dAB = nullptr, *dq = nullptr, *dl = nullptr, *du = nullptr;   // staging of host-memory updates
};

extern "C" {

int mpcqp_stageqp_pattern(const mpcqp_stageqp_dims *d, int *sizes4, int *Pp, int *Pi, int *Ap, int *Ai) {
  const SqPattern q = build_pattern(d);
  if (!q.error.empty()) return mpcqp_set_error(MPCQP_ERR_ARG, q.error);
  if (sizes4) { sizes4[0] = q.n; sizes4[1] = q.m; sizes4[2] = (int)q.Pi.size(); sizes4[3] = (int)q.Ai.size(); }
  if (Pp) std::copy(q.Pp.begin(), q.Pp.end(), Pp);
  if (Pi) std::copy(q.Pi.begin(), q.Pi.end(), Pi);
  if (Ap) std::copy(q.Ap.begin(), q.Ap.end(), Ap);
  if (Ai) std::copy(q.Ai.begin(), q.Ai.end(), Ai);
  return MPCQP_OK;
}

void mpcqp_stageqp_destroy(mpcqp_stageqp *s) {
  if (!s) return;
  if (s->h) mpcqp_destroy(s->h);       // (waits for the handle's last solve: nothing reads the arrays below after this)
  (void)hipSetDevice(s->device);
  for (void *p : {(void *)s->dsrcP, (void *)s->dsrcA, (void *)s->dP, (void *)s->dA, (void *)s->dH, (void *)s->dHp, (void *)s->dHpp, (void *)s->dAB, (void *)s->dq, (void *)s->dl, (void *)s->du})
    if (p) (void)hipFree(p);
  delete s;
}

int mpcqp_stageqp_create(const mpcqp_stageqp_dims *d, int batch, const mpcqp_settings *settings, mpcqp_stageqp **out) {
  if (!out) return mpcqp_set_error(MPCQP_ERR_ARG, "out is null");
  *out = nullptr;
  if (batch <= 0) return mpcqp_set_error(MPCQP_ERR_ARG, "Invalid dimensions.");
  mpcqp_stageqp *s = new mpcqp_stageqp();
  s->pat = build_pattern(d);
  if (!s->pat.error.empty()) { const std::string e = s->pat.error; delete s; return mpcqp_set_error(MPCQP_ERR_ARG, e); }
  s->batch = batch;
  const SqPattern &q = s->pat;
  int rc = mpcqp_create(q.n, q.m, batch, q.Pp.data(), q.Pi.data(), q.Ap.data(), q.Ai.data(), settings, &s->h);
  if (rc) { delete s; return rc; }
  if ((rc = mpcqp_pick_device(settings ? settings->device : -1, &s->device))) { mpcqp_stageqp_destroy(s); return rc; }
  auto bail = [&](hipError_t e, const char *what) { mpcqp_stageqp_destroy(s); return mpcqp_set_error(MPCQP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); };
  hipError_t e;
  if ((e = hipSetDevice(s->device)) != hipSuccess) return bail(e, "hipSetDevice");
  const size_t nP = q.Pi.size(), nA = q.Ai.size(), B = (size_t)batch;
  static_assert(sizeof(SqSrc) == sizeof(int2), "source records are uploaded as int2");
  if ((e = hipMalloc((void **)&s->dsrcP, std::max<size_t>(nP, 1) * sizeof(int2))) != hipSuccess) return bail(e, "hipMalloc");
  if ((e = hipMalloc((void **)&s->dsrcA, std::max<size_t>(nA, 1) * sizeof(int2))) != hipSuccess) return bail(e, "hipMalloc");
  if ((e = hipMalloc((void **)&s->dP, std::max<size_t>(B * nP, 1) * sizeof(double))) != hipSuccess) return bail(e, "hipMalloc");
  if ((e = hipMalloc((void **)&s->dA, std::max<size_t>(B * nA, 1) * sizeof(double))) != hipSuccess) return bail(e, "hipMalloc");
  if (nP && (e = hipMemcpy(s->dsrcP, q.Ps.data(), nP * sizeof(int2), hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy");
  if (nA && (e = hipMemcpy(s->dsrcA, q.As.data(), nA * sizeof(int2), hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy");
  *out = s;
  return MPCQP_OK;
}

mpcqp_handle *mpcqp_stageqp_handle(mpcqp_stageqp *s) { return s ? s->h : nullptr; }

int mpcqp_stageqp_update(mpcqp_stageqp *s, const double *H, const double *Hp, const double *Hpp, const double *AB,
                         const double *q, const double *l, const double *u, int mem, void *stream) {
  if (!s) return mpcqp_set_error(MPCQP_ERR_ARG, "null handle");
  const SqPattern &p = s->pat;
  if (!H || !AB || !q || !l || !u || (p.np > 0 && (!Hp || !Hpp))) return mpcqp_set_error(MPCQP_ERR_ARG, "null data pointer");
  if (mem != MPCQP_MEM_HOST && mem != MPCQP_MEM_DEVICE) return mpcqp_set_error(MPCQP_ERR_ARG, "mem must be MPCQP_MEM_HOST or MPCQP_MEM_DEVICE");
  MPCQP_HIPCHK(hipSetDevice(s->device));
  hipStream_t st = (hipStream_t)stream;
  const long sH = (long)p.N * p.f * p.f, sHp = (long)p.N * p.np * p.f, sHpp = (long)p.np * p.np, sAB = (long)(p.N - 1) * p.nx * p.f;
  const size_t B = (size_t)s->batch;
  if (mem == MPCQP_MEM_HOST) {
    // host blocks: staged once per update into owned device buffers (like mpcqp_update with MPCQP_MEM_HOST: the caller may free at once)
    auto up = [&](double **own, const double *src, size_t count) -> int {
      if (!count) return MPCQP_OK;
      if (!*own) MPCQP_HIPCHK(hipMalloc((void **)own, count * sizeof(double)));
      MPCQP_HIPCHK(hipMemcpyAsync(*own, src, count * sizeof(double), hipMemcpyHostToDevice, st));
      return MPCQP_OK;
    };
    int rc;
    if ((rc = mpcqp_sync(s->h))) return rc;       // (a solve may still be reading the staging buffers of the previous update)
    if ((rc = up(&s->dH, H, B * sH)) || (rc = up(&s->dHp, Hp, B * sHp)) || (rc = up(&s->dHpp, Hpp, B * sHpp)) || (rc = up(&s->dAB, AB, B * sAB)) ||
        (rc = up(&s->dq, q, B * p.n)) || (rc = up(&s->dl, l, B * p.m)) || (rc = up(&s->du, u, B * p.m))) return rc;
    MPCQP_HIPCHK(hipStreamSynchronize(st));                                // (pageable host memory: the caller's arrays are free from here on)
    H = s->dH; Hp = s->dHp; Hpp = s->dHpp; AB = s->dAB; q = s->dq; l = s->dl; u = s->du;
  }
  const int nP = (int)p.Pi.size(), nA = (int)p.Ai.size();
  const long total = ((long)nP + nA) * s->batch;
  { int rc = mpcqp_order_after_last_solve(s->h, st); if (rc) return rc; }      // the packed P, A are rewritten below: a solve on another stream may still read them
  hipLaunchKernelGGL(stageqp_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, s->batch, nP, nA, s->dsrcP, s->dsrcA,
                     H, sH, Hp, sHp, Hpp, sHpp, AB, sAB, s->dP, s->dA);
  MPCQP_HIPCHK(hipGetLastError());
  return mpcqp_update(s->h, s->dP, nP, q, p.n, s->dA, nA, l, p.m, u, p.m, MPCQP_MEM_DEVICE);
}

}  // extern "C"
