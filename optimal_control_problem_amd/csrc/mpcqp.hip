// mpcqp.hip -- MI355X (gfx950) batched OSQP-style ADMM: device kernels + the C ABI of include/mpcqp.h.
//
// Replaces, for a batch of QPs sharing one sparsity, what the reference does per QP on the host through
// CuCaQP::setSystem -> initSolver -> solve -> getSolution (reference src/sqp_solver/CuCaQP.cpp:271-288,
// 183-224), i.e. OSQP's osqp_setup + osqp_solve (external to the reference, see oracle/osqp_oracle.h).
//
// One launch does the whole sequence for every QP of the batch (grid = batch, one workgroup per QP).  Kernel families
// (mpcqp_create picks one per sparsity and batch size from measured rules):
//   * mpcqp_res_kernel<NW, MINW, GB, REUSE, ZYG> -- NW wavefronts per QP (1 or 4), W-fused block LDL' of
//     M = P + sigma I + A' diag(rho) A in 16x16 blocks, solve driven by a host-built schedule of arithmetic-progression
//     segments (plan.hpp).  GB = false: the factor lives in LDS (small / mid-size problems, and any batch that fits one
//     resident round); GB = true: the factor stays in the per-QP HBM slab and LDS holds only vectors, temp tiles and the
//     schedule, so that 3-4 workgroups share a CU (the default for the 12-state quadrotor sizes; HBM-roofline-bound).
//     MINW selects the register budget (128 / 168 / 256 VGPRs), REUSE the kept-workspace entry (mpcqp_update_vectors),
//     ZYG keeps z and y in the slab too (long horizons).
//   * mpcqp_admm_kernel<PD> -- the first-generation streaming kernel, one QP per wavefront, block Cholesky streamed from
//     the slab; fallback when even the vectors exceed LDS, and a cross-check in the variant tests.
// Common to all: ADMM iterates in LDS; scaled A in two ELL orientations and scaled P in the slab, streamed with coalesced
// 512 B wave loads; the linear solve is a stream of 16x16 block mat-vecs (4 lanes per row, quad DPP reduction); the
// factorisation's block products are dense 16x16x16 GEMMs on the matrix cores (v_mfma_f64_16x16x4_f64); box projection,
// dual update and residual norms fused into the ELL sweeps; termination, infeasibility certificates and adaptive-rho
// re-factorisation in-kernel; workgroup-uniform state in scalar registers (uni()).
// Numerics are fp64 throughout and follow oracle/osqp_oracle.c step by step (same scaling rule, rho rule,
// termination / infeasibility tests and deterministic adaptive-rho schedule).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mpcqp.h"
#include "plan.hpp"
#include "common.hpp"

using namespace mpcqp;

#define Q_INFTY 1e30
#define Q_MIN_SCALING 1e-4
#define Q_MAX_SCALING 1e4
#define Q_RHO_MIN 1e-6
#define Q_RHO_MAX 1e6
#define Q_RHO_TOL 1e-4
#define Q_RHO_EQ 1e3
#define Q_DIV_TOL 1e-10

typedef double d4 __attribute__((ext_vector_type(4)));

struct DevEll { int nchunks; const int *chunk_off, *idx, *src, *flag; long entries; };
struct DevPlan {
  int n, m, npad, mpad, nb, nblk, nfac, nT;
  DevEll A, At, P;
  const int *pos, *perm, *fwd_ops, *bwd_ops, *bwd_of;
  const int4 *fac;
  const int *tpos, *asm_ptr, *asm_a, *asm_b, *asm_pidx, *blk_diag;
  long o_ellA, o_ellAt, o_ellP, o_Lf, o_Lb, o_T, o_l, o_u, o_D, o_E, o_dx, o_dy, ws_stride, o_Zg, o_Yg;
};
struct DevIO {
  const double *P, *q, *A, *l, *u; long sP, sq, sA, sl, su;
  const double *x0, *y0, *rho0;
  double *x, *y, *z; int *status, *iters; double *info;
  double *ws; double *cscale; long long *dbg;
  const int *order;    // dispatch order: workgroup g solves instance order[g] (NULL = identity); see mpcqp_order_kernel
  int reuse, keep;     // kept workspace: skip scaling + factorisation (mpcqp_update_vectors) / store the factor for that
};

// ------------------------------------------------------------------------------------------ device helpers
// A value every lane holds identically (the result of a workgroup reduction, rho, the cost scale): moved to scalar registers,
// so that state that lives across the whole ADMM loop does not occupy vector registers of the 128-VGPR kernel instances.
__device__ __forceinline__ double uni(double v) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(b & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double limit_scaling(double v) {
  v = v < Q_MIN_SCALING ? 1.0 : v;
  return v > Q_MAX_SCALING ? Q_MAX_SCALING : v;
}
__device__ __forceinline__ double rho_of(double l, double u, double rho) {
  if (l < -Q_INFTY * Q_MIN_SCALING && u > Q_INFTY * Q_MIN_SCALING) return Q_RHO_MIN;
  if (u - l < Q_RHO_TOL) return Q_RHO_EQ * rho;
  return rho;
}
// single-wave workgroup: orders LDS / global accesses between lanes of the wave
__device__ __forceinline__ void wsync() { __syncthreads(); }

// out(row, sum_s val * in[idx]) over an ELL structure; rows are lane-mapped, loads are 512 B coalesced
template <class F>
__device__ __forceinline__ void ell_rows(const DevEll &E, const double *__restrict__ val, const double *in, F &&f) {
  const int lane = threadIdx.x;
  for (int c = 0; c < E.nchunks; c++) {
    const int s0 = E.chunk_off[c], s1 = E.chunk_off[c + 1];
    double acc = 0.0;
#pragma unroll 4
    for (int s = s0; s < s1; s++) {
      const unsigned e = (unsigned)s * WAVE + lane;
      acc += val[e] * in[E.idx[e]];
    }
    f(c * WAVE + lane, acc);
  }
}
// max_s |val| * in[idx]
template <class F>
__device__ __forceinline__ void ell_rowmax(const DevEll &E, const double *__restrict__ val, const double *in, F &&f) {
  const int lane = threadIdx.x;
  for (int c = 0; c < E.nchunks; c++) {
    const int s0 = E.chunk_off[c], s1 = E.chunk_off[c + 1];
    double acc = 0.0;
#pragma unroll 4
    for (int s = s0; s < s1; s++) {
      const unsigned e = (unsigned)s * WAVE + lane;
      acc = fmax(acc, fabs(val[e]) * in[E.idx[e]]);
    }
    f(c * WAVE + lane, acc);
  }
}

// acc += A * B^T for row-major 16x16 blocks in global memory, on the matrix cores.
// v_mfma_f64_16x16x4_f64: lane l feeds A[l&15][k0 + (l>>4)] and B^T[k][j] = B[l&15][k0 + (l>>4)];
// result register g of lane l is C[(l>>4) + 4g][l&15].
__device__ __forceinline__ d4 mfma_abt(const double *__restrict__ A, const double *__restrict__ B, d4 acc) {
  const int lane = threadIdx.x, rr = lane & 15, kk = lane >> 4;
#pragma unroll
  for (int k0 = 0; k0 < BS; k0 += 4) {
    const double a = A[rr * BS + k0 + kk];
    const double b = B[rr * BS + k0 + kk];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  return acc;
}

// In-LDS Cholesky of a 16x16 SPD block followed by the inverse of its factor.
// gf: block in global (in: S, lower triangle used; out: Linv, row-major, zero above the diagonal);
// gb: transposed copy (backward stream). S0/S1: 16x17 LDS tiles. Returns false on a non-positive pivot.
__device__ bool potrf_inv(double *gf, double *gb, double *S0, double *S1) {
  const int lane = threadIdx.x, r = lane >> 2, j = lane & 3;
  {
    const d4 v = reinterpret_cast<const d4 *>(gf)[lane];
#pragma unroll
    for (int c = 0; c < 4; c++) S0[r * 17 + 4 * j + c] = v[c];
  }
  wsync();
  for (int k = 0; k < BS; k++) {
    const double d = S0[k * 17 + k];
    if (!(d > 0.0)) return false;          // uniform across the wave
    const double sd = sqrt(d), inv = 1.0 / sd;
    const double lrk = S0[r * 17 + k] * inv;
    double lc[4];
#pragma unroll
    for (int c = 0; c < 4; c++) lc[c] = S0[(4 * j + c) * 17 + k] * inv;
    wsync();
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int col = 4 * j + c;
      if (col > k && r >= col) S0[r * 17 + col] -= lrk * lc[c];
    }
    if ((k >> 2) == j) {
      if (r > k) S0[r * 17 + k] = lrk;
      else if (r == k) S0[r * 17 + k] = sd;
    }
    wsync();
  }
  if (lane < BS) {
    const int c = lane;
    for (int i = 0; i < BS; i++) {
      double v;
      if (i < c) v = 0.0;
      else if (i == c) v = 1.0 / S0[i * 17 + i];
      else {
        double s = 0.0;
        for (int k = c; k < i; k++) s += S0[i * 17 + k] * S1[k * 17 + c];
        v = -s / S0[i * 17 + i];
      }
      S1[i * 17 + c] = v;
    }
  }
  wsync();
  d4 f, t;
#pragma unroll
  for (int c = 0; c < 4; c++) { f[c] = S1[r * 17 + 4 * j + c]; t[c] = S1[(4 * j + c) * 17 + r]; }
  reinterpret_cast<d4 *>(gf)[lane] = f;
  reinterpret_cast<d4 *>(gb)[lane] = t;
  return true;
}

// Stream of 16x16 block mat-vecs over an LDS vector: op t uses block t of `blk` (2 KiB, coalesced),
// DIAG: vec_d = B vec_s (in place), OFF: vec_d -= B vec_s. 4 lanes per row + quad shuffle reduction.
__device__ __forceinline__ void stream_step(const d4 bb, const int op, double *vec, const int r, const int j) {
  const int kind = op & 1, s = (op >> 1) & 0x7fff, d = op >> 16;
  const double *v = vec + BS * s + 4 * j;
  double part = bb[0] * v[0] + bb[1] * v[1] + bb[2] * v[2] + bb[3] * v[3];
  part += __shfl_xor(part, 1);
  part += __shfl_xor(part, 2);
  if (j == 0) {
    double *o = vec + BS * d + r;
    *o = kind ? *o - part : part;
  }
  wsync();
}
template <int PD>
__device__ __forceinline__ void run_stream(const double *__restrict__ blk, const int *__restrict__ ops, const int nops, double *vec) {
  const int lane = threadIdx.x, r = lane >> 2, j = lane & 3;
  const d4 *p = reinterpret_cast<const d4 *>(blk) + lane;
  // PD blocks (2 KiB each) of the stream in flight per wave: 4 when six QPs share a CU (8 measured slower there),
  // 8 when the LDS footprint leaves only a few waves per CU to cover the HBM latency
  d4 b[PD];
#pragma unroll
  for (int u = 0; u < PD; u++) { b[u] = d4{0, 0, 0, 0}; if (u < nops) b[u] = p[(long)u * WAVE]; }
  for (int t = 0; t < nops; t += PD) {
#pragma unroll
    for (int u = 0; u < PD; u++) {
      if (t + u < nops) {
        stream_step(b[u], ops[t + u], vec, r, j);
        if (t + u + PD < nops) b[u] = p[(long)(t + u + PD) * WAVE];
      }
    }
  }
}

// residual / norm bundle produced by update_info (oracle/osqp_oracle.c update_info + rho_estimate inputs)
struct Info {
  double prim_res, dual_res, obj;
  double nz, nax, nq, naty, npx;            // norms entering eps_prim / eps_dual (unscaled unless scaled_termination)
  double prs, drs, nzs, naxs, nqs, natys, npxs;  // scaled-space norms for the rho estimate
};

struct Ctx {
  const DevPlan *pl; const mpcqp_settings *st; double *ws;
  double *X, *Q, *R, *Z, *Y, *W, *S0, *S1;
  double c, cinv, rho; int unscale;
};

// Factorise M(rho): build rho vector, singleton diagonal, T = sqrt(rho) A_general^T, assemble blocks on the
// matrix cores, left-looking block Cholesky. Leaves W = rho*Z - Y. Returns false if M is not positive definite
// (same inertia test as OSQP's "KKT has n positive pivots").
__device__ bool factorize(Ctx &cx) {
  const DevPlan &pl = *cx.pl; double *ws = cx.ws;
  const int lane = threadIdx.x;
  const double *lb = ws + pl.o_l, *ub = ws + pl.o_u;
  double *valA = ws + pl.o_ellA, *valAt = ws + pl.o_ellAt, *valP = ws + pl.o_ellP;
  double *Lf = ws + pl.o_Lf, *Lb = ws + pl.o_Lb, *T = ws + pl.o_T;
  for (int i = lane; i < pl.mpad; i += WAVE) cx.W[i] = i < pl.m ? rho_of(lb[i], ub[i], cx.rho) : 0.0;
  wsync();
  {
    const double sigma = cx.st->sigma;
    const DevEll &E = pl.At;
    for (int c = 0; c < E.nchunks; c++) {
      double acc = 0.0;
      for (int s = E.chunk_off[c]; s < E.chunk_off[c + 1]; s++) {
        const unsigned e = (unsigned)s * WAVE + lane;
        const double v = valAt[e];
        if (E.flag[e]) acc += cx.W[E.idx[e]] * v * v;
      }
      const int t = c * WAVE + lane;
      if (t < pl.npad) cx.R[t] = pl.perm[t] >= 0 ? sigma + acc : 1.0;
    }
  }
  for (long k = lane; k < (long)pl.nT * BLK; k += WAVE) T[k] = 0.0;
  wsync();
  {
    const DevEll &E = pl.A;
    for (int c = 0; c < E.nchunks; c++) {
      const int i = c * WAVE + lane;
      const double sr = sqrt(cx.W[i]);
      for (int s = E.chunk_off[c]; s < E.chunk_off[c + 1]; s++) {
        const unsigned e = (unsigned)s * WAVE + lane;
        const int tp = pl.tpos[e];
        if (tp >= 0) T[tp] = valA[e] * sr;
      }
    }
  }
  wsync();
  const int row0 = lane >> 4, col = lane & 15;
  for (int b = 0; b < pl.nblk; b++) {
    d4 acc = {0, 0, 0, 0};
    for (int g = pl.asm_ptr[b]; g < pl.asm_ptr[b + 1]; g++) acc = mfma_abt(T + (long)pl.asm_a[g] * BLK, T + (long)pl.asm_b[g] * BLK, acc);
    const int J = pl.blk_diag[b];
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int pi = pl.asm_pidx[(long)b * BLK + g * WAVE + lane];
      if (pi >= 0) acc[g] += valP[pi];
      const int row = row0 + 4 * g;
      if (J >= 0 && row == col) acc[g] += cx.R[J * BS + row];
      Lf[(long)b * BLK + row * BS + col] = acc[g];
    }
  }
  wsync();
  for (int f = 0; f < pl.nfac; f++) {
    const int4 op = pl.fac[f];
    double *dst = Lf + (long)op.y * BLK;
    if (op.x == FAC_SUB) {
      d4 prod = {0, 0, 0, 0};
      prod = mfma_abt(Lf + (long)op.z * BLK, Lf + (long)op.w * BLK, prod);
#pragma unroll
      for (int g = 0; g < 4; g++) dst[(row0 + 4 * g) * BS + col] -= prod[g];
    } else if (op.x == FAC_POTRF) {
      if (!potrf_inv(dst, Lb + (long)pl.bwd_of[op.y] * BLK, cx.S0, cx.S1)) return false;
    } else {
      d4 prod = {0, 0, 0, 0};
      prod = mfma_abt(dst, Lf + (long)op.z * BLK, prod);
      double *dbt = Lb + (long)pl.bwd_of[op.y] * BLK;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        dst[(row0 + 4 * g) * BS + col] = prod[g];
        dbt[col * BS + row0 + 4 * g] = prod[g];
      }
    }
    wsync();
  }
  for (int i = lane; i < pl.mpad; i += WAVE) cx.W[i] = cx.W[i] * cx.Z[i] - cx.Y[i];
  wsync();
  return true;
}

// residuals, norms and objective at the current iterate (oracle update_info); clobbers R
__device__ void update_info(Ctx &cx, Info &in) {
  const DevPlan &pl = *cx.pl; double *ws = cx.ws;
  const double *Dg = ws + pl.o_D, *Eg = ws + pl.o_E;
  const double *valA = ws + pl.o_ellA, *valAt = ws + pl.o_ellAt, *valP = ws + pl.o_ellP;
  const int unscale = cx.unscale;
  double pr = 0, nz = 0, nax = 0, prs = 0, nzs = 0, naxs = 0;
  ell_rows(pl.A, valA, cx.X, [&](int i, double ax) {
    if (i < pl.m) {
      const double einv = unscale ? 1.0 / Eg[i] : 1.0, zi = cx.Z[i];
      pr = fmax(pr, fabs(einv * (ax - zi))); nax = fmax(nax, fabs(einv * ax)); nz = fmax(nz, fabs(einv * zi));
      prs = fmax(prs, fabs(ax - zi)); naxs = fmax(naxs, fabs(ax)); nzs = fmax(nzs, fabs(zi));
    }
  });
  ell_rows(pl.P, valP, cx.X, [&](int t, double px) { if (t < pl.npad) cx.R[t] = px; });
  wsync();
  double dr = 0, nq = 0, naty = 0, npx = 0, drs = 0, nqs = 0, natys = 0, npxs = 0, obj = 0;
  ell_rows(pl.At, valAt, cx.Y, [&](int t, double aty) {
    if (t < pl.npad) {
      const double dinv = unscale ? 1.0 / Dg[t] : 1.0, px = cx.R[t], qv = cx.Q[t], du = qv + px + aty;
      dr = fmax(dr, fabs(dinv * du)); nq = fmax(nq, fabs(dinv * qv)); naty = fmax(naty, fabs(dinv * aty)); npx = fmax(npx, fabs(dinv * px));
      drs = fmax(drs, fabs(du)); nqs = fmax(nqs, fabs(qv)); natys = fmax(natys, fabs(aty)); npxs = fmax(npxs, fabs(px));
      obj += cx.X[t] * (0.5 * px + qv);
    }
  });
  in.prim_res = wave_max(pr); in.nz = wave_max(nz); in.nax = wave_max(nax);
  in.prs = wave_max(prs); in.nzs = wave_max(nzs); in.naxs = wave_max(naxs);
  dr = wave_max(dr); in.nq = wave_max(nq); in.naty = wave_max(naty); in.npx = wave_max(npx);
  in.drs = wave_max(drs); in.nqs = wave_max(nqs); in.natys = wave_max(natys); in.npxs = wave_max(npxs);
  in.dual_res = unscale ? cx.cinv * dr : dr;
  obj = wave_sum(obj);
  in.obj = cx.st->scaling ? cx.cinv * obj : obj;
  wsync();
}

// oracle is_primal_infeasible; uses W as scratch and restores W = rho*Z - Y
__device__ bool primal_infeasible(Ctx &cx, double eps) {
  const DevPlan &pl = *cx.pl; double *ws = cx.ws; const int lane = threadIdx.x;
  const double *lb = ws + pl.o_l, *ub = ws + pl.o_u, *Eg = ws + pl.o_E, *Dg = ws + pl.o_D, *dy = ws + pl.o_dy;
  double nrm = 0, lhs = 0;
  for (int i = lane; i < pl.mpad; i += WAVE) {
    double v = 0.0;
    if (i < pl.m) {
      v = dy[i];
      const double lo = lb[i], up = ub[i];
      if (up > Q_INFTY * Q_MIN_SCALING) { if (lo < -Q_INFTY * Q_MIN_SCALING) v = 0.0; else v = fmin(v, 0.0); }
      else if (lo < -Q_INFTY * Q_MIN_SCALING) v = fmax(v, 0.0);
      nrm = fmax(nrm, fabs(cx.unscale ? Eg[i] * v : v));
      lhs += up * fmax(v, 0.0) + lo * fmin(v, 0.0);
    }
    cx.W[i] = v;
  }
  nrm = wave_max(nrm); lhs = wave_sum(lhs);
  wsync();
  bool res = false;
  if (nrm > eps && lhs < -eps * nrm) {
    double a = 0;
    ell_rows(pl.At, ws + pl.o_ellAt, cx.W, [&](int t, double v) { if (t < pl.npad) a = fmax(a, fabs(cx.unscale ? (1.0 / Dg[t]) * v : v)); });
    a = wave_max(a);
    res = a < eps * nrm;
  }
  wsync();
  for (int i = lane; i < pl.mpad; i += WAVE) cx.W[i] = i < pl.m ? rho_of(lb[i], ub[i], cx.rho) * cx.Z[i] - cx.Y[i] : 0.0;
  wsync();
  return res;
}

// oracle is_dual_infeasible; uses R as scratch
__device__ bool dual_infeasible(Ctx &cx, double eps) {
  const DevPlan &pl = *cx.pl; double *ws = cx.ws; const int lane = threadIdx.x;
  const double *lb = ws + pl.o_l, *ub = ws + pl.o_u, *Eg = ws + pl.o_E, *Dg = ws + pl.o_D, *dx = ws + pl.o_dx;
  double nrm = 0, qdx = 0;
  for (int t = lane; t < pl.npad; t += WAVE) {
    const double v = dx[t];
    cx.R[t] = v;
    nrm = fmax(nrm, fabs(cx.unscale ? Dg[t] * v : v));
    qdx += cx.Q[t] * v;
  }
  nrm = wave_max(nrm); qdx = wave_sum(qdx);
  wsync();
  const double cs = cx.unscale ? cx.c : 1.0;
  bool res = false;
  if (nrm > eps && qdx < -cs * eps * nrm) {
    double a = 0;
    ell_rows(pl.P, ws + pl.o_ellP, cx.R, [&](int t, double v) { if (t < pl.npad) a = fmax(a, fabs(cx.unscale ? (1.0 / Dg[t]) * v : v)); });
    a = wave_max(a);
    if (a < cs * eps * nrm) {
      int bad = 0;
      ell_rows(pl.A, ws + pl.o_ellA, cx.R, [&](int i, double v) {
        if (i < pl.m) {
          if (cx.unscale) v = (1.0 / Eg[i]) * v;
          if ((ub[i] < Q_INFTY * Q_MIN_SCALING && v > eps * nrm) || (lb[i] > -Q_INFTY * Q_MIN_SCALING && v < -eps * nrm)) bad = 1;
        }
      });
      res = !__any(bad);
    }
  }
  wsync();
  return res;
}

// oracle check_termination; returns new status (or UNSOLVED)
__device__ int check_termination(Ctx &cx, Info &in, int approximate) {
  const mpcqp_settings &st = *cx.st;
  double eps_abs = st.eps_abs, eps_rel = st.eps_rel, epi = st.eps_prim_inf, edi = st.eps_dual_inf;
  if (in.prim_res > Q_INFTY || in.dual_res > Q_INFTY || in.prim_res != in.prim_res || in.dual_res != in.dual_res) { in.obj = NAN; return MPCQP_NON_CVX; }
  if (approximate) { eps_abs *= 10; eps_rel *= 10; epi *= 10; edi *= 10; }
  bool pc = false, dc = false, pic = false, dic = false;
  if (cx.pl->m == 0) pc = true;
  else {
    const double eps_prim = eps_abs + eps_rel * fmax(in.nz, in.nax);
    if (in.prim_res < eps_prim) pc = true; else pic = primal_infeasible(cx, epi);
  }
  {
    double mx = fmax(in.nq, fmax(in.naty, in.npx));
    if (cx.unscale) mx *= cx.cinv;
    const double eps_dual = eps_abs + eps_rel * mx;
    if (in.dual_res < eps_dual) dc = true; else dic = dual_infeasible(cx, edi);
  }
  if (pc && dc) return approximate ? MPCQP_SOLVED_INACCURATE : MPCQP_SOLVED;
  if (pic) { in.obj = Q_INFTY; return approximate ? MPCQP_PRIMAL_INFEASIBLE_INACCURATE : MPCQP_PRIMAL_INFEASIBLE; }
  if (dic) { in.obj = -Q_INFTY; return approximate ? MPCQP_DUAL_INFEASIBLE_INACCURATE : MPCQP_DUAL_INFEASIBLE; }
  return MPCQP_UNSOLVED;
}

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


// =========================================================================================================
// Resident variant: NW waves per QP, block LDL' factor (G_J = D_J^-1, W_IJ) held in LDS for the whole solve.
// HBM is touched per iteration only for the ELL sweeps of A / A' (L2-resident per-QP slabs) and l, u.
// =========================================================================================================
#ifdef MPCQP_TIMING
#define TS_DECL unsigned long long ts_last = __builtin_amdgcn_s_memtime(), ts_acc[16] = {0}
#define TS(k) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); ts_acc[k] += t_ - ts_last; ts_last = t_; } while (0)
#define TS_STORE(ptr) do { if (tid == 0 && (ptr)) for (int k_ = 0; k_ < 16; k_++) (ptr)[16L * b + k_] = (long long)ts_acc[k_]; } while (0)
#else
#define TS_DECL
#define TS(k)
#define TS_STORE(ptr)
#endif
struct DevRes {
  int nphase, ntemp, nconst, rext;   // nconst constant blocks behind the factor blocks (slot nblk = -I), rext partial-sum doubles behind the solve vector
  const int *lv_ptr, *lv_diag, *lw_ptr, *lw_slot, *lw_g, *lu_ptr, *lu_dst, *lu_tmp, *lu_b, *g_ptr, *g_seg;
  int n_seg, nlev; long stage;
};

template <int NW> __device__ __forceinline__ void bsync() { __syncthreads(); }

struct RCtx {
  const DevPlan *pl; const DevRes *rs; const mpcqp_settings *st; double *ws;
  double *BL, *TMP, *X, *Q, *R, *Z, *Y, *W, *RB, *RED;
  double c, cinv, rho; int unscale; int wid, lane;
  unsigned long long fts[4];
};

// reduce K per-thread values over the workgroup: the first K - NSUM by max, the last NSUM by sum
template <int NW, int K, int NSUM>
__device__ __forceinline__ void block_combine(double (&v)[K], double *red, int wid, int lane) {
#pragma unroll
  for (int k = 0; k < K; k++) v[k] = k >= K - NSUM ? wave_sum(v[k]) : wave_max(v[k]);
  if (NW > 1) {
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < K; k++) red[wid * K + k] = v[k];
    }
    bsync<NW>();
#pragma unroll
    for (int k = 0; k < K; k++) {
      double r = red[k];
      for (int w = 1; w < NW; w++) r = k >= K - NSUM ? r + red[w * K + k] : fmax(r, red[w * K + k]);
      v[k] = r;
    }
    bsync<NW>();
  }
}

// One ELL chunk (64 rows, slots s0..s1) for this lane's row: sum_s val * in[idx]  (or max_s |val| * in[idx]).
// Pointer bumps + compile-time strides keep the address arithmetic in the loads' immediate offsets (the generic
// indexed form spent ~12 instructions per slot on 64-bit address math); batches of 8 / 4 / 2 / 1 slots issue all
// their loads before the first use.
template <bool MAXABS, int U>
__device__ __forceinline__ double ell_batch(const double *__restrict__ &vp, const int *__restrict__ &ip, const double *in, double acc) {
  double v[U]; int ix[U];
#pragma unroll
  for (int u = 0; u < U; u++) { v[u] = vp[u * WAVE]; ix[u] = ip[u * WAVE]; }
#pragma unroll
  for (int u = 0; u < U; u++) {
    const double x = in[ix[u]];
    acc = MAXABS ? fmax(acc, fabs(v[u]) * x) : acc + v[u] * x;
  }
  vp += U * WAVE; ip += U * WAVE;
  return acc;
}
template <bool MAXABS, int UMAX = 16>
__device__ __forceinline__ double ell_chunk(const double *__restrict__ val, const int *__restrict__ idx, const double *in, const int s0, const int s1, const int lane) {
  const double *__restrict__ vp = val + ((long)s0 * WAVE + lane);
  const int *__restrict__ ip = idx + ((long)s0 * WAVE + lane);
  double acc = 0.0;
  int rem = s1 - s0;
  if (UMAX >= 16) {
    for (; rem >= 16; rem -= 16) acc = ell_batch<MAXABS, 16>(vp, ip, in, acc);
    if (rem & 8) acc = ell_batch<MAXABS, 8>(vp, ip, in, acc);
  } else {
    for (; rem >= 8; rem -= 8) acc = ell_batch<MAXABS, 8>(vp, ip, in, acc);     // 16 loads in flight spill in the 128-VGPR instances
  }
  if (rem & 4) acc = ell_batch<MAXABS, 4>(vp, ip, in, acc);
  if (rem & 2) acc = ell_batch<MAXABS, 2>(vp, ip, in, acc);
  if (rem & 1) acc = ell_batch<MAXABS, 1>(vp, ip, in, acc);
  return acc;
}
// ELL sweeps for the multi-wave kernels: wave `wid` takes chunks wid, wid + NW, ... (A 16-deep clamped full unroll
// and a 4-lanes-per-row split were both measured slower on MI355X: spills / more latency rounds; see DESIGN.md.)
template <int NW, int UMAX = 16, class F>
__device__ __forceinline__ void ell_rows_w(const DevEll &E, const double *__restrict__ val, const double *in, int wid, int lane, F &&f) {
  for (int c = wid; c < E.nchunks; c += NW) f(c * WAVE + lane, ell_chunk<false, UMAX>(val, E.idx, in, E.chunk_off[c], E.chunk_off[c + 1], lane));
}
template <int NW, class F>
__device__ __forceinline__ void ell_rowmax_w(const DevEll &E, const double *__restrict__ val, const double *in, int wid, int lane, F &&f) {
  for (int c = wid; c < E.nchunks; c += NW) f(c * WAVE + lane, ell_chunk<true>(val, E.idx, in, E.chunk_off[c], E.chunk_off[c + 1], lane));
}

// sum over the 4 lanes of a quad with DPP quad_perm (no LDS crossbar round trip)
__device__ __forceinline__ double quad_sum(double v) {
  union { double d; int i[2]; } a, t;
  a.d = v;
  t.i[0] = __builtin_amdgcn_mov_dpp(a.i[0], 0xB1, 0xF, 0xF, true);   // quad_perm:[1,0,3,2]
  t.i[1] = __builtin_amdgcn_mov_dpp(a.i[1], 0xB1, 0xF, 0xF, true);
  a.d += t.d;
  t.i[0] = __builtin_amdgcn_mov_dpp(a.i[0], 0x4E, 0xF, 0xF, true);   // quad_perm:[2,3,0,1]
  t.i[1] = __builtin_amdgcn_mov_dpp(a.i[1], 0x4E, 0xF, 0xF, true);
  return a.d + t.d;
}

// acc += A * B^T, operands row-major 16x16 tiles (LDS or global), lane = lane within the wave
__device__ __forceinline__ d4 mfma_abt_l(const double *A, const double *B, d4 acc, int lane) {
  const int rr = lane & 15, kk = lane >> 4;
#pragma unroll
  for (int k0 = 0; k0 < BS; k0 += 4) {
    const double a = A[rr * BS + k0 + kk];
    const double b = B[rr * BS + k0 + kk];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  return acc;
}

// One wave: in-place inverse of the SPD 16x16 tile `a` (LDS, row-major) by symmetric sweeps; rb = 16 doubles of LDS.
// Pivots of the sweeps are the Cholesky pivots squared, so "all pivots > 0" is the positive-definiteness test.
__device__ __forceinline__ bool sweep_inverse(double *a, double *rb, int lane) {
  const int r = lane >> 2, j = lane & 3;
  d4 v = reinterpret_cast<const d4 *>(a)[lane];      // a[r][4j .. 4j+3]
#pragma unroll
  for (int k = 0; k < BS; k++) {
    if (r == k) reinterpret_cast<d4 *>(rb)[j] = v;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const d4 rk = reinterpret_cast<const d4 *>(rb)[j];
    const double colk = rb[r], d = rb[k];
    if (!(d > 0.0)) return false;                       // uniform
    const double p = 1.0 / d;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int col = 4 * j + c;
      double nv;
      if (r != k && col != k) nv = v[c] - colk * rk[c] * p;
      else if (r == k && col != k) nv = rk[c] * p;
      else if (r != k && col == k) nv = colk * p;
      else nv = -p;
      v[c] = nv;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  reinterpret_cast<d4 *>(a)[lane] = -v;
  return true;
}

// ---- solve-schedule executor -------------------------------------------------------------------------------
// A lone wave issues roughly one instruction per 4-8 cycles, so the executor is written for instruction count:
// the host compresses each wave's op list into segments whose block / src / dst byte offsets are arithmetic
// progressions; each segment is one tight, branch-free, software-pipelined loop (4 lanes per output row; the next
// op's 4 block entries and old destination value are fetched while the current op is reduced with a quad DPP sum).
template <bool T>
__device__ __forceinline__ d4 load_blk(const char *BLb, const int off, const int offN, const int offT) {
  if (T) {
    const char *B = BLb + off + offT;
    d4 r;
    r[0] = *reinterpret_cast<const double *>(B); r[1] = *reinterpret_cast<const double *>(B + BS * 8);
    r[2] = *reinterpret_cast<const double *>(B + 2 * BS * 8); r[3] = *reinterpret_cast<const double *>(B + 3 * BS * 8);
    return r;
  }
  return *reinterpret_cast<const d4 *>(BLb + off + offN);
}
__device__ __forceinline__ void wave_order() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// blocks streamed from the HBM slab (GB variants): PD blocks of the segment in flight, destination read at use time
template <bool T, bool SET, int PD>
__device__ __forceinline__ void seg_each_g(const char *BLb, char *vecb, int b, int s, int d, const int cnt, const int db, const int ds, const int dd,
                                           const int offN, const int offT, const int offV, const int offD, const int j, double carry) {
  d4 ring[PD];
#pragma unroll
  for (int u = 0; u < PD; u++) { ring[u] = d4{0, 0, 0, 0}; if (u < cnt) ring[u] = load_blk<T>(BLb, b + u * db, offN, offT); }
  for (int k0 = 0; k0 < cnt; k0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; u++) {
      if (k0 + u < cnt) {
        const d4 bb = ring[u];
        if (k0 + u + PD < cnt) ring[u] = load_blk<T>(BLb, b + (k0 + u + PD) * db, offN, offT);
        const d4 v = *reinterpret_cast<const d4 *>(vecb + s + offV);
        const double sum = quad_sum(carry + (bb[0] * v[0] + bb[1] * v[1] + bb[2] * v[2] + bb[3] * v[3]));
        carry = 0.0;
        if (j == 0) { double *o = reinterpret_cast<double *>(vecb + d + offD); *o = SET ? sum : *o - sum; }
        wave_order();
        s += ds; d += dd;
      }
    }
  }
}
// every op writes its own destination: dst_k = (SET ? 0 : dst_k) -/+ B_k * src_k ; `carry` joins the first op
template <bool T, bool SET>
__device__ __forceinline__ void seg_each(const char *BLb, char *vecb, int b, int s, int d, const int cnt, const int db, const int ds, const int dd,
                                         const int offN, const int offT, const int offV, const int offD, const int j, double carry) {
  d4 bb = load_blk<T>(BLb, b, offN, offT);
  double old = SET ? 0.0 : *reinterpret_cast<const double *>(vecb + d + offD);
  for (int k = 0; k < cnt; k++) {
    const d4 v = *reinterpret_cast<const d4 *>(vecb + s + offV);
    d4 nb = bb; double nold = old;
    if (k + 1 < cnt) {
      nb = load_blk<T>(BLb, b + db, offN, offT);
      if (!SET) nold = *reinterpret_cast<const double *>(vecb + d + dd + offD);
    }
    const double sum = quad_sum(carry + (bb[0] * v[0] + bb[1] * v[1] + bb[2] * v[2] + bb[3] * v[3]));
    carry = 0.0;
    if (j == 0) *reinterpret_cast<double *>(vecb + d + offD) = SET ? sum : old - sum;
    wave_order();
    bb = nb; old = nold; b += db; s += ds; d += dd;
  }
}
// SG_IND segments: the ops are mutually independent, so two are in flight at once (loads, FMAs, DPP sums interleave)
template <bool T, bool SET>
__device__ __forceinline__ void seg_each2(const char *BLb, char *vecb, int b, int s, int d, const int cnt, const int db, const int ds, const int dd,
                                          const int offN, const int offT, const int offV, const int offD, const int j) {
  int k = 0;
  for (; k + 2 <= cnt; k += 2) {
    const d4 b0 = load_blk<T>(BLb, b, offN, offT), b1 = load_blk<T>(BLb, b + db, offN, offT);
    const d4 v0 = *reinterpret_cast<const d4 *>(vecb + s + offV), v1 = *reinterpret_cast<const d4 *>(vecb + s + ds + offV);
    double o0 = 0.0, o1 = 0.0;
    if (!SET) { o0 = *reinterpret_cast<const double *>(vecb + d + offD); o1 = *reinterpret_cast<const double *>(vecb + d + dd + offD); }
    const double s0 = quad_sum(b0[0] * v0[0] + b0[1] * v0[1] + b0[2] * v0[2] + b0[3] * v0[3]);
    const double s1 = quad_sum(b1[0] * v1[0] + b1[1] * v1[1] + b1[2] * v1[2] + b1[3] * v1[3]);
    if (j == 0) {
      *reinterpret_cast<double *>(vecb + d + offD) = SET ? s0 : o0 - s0;
      *reinterpret_cast<double *>(vecb + d + dd + offD) = SET ? s1 : o1 - s1;
    }
    b += 2 * db; s += 2 * ds; d += 2 * dd;
  }
  wave_order();
  if (k < cnt) seg_each<T, SET>(BLb, vecb, b, s, d, cnt - k, db, ds, dd, offN, offT, offV, offD, j, 0.0);
}
// the ops accumulate into one destination; returns the per-lane partial sum
template <bool T, bool GB>
__device__ __forceinline__ double seg_run(const char *BLb, const char *vecb, int b, int s, const int cnt, const int db, const int ds,
                                          const int offN, const int offT, const int offV, double acc) {
#pragma unroll 8
  for (int k = 0; k < cnt; k++) {
    const d4 bb = load_blk<T>(BLb, b, offN, offT);
    const d4 v = *reinterpret_cast<const d4 *>(vecb + s + offV);
    acc += bb[0] * v[0] + bb[1] * v[1] + bb[2] * v[2] + bb[3] * v[3];
    b += db; s += ds;
  }
  return acc;
}
template <int NW, bool GB, int PD = 6>
__device__ __forceinline__ void run_schedule(const int4 *segs, const int g0, const int g1, const char *BLb, char *vecb, const int lane, const int ni_off, long long *trace = nullptr) {
  const int r = lane >> 2, j = lane & 3;
  const int offN = (r * BS + 4 * j) * 8, offT = ((4 * j) * BS + r) * 8, offV = 32 * j, offD = 8 * r;
  double acc = 0.0;
  for (int g = g0; g < g1; g++) {
    int4 a = segs[2 * g], c = segs[2 * g + 1];
    const int b0 = __builtin_amdgcn_readfirstlane(a.x), s0 = __builtin_amdgcn_readfirstlane(a.y), d0 = __builtin_amdgcn_readfirstlane(a.z);
    const int fl = __builtin_amdgcn_readfirstlane(a.w), cnt = __builtin_amdgcn_readfirstlane(c.x);
    const int db = __builtin_amdgcn_readfirstlane(c.y), ds = __builtin_amdgcn_readfirstlane(c.z), dd = __builtin_amdgcn_readfirstlane(c.w);
    if (fl & SG_NOP) { bsync<NW>(); continue; }
    if (GB && (fl & SG_EACH)) {
      if (fl & SG_SET) seg_each_g<false, true, PD>(BLb, vecb, b0, s0, d0, cnt, db, ds, dd, offN, offT, offV, offD, j, 0.0);
      else if (fl & SG_T) seg_each_g<true, false, PD>(BLb, vecb, b0, s0, d0, cnt, db, ds, dd, offN, offT, offV, offD, j, acc);
      else seg_each_g<false, false, PD>(BLb, vecb, b0, s0, d0, cnt, db, ds, dd, offN, offT, offV, offD, j, acc);
      acc = 0.0;
    } else if ((fl & (SG_EACH | SG_IND)) == (SG_EACH | SG_IND)) {
      if (fl & SG_SET) seg_each2<false, true>(BLb, vecb, b0, s0, d0, cnt, db, ds, dd, offN, offT, offV, offD, j);
      else if (fl & SG_T) seg_each2<true, false>(BLb, vecb, b0, s0, d0, cnt, db, ds, dd, offN, offT, offV, offD, j);
      else seg_each2<false, false>(BLb, vecb, b0, s0, d0, cnt, db, ds, dd, offN, offT, offV, offD, j);
    } else if (fl & SG_EACH) {
      if (fl & SG_SET) seg_each<false, true>(BLb, vecb, b0, s0, d0, cnt, db, ds, dd, offN, offT, offV, offD, j, 0.0);
      else if (fl & SG_T) seg_each<true, false>(BLb, vecb, b0, s0, d0, cnt, db, ds, dd, offN, offT, offV, offD, j, acc);
      else seg_each<false, false>(BLb, vecb, b0, s0, d0, cnt, db, ds, dd, offN, offT, offV, offD, j, acc);
      acc = 0.0;
    } else if (GB && b0 == ni_off && db == 0 && (fl & SG_END)) {
      // the combine run of a split accumulation: every op is (-I) * partial, so the run is a plain vector sum -- the constant
      // block is not fetched from the slab
      if (j == 0) {
        double a = 0.0;
        for (int k = 0; k < cnt; k++) a += *reinterpret_cast<const double *>(vecb + s0 + k * ds + offD);
        double *o = reinterpret_cast<double *>(vecb + d0 + offD); *o = *o + a;
      }
      wave_order();
    } else {
      acc = (fl & SG_T) ? seg_run<true, GB>(BLb, vecb, b0, s0, cnt, db, ds, offN, offT, offV, acc)
                        : seg_run<false, GB>(BLb, vecb, b0, s0, cnt, db, ds, offN, offT, offV, acc);
      if (fl & SG_END) {
        const double sum = quad_sum(acc);
        acc = 0.0;
        if (j == 0) { double *o = reinterpret_cast<double *>(vecb + d0 + offD); *o = *o - sum; }
        wave_order();
      }
    }
    if (fl & SG_BAR) bsync<NW>();
#ifdef MPCQP_TIMING
    if (trace && g - g0 < 60) { trace[2 + 2 * (g - g0)] = (long long)__builtin_amdgcn_s_memtime(); trace[3 + 2 * (g - g0)] = ((long long)fl << 32) | cnt; }
#endif
  }
}

template <int NW>
__device__ __forceinline__ bool factorize_res(RCtx &cx) {
  const DevPlan &pl = *cx.pl; const DevRes &rs = *cx.rs; double *ws = cx.ws;
  const int wid = cx.wid, lane = cx.lane, tid = wid * WAVE + lane; constexpr int NT = NW * WAVE;
  const double *lb = ws + pl.o_l, *ub = ws + pl.o_u;
  const double *valA = ws + pl.o_ellA, *valAt = ws + pl.o_ellAt, *valP = ws + pl.o_ellP;
  double *T = ws + pl.o_T;
#ifdef MPCQP_TIMING
  unsigned long long f0 = __builtin_amdgcn_s_memtime();
#endif
  for (int i = tid; i < pl.mpad; i += NT) cx.W[i] = i < pl.m ? rho_of(lb[i], ub[i], cx.rho) : 0.0;
  for (long k = tid; k < (long)pl.nT * BLK; k += NT) T[k] = 0.0;
  bsync<NW>();
  {
    const double sigma = cx.st->sigma;
    const DevEll &E = pl.At;
    for (int c = wid; c < E.nchunks; c += NW) {
      double acc = 0.0;
      for (int s = E.chunk_off[c]; s < E.chunk_off[c + 1]; s++) {
        const unsigned e = (unsigned)s * WAVE + lane;
        const double v = valAt[e];
        if (E.flag[e]) acc += cx.W[E.idx[e]] * v * v;
      }
      const int t = c * WAVE + lane;
      if (t < pl.npad) cx.R[t] = pl.perm[t] >= 0 ? sigma + acc : 1.0;
    }
  }
  {
    const DevEll &E = pl.A;
    for (int c = wid; c < E.nchunks; c += NW) {
      const int i = c * WAVE + lane;
      const double sr = sqrt(cx.W[i]);
      for (int s = E.chunk_off[c]; s < E.chunk_off[c + 1]; s++) {
        const unsigned e = (unsigned)s * WAVE + lane;
        const int tp = pl.tpos[e];
        if (tp >= 0) T[tp] = valA[e] * sr;
      }
    }
  }
  bsync<NW>();
#ifdef MPCQP_TIMING
  unsigned long long f1 = __builtin_amdgcn_s_memtime(); cx.fts[0] += f1 - f0;
#endif
  const int row0 = lane >> 4, col = lane & 15;
  for (int b = wid; b < pl.nblk; b += NW) {
    d4 acc = {0, 0, 0, 0};
    for (int g = pl.asm_ptr[b]; g < pl.asm_ptr[b + 1]; g++) acc = mfma_abt_l(T + (long)pl.asm_a[g] * BLK, T + (long)pl.asm_b[g] * BLK, acc, lane);
    const int J = pl.blk_diag[b];
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int pi = pl.asm_pidx[(long)b * BLK + g * WAVE + lane];
      if (pi >= 0) acc[g] += valP[pi];
      const int row = row0 + 4 * g;
      if (J >= 0 && row == col) acc[g] += cx.R[J * BS + row];
      cx.BL[(long)b * BLK + row * BS + col] = acc[g];
    }
  }
  bsync<NW>();
#ifdef MPCQP_TIMING
  unsigned long long f2 = __builtin_amdgcn_s_memtime(); cx.fts[1] += f2 - f1;
#endif
  // right-looking block LDL' by elimination-tree levels: G_K = S_KK^-1 (one wave per column of the level);
  // W_IK = S_IK G_K into temp tiles; S_IJ -= W_IK S_JK' (same-destination updates on one wave); the W tiles replace
  // the S_IK slots one phase later, once every update that still needs S_JK has read it.
  int nprev = 0, prev0 = 0;
  for (int lev = 0; lev < rs.nlev; lev++) {
    bool ok = true;
#ifdef MPCQP_TIMING
    const unsigned long long s0_ = __builtin_amdgcn_s_memtime();
#endif
    for (int ci = rs.lv_ptr[lev] + wid; ci < rs.lv_ptr[lev + 1]; ci += NW)
      ok = sweep_inverse(cx.BL + (long)rs.lv_diag[ci] * BLK, cx.RB + 16 * wid, lane) && ok;
#ifdef MPCQP_TIMING
    if (wid == 0) cx.fts[3] += __builtin_amdgcn_s_memtime() - s0_;
#endif
    for (int a = wid; a < nprev; a += NW)
      reinterpret_cast<d4 *>(cx.BL + (long)rs.lw_slot[prev0 + a] * BLK)[lane] = reinterpret_cast<const d4 *>(cx.TMP + (long)a * BLK)[lane];
    if (lane == 0) cx.RED[wid] = ok ? 1.0 : 0.0;
    bsync<NW>();
    bool all_ok = true;
    for (int w = 0; w < NW; w++) all_ok = all_ok && cx.RED[w] != 0.0;
    if (!all_ok) return false;
    const int w0 = rs.lw_ptr[lev], nwk = rs.lw_ptr[lev + 1] - w0;
    for (int a = wid; a < nwk; a += NW) {
      d4 acc = {0, 0, 0, 0};
      acc = mfma_abt_l(cx.BL + (long)rs.lw_slot[w0 + a] * BLK, cx.BL + (long)rs.lw_g[w0 + a] * BLK, acc, lane);
      double *t = cx.TMP + (long)a * BLK;
#pragma unroll
      for (int g = 0; g < 4; g++) t[(row0 + 4 * g) * BS + col] = acc[g];
    }
    bsync<NW>();
    for (int u = rs.lu_ptr[lev * NW + wid]; u < rs.lu_ptr[lev * NW + wid + 1]; u++) {
      d4 acc = {0, 0, 0, 0};
      acc = mfma_abt_l(cx.TMP + (long)rs.lu_tmp[u] * BLK, cx.BL + (long)rs.lu_b[u] * BLK, acc, lane);
      double *dst = cx.BL + (long)rs.lu_dst[u] * BLK;
#pragma unroll
      for (int g = 0; g < 4; g++) dst[(row0 + 4 * g) * BS + col] -= acc[g];
    }
    bsync<NW>();
    nprev = nwk; prev0 = w0;
  }
  for (int a = wid; a < nprev; a += NW)   // (the last level has no off-diagonal block; kept for generality)
    reinterpret_cast<d4 *>(cx.BL + (long)rs.lw_slot[prev0 + a] * BLK)[lane] = reinterpret_cast<const d4 *>(cx.TMP + (long)a * BLK)[lane];
  if (rs.nconst) {   // the constant block -I that folds the partial sums of a split run into their destination
    double *ni = cx.BL + (long)pl.nblk * BLK;
    for (int e = tid; e < BLK; e += NT) ni[e] = (e / BS == e % BS) ? -1.0 : 0.0;
  }
  for (int i = tid; i < pl.mpad; i += NT) cx.W[i] = cx.W[i] * cx.Z[i] - cx.Y[i];
  bsync<NW>();
#ifdef MPCQP_TIMING
  cx.fts[2] += __builtin_amdgcn_s_memtime() - f2;
#endif
  return true;
}

template <int NW>
__device__ __forceinline__ void update_info_res(RCtx &cx, Info &in) {
  const DevPlan &pl = *cx.pl; double *ws = cx.ws; const int wid = cx.wid, lane = cx.lane;
  const double *Dg = ws + pl.o_D, *Eg = ws + pl.o_E;
  const double *valA = ws + pl.o_ellA, *valAt = ws + pl.o_ellAt, *valP = ws + pl.o_ellP;
  const int unscale = cx.unscale;
  double v[15];
#pragma unroll
  for (int k = 0; k < 15; k++) v[k] = 0.0;
  // 0 pr 1 nz 2 nax 3 prs 4 nzs 5 naxs 6 dr 7 nq 8 naty 9 npx 10 drs 11 nqs 12 natys 13 npxs | 14 obj (sum)
  ell_rows_w<NW>(pl.A, valA, cx.X, wid, lane, [&](int i, double ax) {
    if (i < pl.m) {
      const double einv = unscale ? 1.0 / Eg[i] : 1.0, zi = cx.Z[i];
      v[0] = fmax(v[0], fabs(einv * (ax - zi))); v[2] = fmax(v[2], fabs(einv * ax)); v[1] = fmax(v[1], fabs(einv * zi));
      v[3] = fmax(v[3], fabs(ax - zi)); v[5] = fmax(v[5], fabs(ax)); v[4] = fmax(v[4], fabs(zi));
    }
  });
  // P x and A' y land on the same rows for a given wave (both chunked by wid), so no barrier is needed in between
  for (int c = wid; c < pl.P.nchunks; c += NW) {
    const int la = lane;
    const double px = ell_chunk<false>(valP, pl.P.idx, cx.X, pl.P.chunk_off[c], pl.P.chunk_off[c + 1], la);
    const double aty = ell_chunk<false>(valAt, pl.At.idx, cx.Y, pl.At.chunk_off[c], pl.At.chunk_off[c + 1], la);
    const int t = c * WAVE + lane;
    if (t < pl.npad) {
      const double dinv = unscale ? 1.0 / Dg[t] : 1.0, qv = cx.Q[t], du = qv + px + aty;
      v[6] = fmax(v[6], fabs(dinv * du)); v[7] = fmax(v[7], fabs(dinv * qv)); v[8] = fmax(v[8], fabs(dinv * aty)); v[9] = fmax(v[9], fabs(dinv * px));
      v[10] = fmax(v[10], fabs(du)); v[11] = fmax(v[11], fabs(qv)); v[12] = fmax(v[12], fabs(aty)); v[13] = fmax(v[13], fabs(px));
      v[14] += cx.X[t] * (0.5 * px + qv);
    }
  }
  block_combine<NW, 15, 1>(v, cx.RED, wid, lane);
  in.prim_res = uni(v[0]); in.nz = uni(v[1]); in.nax = uni(v[2]); in.prs = uni(v[3]); in.nzs = uni(v[4]); in.naxs = uni(v[5]);
  in.dual_res = uni(unscale ? cx.cinv * v[6] : v[6]); in.nq = uni(v[7]); in.naty = uni(v[8]); in.npx = uni(v[9]);
  in.drs = uni(v[10]); in.nqs = uni(v[11]); in.natys = uni(v[12]); in.npxs = uni(v[13]);
  in.obj = uni(cx.st->scaling ? cx.cinv * v[14] : v[14]);
}

template <int NW>
__device__ __forceinline__ bool primal_infeasible_res(RCtx &cx, double eps) {
  const DevPlan &pl = *cx.pl; double *ws = cx.ws; const int wid = cx.wid, lane = cx.lane, tid = wid * WAVE + lane; constexpr int NT = NW * WAVE;
  const double *lb = ws + pl.o_l, *ub = ws + pl.o_u, *Eg = ws + pl.o_E, *Dg = ws + pl.o_D, *dy = ws + pl.o_dy;
  double v[2] = {0.0, 0.0};   // 0 nrm (max) 1 lhs (sum)
  for (int i = tid; i < pl.mpad; i += NT) {
    double x = 0.0;
    if (i < pl.m) {
      x = dy[i];
      const double lo = lb[i], up = ub[i];
      if (up > Q_INFTY * Q_MIN_SCALING) { if (lo < -Q_INFTY * Q_MIN_SCALING) x = 0.0; else x = fmin(x, 0.0); }
      else if (lo < -Q_INFTY * Q_MIN_SCALING) x = fmax(x, 0.0);
      v[0] = fmax(v[0], fabs(cx.unscale ? Eg[i] * x : x));
      v[1] += up * fmax(x, 0.0) + lo * fmin(x, 0.0);
    }
    cx.W[i] = x;
  }
  block_combine<NW, 2, 1>(v, cx.RED, wid, lane);
  if (NW == 1) bsync<NW>();
  const double nrm = v[0], lhs = v[1];
  bool res = false;
  if (nrm > eps && lhs < -eps * nrm) {
    double a[1] = {0.0};
    ell_rows_w<NW>(pl.At, ws + pl.o_ellAt, cx.W, wid, lane, [&](int t, double x) { if (t < pl.npad) a[0] = fmax(a[0], fabs(cx.unscale ? (1.0 / Dg[t]) * x : x)); });
    block_combine<NW, 1, 0>(a, cx.RED, wid, lane);
    res = a[0] < eps * nrm;
  }
  bsync<NW>();
  for (int i = tid; i < pl.mpad; i += NT) cx.W[i] = i < pl.m ? rho_of(lb[i], ub[i], cx.rho) * cx.Z[i] - cx.Y[i] : 0.0;
  bsync<NW>();
  return res;
}

template <int NW>
__device__ __forceinline__ bool dual_infeasible_res(RCtx &cx, double eps) {
  const DevPlan &pl = *cx.pl; double *ws = cx.ws; const int wid = cx.wid, lane = cx.lane, tid = wid * WAVE + lane; constexpr int NT = NW * WAVE;
  const double *lb = ws + pl.o_l, *ub = ws + pl.o_u, *Eg = ws + pl.o_E, *Dg = ws + pl.o_D, *dx = ws + pl.o_dx;
  double v[2] = {0.0, 0.0};   // 0 nrm (max) 1 q'dx (sum)
  for (int t = tid; t < pl.npad; t += NT) {
    const double x = dx[t];
    cx.R[t] = x;
    v[0] = fmax(v[0], fabs(cx.unscale ? Dg[t] * x : x));
    v[1] += cx.Q[t] * x;
  }
  block_combine<NW, 2, 1>(v, cx.RED, wid, lane);
  if (NW == 1) bsync<NW>();
  const double nrm = v[0], qdx = v[1], cs = cx.unscale ? cx.c : 1.0;
  bool res = false;
  if (nrm > eps && qdx < -cs * eps * nrm) {
    double a[1] = {0.0};
    ell_rows_w<NW>(pl.P, ws + pl.o_ellP, cx.R, wid, lane, [&](int t, double x) { if (t < pl.npad) a[0] = fmax(a[0], fabs(cx.unscale ? (1.0 / Dg[t]) * x : x)); });
    block_combine<NW, 1, 0>(a, cx.RED, wid, lane);
    if (a[0] < cs * eps * nrm) {
      double bad[1] = {0.0};
      ell_rows_w<NW>(pl.A, ws + pl.o_ellA, cx.R, wid, lane, [&](int i, double x) {
        if (i < pl.m) {
          if (cx.unscale) x = (1.0 / Eg[i]) * x;
          if ((ub[i] < Q_INFTY * Q_MIN_SCALING && x > eps * nrm) || (lb[i] > -Q_INFTY * Q_MIN_SCALING && x < -eps * nrm)) bad[0] = 1.0;
        }
      });
      block_combine<NW, 1, 0>(bad, cx.RED, wid, lane);
      res = bad[0] == 0.0;
    }
  }
  bsync<NW>();
  return res;
}

template <int NW>
__device__ __forceinline__ int check_termination_res(RCtx &cx, Info &in, int approximate) {
  const mpcqp_settings &st = *cx.st;
  double eps_abs = st.eps_abs, eps_rel = st.eps_rel, epi = st.eps_prim_inf, edi = st.eps_dual_inf;
  if (in.prim_res > Q_INFTY || in.dual_res > Q_INFTY || in.prim_res != in.prim_res || in.dual_res != in.dual_res) { in.obj = NAN; return MPCQP_NON_CVX; }
  if (approximate) { eps_abs *= 10; eps_rel *= 10; epi *= 10; edi *= 10; }
  bool pc = false, dc = false, pic = false, dic = false;
  if (cx.pl->m == 0) pc = true;
  else {
    const double eps_prim = eps_abs + eps_rel * fmax(in.nz, in.nax);
    if (in.prim_res < eps_prim) pc = true; else pic = primal_infeasible_res<NW>(cx, epi);
  }
  {
    double mx = fmax(in.nq, fmax(in.naty, in.npx));
    if (cx.unscale) mx *= cx.cinv;
    const double eps_dual = eps_abs + eps_rel * mx;
    if (in.dual_res < eps_dual) dc = true; else dic = dual_infeasible_res<NW>(cx, edi);
  }
  if (pc && dc) return approximate ? MPCQP_SOLVED_INACCURATE : MPCQP_SOLVED;
  if (pic) { in.obj = Q_INFTY; return approximate ? MPCQP_PRIMAL_INFEASIBLE_INACCURATE : MPCQP_PRIMAL_INFEASIBLE; }
  if (dic) { in.obj = -Q_INFTY; return approximate ? MPCQP_DUAL_INFEASIBLE_INACCURATE : MPCQP_DUAL_INFEASIBLE; }
  return MPCQP_UNSOLVED;
}

// MINW = waves per SIMD the register allocation must leave room for: 1 when the LDS footprint allows only one QP per
// CU anyway (the kernel may then use the whole register file), 2 otherwise
// GB = the factor blocks stay in the per-QP HBM slab (factors that do not fit LDS); LDS then holds only the temp
// tiles, the ADMM vectors and the schedule, and the segment loops keep several blocks in flight.
template <int NW, int MINW, bool GB, bool REUSE, bool ZYG = false>
__global__ void __launch_bounds__(NW * WAVE, MINW) mpcqp_res_kernel(const DevPlan pl, const DevRes rs, const mpcqp_settings st, const DevIO io) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int NT = NW * WAVE;
  constexpr int SPD = MINW == 3 ? 8 : 6;       // factor blocks in flight per wave in the global-block segment loops
  constexpr int EU = 16;   // ELL slots in flight per lane in the two sweeps of every iteration (8 for the 128-VGPR instances: 0.5 % slower)
  const int tid = threadIdx.x, lane = tid & 63;
  const int b = __builtin_amdgcn_readfirstlane(io.order ? io.order[blockIdx.x] : (int)blockIdx.x);
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  RCtx cx;
  cx.pl = &pl; cx.rs = &rs; cx.st = &st; cx.wid = wid; cx.lane = lane;
  cx.fts[0] = cx.fts[1] = cx.fts[2] = cx.fts[3] = 0;
  double *ws = io.ws + (long)b * pl.ws_stride; cx.ws = ws;
  if (GB) { cx.BL = ws + pl.o_Lf; cx.TMP = lds; }
  else { cx.BL = lds; cx.TMP = cx.BL + ((long)pl.nblk + rs.nconst) * BLK; }
  cx.X = lds + rs.stage; cx.Q = cx.X + pl.npad; cx.R = cx.Q + pl.npad;
  double *rend = cx.R + pl.npad + rs.rext;      // the solve vector is followed by the per-wave partial sums of split runs
  if (GB && ZYG) {   // z and y are only ever indexed by their own row: in the slab they cost two coalesced streams per iteration and
    cx.Z = ws + pl.o_Zg; cx.Y = ws + pl.o_Yg; cx.W = rend;   // free 2 * mpad doubles of LDS (one more workgroup per CU for long horizons)
  } else { cx.Z = rend; cx.Y = cx.Z + pl.mpad; cx.W = cx.Y + pl.mpad; }
  cx.RB = cx.W + pl.mpad; cx.RED = cx.RB + 16 * NW + 16;
  int4 *segs = reinterpret_cast<int4 *>(cx.RED + 32 * NW);     // [2 * n_seg] schedule segments, then [NW + 1] list bounds
  int *lptr = reinterpret_cast<int *>(segs + 2 * rs.n_seg);
  double *valA = ws + pl.o_ellA, *valAt = ws + pl.o_ellAt, *valP = ws + pl.o_ellP;
  double *lb = ws + pl.o_l, *ub = ws + pl.o_u, *Dg = ws + pl.o_D, *Eg = ws + pl.o_E;
  const double *inP = io.P + (long)b * io.sP, *inA = io.A + (long)b * io.sA, *inq = io.q + (long)b * io.sq;
  const double *inl = io.l + (long)b * io.sl, *inu = io.u + (long)b * io.su;
  const int n = pl.n, m = pl.m, npad = pl.npad, mpad = pl.mpad;
  cx.unscale = st.scaling && !st.scaled_termination;

  TS_DECL;
  for (int k = tid; k < 2 * rs.n_seg; k += NT) segs[k] = reinterpret_cast<const int4 *>(rs.g_seg)[k];
  if (tid <= NW) lptr[tid] = rs.g_ptr[tid];
  double c = 1.0;
  int refactor = 1, prev_status = MPCQP_UNSOLVED;
  constexpr bool REUSE_T = REUSE;
  const bool reuse = REUSE_T && io.reuse;
  if (reuse) {
    // ---- kept workspace (mpcqp_update_vectors; OSQP's osqp_update_data_vec): P, A, their scaling D, E, c, the factor and rho
    // stay from the previous solve of this instance; q, l, u are replaced and scaled with the kept D, E, c.  The factor is
    // rebuilt only when a row changed between loose / inequality / equality, because rho_i depends on that class.
    prev_status = io.status[b];
    c = io.cscale[b];
    for (int t = tid; t < npad; t += NT) { cx.Q[t] = 0.0; cx.R[t] = Dg[t]; }
    bsync<NW>();
    for (int j = tid; j < n; j += NT) cx.Q[pl.pos[j]] = inq[j];
    bsync<NW>();
    for (int t = tid; t < npad; t += NT) cx.Q[t] *= c * cx.R[t];
    double chg[1] = {0.0};
    for (int i = tid; i < mpad; i += NT) {
      const double ei = Eg[i];
      const double nl = i < m ? ei * fmax(inl[i], -Q_INFTY) : 0.0, nu = i < m ? ei * fmin(inu[i], Q_INFTY) : 0.0;
      if (i < m) {
        const double ol = lb[i], ou = ub[i];
        const int oc = (ol < -Q_INFTY * Q_MIN_SCALING && ou > Q_INFTY * Q_MIN_SCALING) ? 0 : (ou - ol < Q_RHO_TOL ? 2 : 1);
        const int nc = (nl < -Q_INFTY * Q_MIN_SCALING && nu > Q_INFTY * Q_MIN_SCALING) ? 0 : (nu - nl < Q_RHO_TOL ? 2 : 1);
        if (oc != nc) chg[0] = 1.0;
      }
      lb[i] = nl; ub[i] = nu;
    }
    block_combine<NW, 1, 0>(chg, cx.RED, wid, lane);
    refactor = chg[0] != 0.0;
    c = uni(c); cx.c = c; cx.cinv = uni(1.0 / c);
    if (!GB) {   // the factor blocks come back from the slab
      const double *src = ws + pl.o_Lf;
      for (long e = tid; e < ((long)pl.nblk + rs.nconst) * BLK; e += NT) lds[e] = src[e];
    }
    bsync<NW>();
  } else {
    // ---- load: caller's CSC values -> ELL arrays. The block region of LDS is idle until the factorisation, so the
    // ELL values of A, A', P live there for the whole scaling phase (host guarantees they fit) and are written to
    // the HBM slab once, already scaled.
    double *sA = GB ? valA : lds, *sAt = GB ? valAt : sA + pl.A.entries, *sP = GB ? valP : sAt + pl.At.entries;
    for (long e = tid; e < pl.A.entries; e += NT) { const int s = pl.A.src[e]; sA[e] = s >= 0 ? inA[s] : 0.0; }
    for (long e = tid; e < pl.At.entries; e += NT) { const int s = pl.At.src[e]; sAt[e] = s >= 0 ? inA[s] : 0.0; }
    for (long e = tid; e < pl.P.entries; e += NT) { const int s = pl.P.src[e]; sP[e] = s >= 0 ? inP[s] : 0.0; }
    for (int t = tid; t < npad; t += NT) { cx.Q[t] = 0.0; cx.R[t] = 1.0; }
    for (int i = tid; i < mpad; i += NT) cx.W[i] = 1.0;
    bsync<NW>();
    for (int j = tid; j < n; j += NT) cx.Q[pl.pos[j]] = inq[j];
    bsync<NW>();

    TS(0);
    // ---- modified Ruiz equilibration: D in R, E in W, temporaries in X / Z
    c = 1.0;
    for (int it = 0; it < st.scaling; it++) {
      for (int ch = wid; ch < pl.At.nchunks; ch += NW) {
        const int t = ch * WAVE + lane;
        const int la = lane;
        const double nA = ell_chunk<true>(sAt, pl.At.idx, cx.W, pl.At.chunk_off[ch], pl.At.chunk_off[ch + 1], la);
        const double nP = ell_chunk<true>(sP, pl.P.idx, cx.R, pl.P.chunk_off[ch], pl.P.chunk_off[ch + 1], la);
        if (t < npad) { const double dj = cx.R[t]; cx.X[t] = 1.0 / sqrt(limit_scaling(fmax(c * dj * nP, dj * nA))); }
      }
      ell_rowmax_w<NW>(pl.A, sA, cx.R, wid, lane, [&](int i, double v) { if (i < mpad) cx.Z[i] = 1.0 / sqrt(limit_scaling(cx.W[i] * v)); });
      bsync<NW>();
      for (int t = tid; t < npad; t += NT) cx.R[t] *= cx.X[t];
      for (int i = tid; i < mpad; i += NT) cx.W[i] *= cx.Z[i];
      bsync<NW>();
      double v[2] = {0.0, 0.0};   // 0 qn (max) 1 sum
      ell_rowmax_w<NW>(pl.P, sP, cx.R, wid, lane, [&](int t, double x) { if (t < npad) { v[1] += c * cx.R[t] * x; v[0] = fmax(v[0], fabs(c * cx.R[t] * cx.Q[t])); } });
      block_combine<NW, 2, 1>(v, cx.RED, wid, lane);
      const double ct = 1.0 / limit_scaling(fmax(v[1] / (double)n, limit_scaling(v[0])));
      c *= ct;
      bsync<NW>();
    }
    c = uni(c); cx.c = c; cx.cinv = uni(1.0 / c);
    TS(1);
    // scale and write out: A <- E A D, A' likewise, P <- c D P D (coalesced stores of whole 512 B slots)
    for (int ch = wid; ch < pl.A.nchunks; ch += NW) {
      const int i = ch * WAVE + lane; const double ei = cx.W[i];
      for (int s = pl.A.chunk_off[ch]; s < pl.A.chunk_off[ch + 1]; s++) { const unsigned e = (unsigned)s * WAVE + lane; valA[e] = sA[e] * (ei * cx.R[pl.A.idx[e]]); }
    }
    for (int ch = wid; ch < pl.At.nchunks; ch += NW) {
      const int t = ch * WAVE + lane; const double dj = t < npad ? cx.R[t] : 0.0;
      for (int s = pl.At.chunk_off[ch]; s < pl.At.chunk_off[ch + 1]; s++) { const unsigned e = (unsigned)s * WAVE + lane; valAt[e] = sAt[e] * (dj * cx.W[pl.At.idx[e]]); }
      for (int s = pl.P.chunk_off[ch]; s < pl.P.chunk_off[ch + 1]; s++) { const unsigned e = (unsigned)s * WAVE + lane; valP[e] = sP[e] * (c * dj * cx.R[pl.P.idx[e]]); }
    }
    bsync<NW>();
    for (int t = tid; t < npad; t += NT) { cx.Q[t] *= c * cx.R[t]; Dg[t] = cx.R[t]; }
    for (int i = tid; i < mpad; i += NT) {
      const double ei = cx.W[i];
      Eg[i] = ei;
      lb[i] = i < m ? ei * fmax(inl[i], -Q_INFTY) : 0.0;
      ub[i] = i < m ? ei * fmin(inu[i], Q_INFTY) : 0.0;
    }
  }
  for (int t = tid; t < npad; t += NT) cx.X[t] = 0.0;
  for (int i = tid; i < mpad; i += NT) { cx.Z[i] = 0.0; cx.Y[i] = 0.0; }
  bsync<NW>();
  if (st.warm_start && io.x0 && io.y0) {
    for (int j = tid; j < n; j += NT) { const int t = pl.pos[j]; cx.X[t] = io.x0[(long)b * n + j] * (1.0 / Dg[t]); }
    for (int i = tid; i < m; i += NT) cx.Y[i] = io.y0[(long)b * m + i] * (1.0 / Eg[i]) * c;
    bsync<NW>();
    ell_rows_w<NW>(pl.A, valA, cx.X, wid, lane, [&](int i, double ax) { if (i < m) cx.Z[i] = ax; });
    bsync<NW>();
  }
  // a kept factor belongs to the rho it was built with: that instance's final rho of the previous solve
  cx.rho = uni(reuse ? io.info[4L * b + 3] : fmin(fmax(io.rho0 && io.rho0[b] > 0.0 ? io.rho0[b] : st.rho, Q_RHO_MIN), Q_RHO_MAX));
  int status = MPCQP_UNSOLVED, iter_done = 0;
  Info in; memset(&in, 0, sizeof(in));
  TS(2);
  bool ok = !(reuse && prev_status == MPCQP_NON_CVX);
  if (ok && refactor) ok = factorize_res<NW>(cx);
  else if (ok) {   // kept factor: only w = rho z - y, which the factorisation leaves behind otherwise
    for (int i = tid; i < mpad; i += NT) cx.W[i] = i < m ? rho_of(lb[i], ub[i], cx.rho) * cx.Z[i] - cx.Y[i] : 0.0;
    bsync<NW>();
  }
  if (!ok) status = MPCQP_NON_CVX;
  TS(3);

  int interval = st.adaptive_rho_interval;
  if (st.adaptive_rho && interval == 0) interval = st.check_termination ? 4 * st.check_termination : 100;
  const double alpha = st.alpha, sigma = st.sigma;
  double *dxg = ws + pl.o_dx, *dyg = ws + pl.o_dy;
  int can_check = 0;
  const int sq0 = lptr[wid], sq1 = lptr[wid + 1];
  const int ni_off = rs.nconst ? pl.nblk * BLK * 8 : -1;      // byte offset of the constant -I block (split accumulation runs)
  if (ok) {
    int iter;
    for (iter = 1; iter <= st.max_iter; iter++) {
      ell_rows_w<NW, EU>(pl.At, valAt, cx.W, wid, lane, [&](int t, double v) { if (t < npad) cx.R[t] = sigma * cx.X[t] - cx.Q[t] + v; });
      for (int t = tid; t < rs.rext; t += NT) cx.R[npad + t] = 0.0;
      bsync<NW>();
      TS(4);
#ifdef MPCQP_TIMING
      long long *trace = (b == 0 && wid == 0 && iter == 3 && io.dbg) ? io.dbg + 16L * gridDim.x : nullptr;
      if (trace) trace[0] = (long long)__builtin_amdgcn_s_memtime();
      run_schedule<NW, GB, SPD>(segs, sq0, sq1, reinterpret_cast<const char *>(cx.BL), reinterpret_cast<char *>(cx.R), lane, ni_off, trace);
      if (trace) trace[1] = (long long)__builtin_amdgcn_s_memtime();
#else
      run_schedule<NW, GB, SPD>(segs, sq0, sq1, reinterpret_cast<const char *>(cx.BL), reinterpret_cast<char *>(cx.R), lane, ni_off);
#endif
      if (NW == 1) bsync<NW>();
      TS(5);
      can_check = st.check_termination && (iter % st.check_termination == 0);
      const int do_rho = st.adaptive_rho && interval && (iter % interval == 0);
      const int save = can_check || do_rho;
      {
        // ztilde = A xtilde fused with relaxation, projection onto [l, u], dual update and w = rho z - y.
        // l, u are fetched before the row sum is accumulated; rho_i and 1/rho_i are selected from the three values
        // the rho rule can produce (no per-row division).
        const double rho_eq = uni(Q_RHO_EQ * cx.rho), ri_min = 1.0 / Q_RHO_MIN, ri_eq = uni(1.0 / rho_eq), ri_in = uni(1.0 / cx.rho);
        for (int c = wid; c < pl.A.nchunks; c += NW) {
          const int i = c * WAVE + lane;
          const double lo = lb[i], up = ub[i];
          // z, y of this row as well when they live in the slab: their latency hides behind the row sum like that of l, u
          const double zo = (GB && ZYG && i < mpad) ? cx.Z[i] : 0.0, yp = (GB && ZYG && i < mpad) ? cx.Y[i] : 0.0;   // (the last chunk may run past mpad)
          const double zt = ell_chunk<false, EU>(valA, pl.A.idx, cx.R, pl.A.chunk_off[c], pl.A.chunk_off[c + 1], lane);
          if (i < m) {
            const bool loose = lo < -Q_INFTY * Q_MIN_SCALING && up > Q_INFTY * Q_MIN_SCALING, eq = up - lo < Q_RHO_TOL;
            const double rh = loose ? Q_RHO_MIN : (eq ? rho_eq : cx.rho), rinv = loose ? ri_min : (eq ? ri_eq : ri_in);
            const double zr = alpha * zt + (1.0 - alpha) * ((GB && ZYG) ? zo : cx.Z[i]), yo = (GB && ZYG) ? yp : cx.Y[i];
            const double zn = fmin(fmax(zr + rinv * yo, lo), up);
            const double dy = rh * (zr - zn), yn = yo + dy;
            cx.Z[i] = zn; cx.Y[i] = yn; cx.W[i] = rh * zn - yn;
            if (save) dyg[i] = dy;
          }
        }
      }
      bsync<NW>();     // every wave has finished reading xtilde (R) as the gather source before X/R move on
      for (int t = tid; t < npad; t += NT) {
        const double xo = cx.X[t], xn = alpha * cx.R[t] + (1.0 - alpha) * xo;
        cx.X[t] = xn;
        if (save) dxg[t] = xn - xo;
      }
      bsync<NW>();
      TS(6);
      iter_done = iter;
      if (can_check) {
        update_info_res<NW>(cx, in);
        status = check_termination_res<NW>(cx, in, 0);
        TS(7);
        if (status != MPCQP_UNSOLVED) break;
      }
      if (do_rho) {
        if (!can_check) update_info_res<NW>(cx, in);
        const double pr = in.prs / (fmax(in.nzs, in.naxs) + Q_DIV_TOL);
        const double dr = in.drs / (fmax(in.nqs, fmax(in.natys, in.npxs)) + Q_DIV_TOL);
        double rn = cx.rho * sqrt(pr / (dr + Q_DIV_TOL));
        rn = fmin(fmax(rn, Q_RHO_MIN), Q_RHO_MAX);
        if (rn > cx.rho * st.adaptive_rho_tolerance || rn < cx.rho / st.adaptive_rho_tolerance) {
          cx.rho = uni(rn);
          if (!factorize_res<NW>(cx)) { status = MPCQP_NON_CVX; break; }
        }
      }
    }
    if (iter > st.max_iter) iter_done = st.max_iter;
    if (status == MPCQP_UNSOLVED) {
      if (!can_check) { update_info_res<NW>(cx, in); status = check_termination_res<NW>(cx, in, 0); }
      if (status == MPCQP_UNSOLVED) { status = check_termination_res<NW>(cx, in, 1); if (status == MPCQP_UNSOLVED) status = MPCQP_MAX_ITER_REACHED; }
    }
  }
  const bool bad = status == MPCQP_PRIMAL_INFEASIBLE || status == MPCQP_PRIMAL_INFEASIBLE_INACCURATE ||
                   status == MPCQP_DUAL_INFEASIBLE || status == MPCQP_DUAL_INFEASIBLE_INACCURATE || status == MPCQP_NON_CVX;
  for (int j = tid; j < n; j += NT) { const int t = pl.pos[j]; io.x[(long)b * n + j] = bad ? NAN : Dg[t] * cx.X[t]; }
  for (int i = tid; i < m; i += NT) {
    io.y[(long)b * m + i] = bad ? NAN : cx.cinv * Eg[i] * cx.Y[i];
    io.z[(long)b * m + i] = bad ? NAN : (1.0 / Eg[i]) * cx.Z[i];
  }
  if (!GB && io.keep) {   // resident variants: park the factor in the slab for a following mpcqp_update_vectors solve
    double *dst = ws + pl.o_Lf;
    for (long e = tid; e < ((long)pl.nblk + rs.nconst) * BLK; e += NT) dst[e] = lds[e];
  }
  if (tid == 0) {
    io.status[b] = status; io.iters[b] = iter_done;
    io.info[4L * b] = in.obj; io.info[4L * b + 1] = in.prim_res; io.info[4L * b + 2] = in.dual_res; io.info[4L * b + 3] = cx.rho;
    io.cscale[b] = c;
  }
  TS(8);
#ifdef MPCQP_TIMING
  ts_acc[12] = cx.fts[0]; ts_acc[13] = cx.fts[1]; ts_acc[14] = cx.fts[2]; ts_acc[15] = cx.fts[3];
#endif
  TS_STORE(io.dbg);
}

// block-primitive self test (mpcqp_debug_blockops)
extern "C" __global__ void __launch_bounds__(WAVE) mpcqp_blockops_kernel(const double *A, const double *B, const double *C, double *S,
                                                                          double *out_gemm, double *Sb, int *fail) {
  __shared__ double t0[BS * 17], t1[BS * 17];
  const int lane = threadIdx.x, row0 = lane >> 4, col = lane & 15;
  d4 prod = {0, 0, 0, 0};
  prod = mfma_abt(A, B, prod);
#pragma unroll
  for (int g = 0; g < 4; g++) out_gemm[(row0 + 4 * g) * BS + col] = C[(row0 + 4 * g) * BS + col] - prod[g];
  const bool ok = potrf_inv(S, Sb, t0, t1);
  if (lane == 0) *fail = ok ? 0 : 1;
}

// OSQP validates its data at setup and refuses a problem with l_i > u_i (OSQP_DATA_VALIDATION_ERROR: OsqpEigen's initSolver,
// reference src/sqp_solver/CuCaQP.cpp:183-197, returns false and nothing is solved).  Batched form of that refusal, run behind the
// solve kernel so that the hot kernels carry no extra state: an instance with crossed bounds reports MPCQP_UNSOLVED with 0
// iterations and NaN in x, y, z and the residuals (rho, info[3], is left for a kept workspace).  One wave per instance.
extern "C" __global__ void __launch_bounds__(256) mpcqp_validate_kernel(int batch, int n, int m, const double *__restrict__ l, long sl,
                                                                          const double *__restrict__ u, long su, double *x, double *y, double *z,
                                                                          int *status, int *iters, double *info) {
  const int lane = threadIdx.x & (WAVE - 1), b = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
  if (b >= batch) return;
  const double *lb = l + (long)b * sl, *ub = u + (long)b * su;
  int crossed = 0;
  for (int i = lane; i < m; i += WAVE) crossed |= lb[i] > ub[i];
  if (!__any(crossed)) return;
  for (int j = lane; j < n; j += WAVE) x[(long)b * n + j] = NAN;
  for (int i = lane; i < m; i += WAVE) { y[(long)b * m + i] = NAN; z[(long)b * m + i] = NAN; }
  if (lane == 0) { status[b] = MPCQP_UNSOLVED; iters[b] = 0; info[4L * b] = NAN; info[4L * b + 1] = NAN; info[4L * b + 2] = NAN; }
}

// Dispatch hint for the NEXT solve on a handle: instances ordered by descending iteration count of the solve that just
// finished (counting sort over iters / unit).  Instances are independent, so the order changes no result -- it only lets the
// long ones start first instead of wherever they sit in the batch: with one QP per workgroup and 25 / 50 / 75-iteration
// instances mixed, the in-order tail leaves CUs idle while the last long instance finishes (longest-processing-time-first
// scheduling; in an MPC loop consecutive solves of the same plants have correlated iteration counts).
__global__ void __launch_bounds__(1024) mpcqp_order_kernel(const int *__restrict__ iters, int *__restrict__ order, int batch, int unit) {
  constexpr int NB = 256;
  __shared__ int start[NB];
  const int tid = threadIdx.x;
  for (int k = tid; k < NB; k += blockDim.x) start[k] = 0;
  __syncthreads();
  for (int i = tid; i < batch; i += blockDim.x) atomicAdd(&start[min(max(iters[i], 0) / unit, NB - 1)], 1);
  __syncthreads();
  if (tid == 0) { int acc = 0; for (int k = NB - 1; k >= 0; k--) { const int c = start[k]; start[k] = acc; acc += c; } }
  __syncthreads();
  for (int i = tid; i < batch; i += blockDim.x) order[atomicAdd(&start[min(max(iters[i], 0) / unit, NB - 1)], 1)] = i;
}

// ------------------------------------------------------------------------------------------ host side
static thread_local std::string g_last_error;
static int fail(int code, const std::string &msg) { g_last_error = msg; return code; }
int mpcqp_set_error(int code, const std::string &msg) { return fail(code, msg); }
int mpcqp_pick_device(int requested, int *device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(MPCQP_ERR_NO_GPU, "hipGetDeviceCount found no device");
  int dev = requested;
  if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
  if (dev >= ndev) return fail(MPCQP_ERR_ARG, "device ordinal out of range");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return fail(MPCQP_ERR_HIP, "hipGetDeviceProperties failed");
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
    return fail(MPCQP_ERR_NO_GPU, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
  *device = dev;
  return MPCQP_OK;
}
#define HIPCHK(expr)                                                                              \
  do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(MPCQP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

struct mpcqp_handle {
  int n = 0, m = 0, batch = 0, device = 0;
  mpcqp_settings st;
  Plan plan; WsLayout wl; long lds = 0;
  int variant = 0;              // 0 = streaming (1 wave / QP), NW > 0 = LDS-resident factor with NW waves / QP
  bool wide = false;            // resident kernel instance that may use the whole register file (one QP per CU)
  bool gblocks = false;         // multi-wave LDL' kernel with the factor blocks streamed from the HBM slab
  bool zyg = false;             // ... with z, y in the slab instead of LDS (lifts workgroups per CU for long horizons)
  bool occ3 = false;            // ... its 168-VGPR instance (exactly 3 workgroups per CU fit in LDS), 8 blocks in flight
  bool occ4 = false;            // ... its 128-VGPR instance (>= 3 workgroups per CU fit in LDS)
  ResPlan rplan; DevRes dres;
  DevPlan dp; DevIO io;
  std::vector<void *> dev_allocs;
  double *ws = nullptr;
  double *dP = nullptr, *dq = nullptr, *dA = nullptr, *dl = nullptr, *du = nullptr;  // owned copies (host-memory updates)
  double *dx0 = nullptr, *dy0 = nullptr, *drho0 = nullptr;
  bool keep = false, have_factor = false, reuse_next = false;
  int *order[2] = {nullptr, nullptr}; int order_cur = -1; bool lpt = true; hipEvent_t ev_order = nullptr;   // dispatch hint, double-buffered
  static constexpr int NPIPE = 8;
  hipStream_t pipe[NPIPE] = {};               // mpcqp_solve_host: compute streams, one per slice in flight
  hipStream_t pipe_copy = nullptr;            // ... and the one stream all host-to-device copies queue on, in slice order
  std::vector<hipEvent_t> pipe_ev;            // slice k's inputs have landed
  double *ox = nullptr, *oy = nullptr, *oz = nullptr, *oinfo = nullptr, *ocs = nullptr; int *ostatus = nullptr, *oiters = nullptr;
  long long *odbg = nullptr;
  bool have_data = false, solved = false;
  hipStream_t last_stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

template <class T>
static int upload(mpcqp_handle *h, const std::vector<T> &v, const T **out) {
  void *d = nullptr;
  size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  HIPCHK(hipMalloc(&d, bytes));
  h->dev_allocs.push_back(d);
  if (!v.empty()) HIPCHK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = (const T *)d;
  return MPCQP_OK;
}
static int upload_ell(mpcqp_handle *h, const Ell &e, DevEll *d) {
  d->nchunks = e.nchunks; d->entries = e.entries();
  int rc;
  if ((rc = upload(h, e.chunk_off, &d->chunk_off))) return rc;
  if ((rc = upload(h, e.idx, &d->idx))) return rc;
  if ((rc = upload(h, e.src, &d->src))) return rc;
  if ((rc = upload(h, e.flag, &d->flag))) return rc;
  return MPCQP_OK;
}
template <class T>
static int dalloc(mpcqp_handle *h, T **p, size_t count) {
  void *d = nullptr;
  HIPCHK(hipMalloc(&d, std::max<size_t>(count, 1) * sizeof(T)));
  h->dev_allocs.push_back(d);
  *p = (T *)d;
  return MPCQP_OK;
}

// the kernel instance a handle runs: waves per QP, register budget, factor location, and -- as its own instance so that the
// full-setup kernels carry no code for it -- the kept-workspace entry (mpcqp_update_vectors)
template <bool REUSE>
static const void *res_kernel_pick(const mpcqp_handle *h) {
  if (h->gblocks && h->zyg) return h->occ3 ? (const void *)mpcqp_res_kernel<4, 3, true, REUSE, true> : (const void *)mpcqp_res_kernel<4, 2, true, REUSE, true>;
  if (h->gblocks && h->occ3) return (const void *)mpcqp_res_kernel<4, 3, true, REUSE>;
  if (h->gblocks) return h->occ4 ? (const void *)mpcqp_res_kernel<4, 4, true, REUSE> : (const void *)mpcqp_res_kernel<4, 2, true, REUSE>;
  if (h->variant == 1) return (const void *)mpcqp_res_kernel<1, 2, false, REUSE>;
  if (h->variant == 8) return (const void *)mpcqp_res_kernel<8, 2, false, REUSE>;
  return h->wide ? (const void *)mpcqp_res_kernel<4, 1, false, REUSE> : (const void *)mpcqp_res_kernel<4, 2, false, REUSE>;
}
static const void *res_kernel_of(const mpcqp_handle *h, bool reuse) { return reuse ? res_kernel_pick<true>(h) : res_kernel_pick<false>(h); }

extern "C" {

void mpcqp_default_settings(mpcqp_settings *s) {
  if (!s) return;
  s->rho = 0.1; s->sigma = 1e-6; s->alpha = 1.6; s->eps_abs = 1e-3; s->eps_rel = 1e-3;
  s->eps_prim_inf = 1e-4; s->eps_dual_inf = 1e-4; s->adaptive_rho_tolerance = 5.0;
  s->max_iter = 10000; s->check_termination = 25; s->scaling = 10; s->adaptive_rho = 1;
  s->adaptive_rho_interval = 0; s->scaled_termination = 0; s->warm_start = 0; s->device = -1;
}

const char *mpcqp_strerror(int code) {
  static thread_local std::string buf;
  const char *base = "unknown error";
  switch (code) {
    case MPCQP_OK: base = "ok"; break;
    case MPCQP_ERR_ARG: base = "invalid argument"; break;
    case MPCQP_ERR_HIP: base = "HIP runtime error"; break;
    case MPCQP_ERR_NO_GPU: base = "no usable gfx950 GPU (this library has no CPU fallback)"; break;
    case MPCQP_ERR_STATE: base = "call order violated"; break;
    case MPCQP_ERR_LIMIT: base = "problem exceeds on-chip budget"; break;
  }
  buf = base;
  if (code != MPCQP_OK && !g_last_error.empty()) buf += ": " + g_last_error;
  return buf.c_str();
}

int mpcqp_create(int n, int m, int batch, const int *Pp, const int *Pi, const int *Ap, const int *Ai,
                 const mpcqp_settings *settings, mpcqp_handle **out) {
  if (!out) return fail(MPCQP_ERR_ARG, "out is null");
  *out = nullptr;
  if (n <= 0 || m < 0 || batch <= 0 || !Pp || !Pi || !Ap || !Ai) return fail(MPCQP_ERR_ARG, "Invalid dimensions.");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(MPCQP_ERR_NO_GPU, "hipGetDeviceCount found no device");
  mpcqp_handle *h = new mpcqp_handle();
  if (settings) h->st = *settings; else mpcqp_default_settings(&h->st);
  int dev = h->st.device;
  if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
  if (dev >= ndev) { delete h; return fail(MPCQP_ERR_ARG, "device ordinal out of range"); }
  h->device = dev;
  auto bail = [&](int rc) { mpcqp_destroy(h); return rc; };
  if (hipSetDevice(dev) != hipSuccess) return bail(fail(MPCQP_ERR_HIP, "hipSetDevice failed"));
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return bail(fail(MPCQP_ERR_HIP, "hipGetDeviceProperties failed"));
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
    return bail(fail(MPCQP_ERR_NO_GPU, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only"));
  h->n = n; h->m = m; h->batch = batch;
  h->plan = build_plan(n, m, Pp, Pi, Ap, Ai);
  if (!h->plan.error.empty()) return bail(fail(MPCQP_ERR_ARG, h->plan.error));
  const Plan &pl = h->plan;
  h->wl = ws_layout(pl);
  h->lds = lds_bytes(pl);
  // Variant: keep the factor in LDS when it fits. Small problems take one wave per QP (several QPs per CU);
  // larger ones that still fit take four waves per QP. MPCQP_VARIANT=stream|res1|res4|res8 overrides.
  {
    const long LDS_MAX = 160 * 1024;
    int want = -1;
    if (const char *e = getenv("MPCQP_VARIANT")) {
      std::string v(e);
      if (v == "stream") want = 0; else if (v == "res1") want = 1; else if (v == "res4") want = 4; else if (v == "res8") want = 8;
      else if (v == "gres4") { want = 4; h->gblocks = true; }
    }
    // candidate plans of the multi-wave kernels: ELL chunk widths padded to multiples of 4 (fewer load batches per chunk)
    // and the stage chain eliminated from both ends (two concurrent half-length chains)
    const bool twist = !getenv("MPCQP_NO_TWIST");
    Plan p1 = build_plan(n, m, Pp, Pi, Ap, Ai, -1, true);
    Plan p4 = build_plan(n, m, Pp, Pi, Ap, Ai, twist ? 2 : -1, true);
    if (!p1.error.empty() || p1.nblk > h->plan.nblk) p1 = h->plan;
    if (!p4.error.empty() || p4.nblk > h->plan.nblk) p4 = p1;
    const bool small_ok = pl.nblk < 4096 && pl.nb < 512;
    if (want < 0) {
      // measured on MI355X: one wave per QP with the factor in LDS when it is tiny; four waves per QP
      // with the factor in LDS when at least two QPs fit per CU; otherwise occupancy beats residency and the factor
      // blocks are streamed from the HBM slab by the same LDL' / segment machinery (several workgroups per CU)
      ResPlan r1 = build_res_plan(p1, 1), r4 = build_res_plan(p4, 4);
      const long l1 = lds_bytes_res(p1, r1), l4 = lds_bytes_res(p4, r4);
      // latency regime (measured, tools/graph_tick.py): when the whole batch is resident in one round of 4-wave workgroups
      // (two per CU by registers, one when the factor needs more than half the LDS), four waves per QP with the factor in
      // LDS finish a QP soonest (double integrator x256: 1.12 ms vs 1.41 ms with one wave per QP; quadrotor N=20 x256:
      // 1.21 ms vs 1.96 ms with the factor streamed from HBM)
      const long cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
      const long cap4 = (small_ok && l4 <= LDS_MAX) ? std::min<long>(LDS_MAX / l4, 2) : 0;
      if (cap4 > 0 && (long)batch <= cus * cap4) want = 4;
      else if (small_ok && l1 <= 40 * 1024) want = 1;
      else {
        if (small_ok && l4 <= 80 * 1024) want = 4;
        else if (small_ok && lds_bytes_res_gb(p4, build_res_plan(p4, 4, true), !getenv("MPCQP_NO_ZYG")) <= LDS_MAX) { want = 4; h->gblocks = true; }
        else want = 0;
      }
    }
    if (want > 0) { h->plan = want >= 2 ? p4 : p1; h->wl = ws_layout(h->plan); }
    if (want > 0) {
      h->rplan = build_res_plan(pl, want, h->gblocks);
      long need = h->gblocks ? lds_bytes_res_gb(pl, h->rplan) : lds_bytes_res(pl, h->rplan);
      if (h->gblocks && !getenv("MPCQP_NO_ZYG")) {
        // long horizons: with z and y in the slab one more workgroup fits per CU (2 -> 3 or 1 -> 2); measured on quadrotor N=50
        const long alt = lds_bytes_res_gb(pl, h->rplan, true);
        const long fit = LDS_MAX / need, fit_alt = std::min<long>(LDS_MAX / alt, 3);
        if (fit <= 2 && fit_alt > fit) { h->zyg = true; need = alt; }
      }
      h->occ4 = h->gblocks && !h->zyg && need <= 53 * 1024 && !getenv("MPCQP_GB_OCC2");
      // LDS between 40 and 53 KiB: three workgroups per CU fit, so the instance compiled for three waves per SIMD (168 VGPRs, no
      // spills, 8 blocks in flight) replaces the 128-VGPR one (cart-pole N=100: 92.9k -> 95.8k QP/s; at 32 KiB it loses, 589k -> 551k)
      h->occ3 = h->gblocks && need <= 53 * 1024 && (need > 40 * 1024 || h->zyg || getenv("MPCQP_GB_OCC3")) && !getenv("MPCQP_GB_OCC2");
      if (!small_ok || need > LDS_MAX) return bail(fail(MPCQP_ERR_LIMIT, "resident variant needs " + std::to_string(need) + " B of LDS"));
      h->lds = need;
      if (const char *pad = getenv("MPCQP_LDS_MIN")) h->lds = std::max<long>(h->lds, atol(pad));   // experiment: limit workgroups per CU
    }
    h->variant = want;
  }
  if (h->lds > 160 * 1024) return bail(fail(MPCQP_ERR_LIMIT, "LDS footprint " + std::to_string(h->lds) + " B exceeds 160 KiB per CU"));
  DevPlan &dp = h->dp;
  memset(&dp, 0, sizeof(dp));
  dp.n = n; dp.m = m; dp.npad = pl.npad; dp.mpad = pl.mpad; dp.nb = pl.nb; dp.nblk = pl.nblk; dp.nfac = (int)pl.fac.size(); dp.nT = pl.nT;
  int rc;
#define UP(expr) if ((rc = (expr))) return bail(rc)
  UP(upload_ell(h, pl.A, &dp.A)); UP(upload_ell(h, pl.At, &dp.At)); UP(upload_ell(h, pl.P, &dp.P));
  UP(upload(h, pl.pos, &dp.pos)); UP(upload(h, pl.perm, &dp.perm));
  UP(upload(h, pl.fwd_ops, &dp.fwd_ops)); UP(upload(h, pl.bwd_ops, &dp.bwd_ops)); UP(upload(h, pl.bwd_of, &dp.bwd_of));
  {
    std::vector<int4> f(pl.fac.size());
    for (size_t i = 0; i < f.size(); i++) f[i] = make_int4(pl.fac[i].type, pl.fac[i].dst, pl.fac[i].a, pl.fac[i].b);
    UP(upload(h, f, &dp.fac));
  }
  UP(upload(h, pl.tpos, &dp.tpos)); UP(upload(h, pl.asm_ptr, &dp.asm_ptr)); UP(upload(h, pl.asm_a, &dp.asm_a));
  UP(upload(h, pl.asm_b, &dp.asm_b)); UP(upload(h, pl.asm_pidx, &dp.asm_pidx)); UP(upload(h, pl.blk_diag, &dp.blk_diag));
  if (h->variant > 0) {
    const ResPlan &rp = h->rplan; DevRes &dr = h->dres;
    dr.nphase = rp.nphase; dr.ntemp = rp.ntemp; dr.nconst = rp.nconst; dr.rext = rp.rext;
    dr.nlev = rp.nlev;
    UP(upload(h, rp.lv_ptr, &dr.lv_ptr)); UP(upload(h, rp.lv_diag, &dr.lv_diag)); UP(upload(h, rp.lw_ptr, &dr.lw_ptr));
    UP(upload(h, rp.lw_slot, &dr.lw_slot)); UP(upload(h, rp.lw_g, &dr.lw_g)); UP(upload(h, rp.lu_ptr, &dr.lu_ptr));
    UP(upload(h, rp.lu_dst, &dr.lu_dst)); UP(upload(h, rp.lu_tmp, &dr.lu_tmp)); UP(upload(h, rp.lu_b, &dr.lu_b));
    UP(upload(h, rp.g_ptr, &dr.g_ptr)); UP(upload(h, rp.g_seg, &dr.g_seg)); dr.n_seg = (int)rp.g_seg.size() / 8; dr.stage = h->gblocks ? res_stage_doubles_gb(rp) : res_stage_doubles(pl, rp);
  }
  const WsLayout &w = h->wl;
  dp.o_ellA = w.ellA; dp.o_ellAt = w.ellAt; dp.o_ellP = w.ellP; dp.o_Lf = w.Lf; dp.o_Lb = w.Lb; dp.o_T = w.T;
  dp.o_l = w.l; dp.o_u = w.u; dp.o_D = w.D; dp.o_E = w.E; dp.o_dx = w.dx; dp.o_dy = w.dy; dp.o_Zg = w.Zg; dp.o_Yg = w.Yg; dp.ws_stride = w.stride;
  UP(dalloc(h, &h->ws, (size_t)w.stride * batch));
  UP(dalloc(h, &h->ox, (size_t)batch * n)); UP(dalloc(h, &h->oy, (size_t)batch * std::max(m, 1))); UP(dalloc(h, &h->oz, (size_t)batch * std::max(m, 1)));
  UP(dalloc(h, &h->oinfo, (size_t)batch * 4)); UP(dalloc(h, &h->ocs, (size_t)batch));
  UP(dalloc(h, &h->ostatus, (size_t)batch)); UP(dalloc(h, &h->oiters, (size_t)batch));
#ifdef MPCQP_TIMING
  UP(dalloc(h, &h->odbg, (size_t)batch * 16 + 128));
#endif
#undef UP
  h->wide = h->variant == 4 && h->lds > 80 * 1024;
  if (h->lds > 48 * 1024) {
    const void *fns[2] = {res_kernel_of(h, false), res_kernel_of(h, true)};
    if (h->variant == 0) fns[0] = fns[1] = h->lds > 40 * 1024 ? (const void *)mpcqp_admm_kernel<8> : (const void *)mpcqp_admm_kernel<4>;
    for (const void *fn : fns)
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds) != hipSuccess)
        return bail(fail(MPCQP_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed"));
  }
  if (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) return bail(fail(MPCQP_ERR_HIP, "hipEventCreate failed"));
  memset(&h->io, 0, sizeof(h->io));
  h->lpt = !getenv("MPCQP_NO_LPT");
  {   // dispatch-hint buffers up front: nothing is allocated inside mpcqp_solve, so a solve can be captured in a HIP graph
    int rc;
    if ((rc = dalloc(h, &h->order[0], (size_t)batch)) || (rc = dalloc(h, &h->order[1], (size_t)batch))) return bail(rc);
    if (hipEventCreateWithFlags(&h->ev_order, hipEventDisableTiming) != hipSuccess) return bail(fail(MPCQP_ERR_HIP, "hipEventCreate failed"));
  }
  *out = h;
  return MPCQP_OK;
}

static int stage(mpcqp_handle *h, double **own, const double *src, long stride, long width, const double **dst, long *dstride) {
  // host-memory update: copy into an owned device buffer
  size_t count = stride == 0 ? (size_t)width : (size_t)stride * (h->batch - 1) + width;
  if (!*own) { int rc = dalloc(h, own, (size_t)std::max<long>(width, 1) * h->batch); if (rc) return rc; }
  if (stride != 0 && stride != width) {
    for (int b = 0; b < h->batch; b++) HIPCHK(hipMemcpy(*own + (size_t)b * width, src + (size_t)b * stride, width * sizeof(double), hipMemcpyHostToDevice));
    *dstride = width;
  } else {
    HIPCHK(hipMemcpy(*own, src, count * sizeof(double), hipMemcpyHostToDevice));
    *dstride = stride;
  }
  *dst = *own;
  return MPCQP_OK;
}

int mpcqp_update(mpcqp_handle *h, const double *P, long sP, const double *q, long sq, const double *A, long sA,
                 const double *l, long sl, const double *u, long su, int mem) {
  if (!h) return fail(MPCQP_ERR_ARG, "null handle");
  if (!q || (h->plan.nnzP_in > 0 && !P) || (h->plan.nnzA_in > 0 && !A) || (h->m > 0 && (!l || !u))) return fail(MPCQP_ERR_ARG, "null data pointer");
  if (sP < 0 || sq < 0 || sA < 0 || sl < 0 || su < 0) return fail(MPCQP_ERR_ARG, "negative stride");
  if ((sP && sP < h->plan.nnzP_in) || (sq && sq < h->n) || (sA && sA < h->plan.nnzA_in) || (sl && sl < h->m) || (su && su < h->m))
    return fail(MPCQP_ERR_ARG, "stride smaller than the array it strides (dimension mismatch)");
  HIPCHK(hipSetDevice(h->device));
  DevIO &io = h->io;
  if (mem == MPCQP_MEM_DEVICE) {
    io.P = P; io.sP = sP; io.q = q; io.sq = sq; io.A = A; io.sA = sA; io.l = l; io.sl = sl; io.u = u; io.su = su;
  } else if (mem == MPCQP_MEM_HOST) {
    if (h->last_stream || h->solved) HIPCHK(hipStreamSynchronize(h->last_stream));
    int rc;
    if ((rc = stage(h, &h->dP, P ? P : q, sP, h->plan.nnzP_in, &io.P, &io.sP))) return rc;
    if ((rc = stage(h, &h->dq, q, sq, h->n, &io.q, &io.sq))) return rc;
    if ((rc = stage(h, &h->dA, A ? A : q, sA, h->plan.nnzA_in, &io.A, &io.sA))) return rc;
    if ((rc = stage(h, &h->dl, l ? l : q, sl, h->m, &io.l, &io.sl))) return rc;
    if ((rc = stage(h, &h->du, u ? u : q, su, h->m, &io.u, &io.su))) return rc;
  } else return fail(MPCQP_ERR_ARG, "mem must be MPCQP_MEM_HOST or MPCQP_MEM_DEVICE");
  h->have_data = true; h->reuse_next = false;
  return MPCQP_OK;
}

int mpcqp_warm_start(mpcqp_handle *h, const double *x0, const double *y0, int mem) {
  if (!h || !x0 || !y0) return fail(MPCQP_ERR_ARG, "null pointer");
  HIPCHK(hipSetDevice(h->device));
  if (mem == MPCQP_MEM_DEVICE) { h->io.x0 = x0; h->io.y0 = y0; return MPCQP_OK; }
  int rc;
  if (!h->dx0) { if ((rc = dalloc(h, &h->dx0, (size_t)h->batch * h->n))) return rc; if ((rc = dalloc(h, &h->dy0, (size_t)h->batch * std::max(h->m, 1)))) return rc; }
  HIPCHK(hipMemcpy(h->dx0, x0, (size_t)h->batch * h->n * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->dy0, y0, (size_t)h->batch * h->m * sizeof(double), hipMemcpyHostToDevice));
  h->io.x0 = h->dx0; h->io.y0 = h->dy0;
  return MPCQP_OK;
}

int mpcqp_set_dispatch_hint(mpcqp_handle *h, int enable) {
  if (!h) return fail(MPCQP_ERR_ARG, "null handle");
  h->lpt = enable != 0;
  if (!h->lpt) h->order_cur = -1;
  return MPCQP_OK;
}

int mpcqp_keep_workspace(mpcqp_handle *h, int enable) {
  if (!h) return fail(MPCQP_ERR_ARG, "null handle");
  if (enable && h->variant == 0) return fail(MPCQP_ERR_LIMIT, "the streaming kernel variant does not keep its workspace");
  h->keep = enable != 0;
  if (!h->keep) { h->have_factor = false; h->reuse_next = false; }
  return MPCQP_OK;
}

int mpcqp_update_vectors(mpcqp_handle *h, const double *q, long sq, const double *l, long sl, const double *u, long su, int mem) {
  if (!h) return fail(MPCQP_ERR_ARG, "null handle");
  if (!h->keep || !h->have_factor)
    return fail(MPCQP_ERR_STATE, "mpcqp_update_vectors needs mpcqp_keep_workspace(h, 1) and a completed mpcqp_update + mpcqp_solve before it");
  if (!q || (h->m > 0 && (!l || !u))) return fail(MPCQP_ERR_ARG, "null data pointer");
  if (sq < 0 || sl < 0 || su < 0 || (sq && sq < h->n) || (sl && sl < h->m) || (su && su < h->m)) return fail(MPCQP_ERR_ARG, "dimension mismatch: stride shorter than the array");
  HIPCHK(hipSetDevice(h->device));
  DevIO &io = h->io;
  if (mem == MPCQP_MEM_DEVICE) {
    io.q = q; io.sq = sq; io.l = l; io.sl = sl; io.u = u; io.su = su;
  } else if (mem == MPCQP_MEM_HOST) {
    if (h->last_stream || h->solved) HIPCHK(hipStreamSynchronize(h->last_stream));
    int rc;
    if ((rc = stage(h, &h->dq, q, sq, h->n, &io.q, &io.sq))) return rc;
    if ((rc = stage(h, &h->dl, l ? l : q, sl, h->m, &io.l, &io.sl))) return rc;
    if ((rc = stage(h, &h->du, u ? u : q, su, h->m, &io.u, &io.su))) return rc;
  } else return fail(MPCQP_ERR_ARG, "mem must be MPCQP_MEM_HOST or MPCQP_MEM_DEVICE");
  h->reuse_next = true;
  return MPCQP_OK;
}

int mpcqp_set_rho(mpcqp_handle *h, const double *rho0, int mem) {
  if (!h) return fail(MPCQP_ERR_ARG, "null handle");
  HIPCHK(hipSetDevice(h->device));
  if (!rho0) { h->io.rho0 = nullptr; return MPCQP_OK; }
  if (mem == MPCQP_MEM_DEVICE) { h->io.rho0 = rho0; return MPCQP_OK; }
  if (mem != MPCQP_MEM_HOST) return fail(MPCQP_ERR_ARG, "mem must be MPCQP_MEM_HOST or MPCQP_MEM_DEVICE");
  int rc;
  if (!h->drho0) { if ((rc = dalloc(h, &h->drho0, (size_t)h->batch))) return rc; }
  HIPCHK(hipMemcpy(h->drho0, rho0, (size_t)h->batch * sizeof(double), hipMemcpyHostToDevice));
  h->io.rho0 = h->drho0;
  return MPCQP_OK;
}

int mpcqp_solve(mpcqp_handle *h, void *stream) {
  if (!h) return fail(MPCQP_ERR_ARG, "null handle");
  if (!h->have_data) return fail(MPCQP_ERR_STATE, "Solver not initialized. Call mpcqp_update() first.");
  HIPCHK(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  DevIO io = h->io;
  io.x = h->ox; io.y = h->oy; io.z = h->oz; io.status = h->ostatus; io.iters = h->oiters; io.info = h->oinfo; io.ws = h->ws; io.cscale = h->ocs; io.dbg = h->odbg;
  io.reuse = h->reuse_next ? 1 : 0; io.keep = h->keep ? 1 : 0;
  io.order = (h->lpt && h->order_cur >= 0) ? h->order[h->order_cur] : nullptr;
  if (io.order && h->last_stream != s) HIPCHK(hipStreamWaitEvent(s, h->ev_order, 0));    // the hint was written on another stream
  HIPCHK(hipEventRecord(h->ev0, s));
  if (h->variant > 0) {
    void *args[] = {(void *)&h->dp, (void *)&h->dres, (void *)&h->st, (void *)&io};
    HIPCHK(hipLaunchKernel(res_kernel_of(h, io.reuse != 0), dim3(h->batch), dim3(h->variant * WAVE), args, (size_t)h->lds, s));
  }
  else if (h->lds > 40 * 1024 && !getenv("MPCQP_PD4")) hipLaunchKernelGGL(mpcqp_admm_kernel<8>, dim3(h->batch), dim3(WAVE), (size_t)h->lds, s, h->dp, h->st, io);
  else hipLaunchKernelGGL(mpcqp_admm_kernel<4>, dim3(h->batch), dim3(WAVE), (size_t)h->lds, s, h->dp, h->st, io);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(h->ev1, s));
  if (h->m > 0) {
    hipLaunchKernelGGL(mpcqp_validate_kernel, dim3((h->batch + 3) / 4), dim3(256), 0, s, h->batch, h->n, h->m, io.l, io.sl, io.u, io.su, io.x, io.y, io.z,
                       io.status, io.iters, io.info);
    HIPCHK(hipGetLastError());
  }
  if (h->lpt && h->batch > 1) {   // order of the next solve from this solve's iteration counts
    const int nxt = h->order_cur == 0 ? 1 : 0;
    hipLaunchKernelGGL(mpcqp_order_kernel, dim3(1), dim3(1024), 0, s, (const int *)h->oiters, h->order[nxt], h->batch, std::max(1, h->st.check_termination));
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->ev_order, s));
    h->order_cur = nxt;
  }
  h->last_stream = s; h->solved = true; h->have_factor = h->keep;
  return MPCQP_OK;
}

// launch of instances [b0, b0 + count) on stream s: every per-instance pointer of `io` is advanced, the kernels index by blockIdx
static int launch_slice(mpcqp_handle *h, DevIO io, int b0, int count, hipStream_t s) {
  const long n = h->n, m = h->m;
  io.P += (long)b0 * io.sP; io.q += (long)b0 * io.sq; io.A += (long)b0 * io.sA; io.l += (long)b0 * io.sl; io.u += (long)b0 * io.su;
  if (io.x0) io.x0 += b0 * n;
  if (io.y0) io.y0 += b0 * m;
  if (io.rho0) io.rho0 += b0;
  io.x += b0 * n; io.y += b0 * m; io.z += b0 * m; io.status += b0; io.iters += b0; io.info += 4L * b0;
  io.ws += (long)b0 * h->dp.ws_stride; io.cscale += b0;
  if (io.dbg) io.dbg += 16L * b0;
  io.order = nullptr;
  if (h->variant > 0) {
    void *args[] = {(void *)&h->dp, (void *)&h->dres, (void *)&h->st, (void *)&io};
    HIPCHK(hipLaunchKernel(res_kernel_of(h, false), dim3(count), dim3(h->variant * WAVE), args, (size_t)h->lds, s));
  }
  else if (h->lds > 40 * 1024 && !getenv("MPCQP_PD4")) hipLaunchKernelGGL(mpcqp_admm_kernel<8>, dim3(count), dim3(WAVE), (size_t)h->lds, s, h->dp, h->st, io);
  else hipLaunchKernelGGL(mpcqp_admm_kernel<4>, dim3(count), dim3(WAVE), (size_t)h->lds, s, h->dp, h->st, io);
  HIPCHK(hipGetLastError());
  if (m > 0) {
    hipLaunchKernelGGL(mpcqp_validate_kernel, dim3((count + 3) / 4), dim3(256), 0, s, count, h->n, h->m, io.l, io.sl, io.u, io.su, io.x, io.y, io.z,
                       io.status, io.iters, io.info);
    HIPCHK(hipGetLastError());
  }
  return MPCQP_OK;
}

int mpcqp_solve_host(mpcqp_handle *h, const double *P, long sP, const double *q, long sq, const double *A, long sA,
                     const double *l, long sl, const double *u, long su,
                     double *x, double *y, int *status, int *iters, int chunks) {
  if (!h) return fail(MPCQP_ERR_ARG, "null handle");
  if (!q || (h->plan.nnzP_in > 0 && !P) || (h->plan.nnzA_in > 0 && !A) || (h->m > 0 && (!l || !u))) return fail(MPCQP_ERR_ARG, "null data pointer");
  const long wP = h->plan.nnzP_in, wA = h->plan.nnzA_in, n = h->n, m = h->m;
  if ((sP && sP != wP) || sq != n || (sA && sA != wA) || (m > 0 && (sl != m || su != m)))
    return fail(MPCQP_ERR_ARG, "dimension mismatch: mpcqp_solve_host takes dense instance-major arrays (stride = width; 0 shares P or A)");
  HIPCHK(hipSetDevice(h->device));
  if (h->last_stream || h->solved) HIPCHK(hipStreamSynchronize(h->last_stream));
  chunks = std::max(1, std::min(chunks > 0 ? chunks : 6, h->batch));
  int rc;
  const size_t B = h->batch;
  if (!h->dP) { if ((rc = dalloc(h, &h->dP, (size_t)std::max<long>(wP, 1) * B))) return rc; }
  if (!h->dq) { if ((rc = dalloc(h, &h->dq, (size_t)n * B))) return rc; }
  if (!h->dA) { if ((rc = dalloc(h, &h->dA, (size_t)std::max<long>(wA, 1) * B))) return rc; }
  if (!h->dl) { if ((rc = dalloc(h, &h->dl, (size_t)std::max<long>(m, 1) * B))) return rc; }
  if (!h->du) { if ((rc = dalloc(h, &h->du, (size_t)std::max<long>(m, 1) * B))) return rc; }
  // a slice's kernel ends with a tail (its slowest instance); several slices in flight fill each other's tails
  const int ns = std::min(chunks, mpcqp_handle::NPIPE);
  for (int i = 0; i < ns; i++) if (!h->pipe[i]) HIPCHK(hipStreamCreateWithFlags(&h->pipe[i], hipStreamNonBlocking));
  DevIO io = h->io;
  io.P = h->dP; io.sP = sP; io.q = h->dq; io.sq = n; io.A = h->dA; io.sA = sA; io.l = h->dl; io.sl = m; io.u = h->du; io.su = m;
  io.x = h->ox; io.y = h->oy; io.z = h->oz; io.status = h->ostatus; io.iters = h->oiters; io.info = h->oinfo; io.ws = h->ws; io.cscale = h->ocs; io.dbg = h->odbg;
  io.reuse = 0; io.keep = h->keep ? 1 : 0; io.order = nullptr;
  if (sP == 0 && wP) HIPCHK(hipMemcpy(h->dP, P, wP * sizeof(double), hipMemcpyHostToDevice));      // shared matrices: once
  if (sA == 0 && wA) HIPCHK(hipMemcpy(h->dA, A, wA * sizeof(double), hipMemcpyHostToDevice));
  if (!h->pipe_copy) HIPCHK(hipStreamCreateWithFlags(&h->pipe_copy, hipStreamNonBlocking));
  while ((int)h->pipe_ev.size() < chunks) { hipEvent_t e; HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); h->pipe_ev.push_back(e); }
  for (int c = 0; c < chunks; c++) {
    const int b0 = (int)((long)h->batch * c / chunks), b1 = (int)((long)h->batch * (c + 1) / chunks), cnt = b1 - b0;
    if (cnt <= 0) continue;
    // inputs of all slices queue on ONE stream, in slice order, so that slice 0 is complete after 1/chunks of the transfer
    // (copies spread over several streams share the link and all finish late)
    hipStream_t cs = h->pipe_copy, s = h->pipe[c % ns];
    if (sP) HIPCHK(hipMemcpyAsync(h->dP + (size_t)b0 * wP, P + (size_t)b0 * wP, (size_t)cnt * wP * sizeof(double), hipMemcpyHostToDevice, cs));
    HIPCHK(hipMemcpyAsync(h->dq + (size_t)b0 * n, q + (size_t)b0 * n, (size_t)cnt * n * sizeof(double), hipMemcpyHostToDevice, cs));
    if (sA) HIPCHK(hipMemcpyAsync(h->dA + (size_t)b0 * wA, A + (size_t)b0 * wA, (size_t)cnt * wA * sizeof(double), hipMemcpyHostToDevice, cs));
    if (m) {
      HIPCHK(hipMemcpyAsync(h->dl + (size_t)b0 * m, l + (size_t)b0 * m, (size_t)cnt * m * sizeof(double), hipMemcpyHostToDevice, cs));
      HIPCHK(hipMemcpyAsync(h->du + (size_t)b0 * m, u + (size_t)b0 * m, (size_t)cnt * m * sizeof(double), hipMemcpyHostToDevice, cs));
    }
    HIPCHK(hipEventRecord(h->pipe_ev[c], cs));
    HIPCHK(hipStreamWaitEvent(s, h->pipe_ev[c], 0));
    if ((rc = launch_slice(h, io, b0, cnt, s))) return rc;
    if (x) HIPCHK(hipMemcpyAsync(x + (size_t)b0 * n, h->ox + (size_t)b0 * n, (size_t)cnt * n * sizeof(double), hipMemcpyDeviceToHost, s));
    if (y && m) HIPCHK(hipMemcpyAsync(y + (size_t)b0 * m, h->oy + (size_t)b0 * m, (size_t)cnt * m * sizeof(double), hipMemcpyDeviceToHost, s));
    if (status) HIPCHK(hipMemcpyAsync(status + b0, h->ostatus + b0, (size_t)cnt * sizeof(int), hipMemcpyDeviceToHost, s));
    if (iters) HIPCHK(hipMemcpyAsync(iters + b0, h->oiters + b0, (size_t)cnt * sizeof(int), hipMemcpyDeviceToHost, s));
  }
  HIPCHK(hipStreamSynchronize(h->pipe_copy));
  for (int i = 0; i < ns; i++) HIPCHK(hipStreamSynchronize(h->pipe[i]));
  h->io.P = io.P; h->io.sP = io.sP; h->io.q = io.q; h->io.sq = io.sq; h->io.A = io.A; h->io.sA = io.sA; h->io.l = io.l; h->io.sl = io.sl; h->io.u = io.u; h->io.su = io.su;
  h->have_data = true; h->reuse_next = false; h->solved = true; h->have_factor = h->keep; h->last_stream = nullptr; h->order_cur = -1;
  return MPCQP_OK;
}

int mpcqp_get(mpcqp_handle *h, double *x, double *y, double *z, int *status, int *iters, double *info, int mem) {
  if (!h) return fail(MPCQP_ERR_ARG, "null handle");
  if (!h->solved) return fail(MPCQP_ERR_STATE, "no solve has been issued");
  HIPCHK(hipSetDevice(h->device));
  hipMemcpyKind k = mem == MPCQP_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  hipStream_t s = h->last_stream;
  const size_t B = h->batch;
  if (x) HIPCHK(hipMemcpyAsync(x, h->ox, B * h->n * sizeof(double), k, s));
  if (y && h->m) HIPCHK(hipMemcpyAsync(y, h->oy, B * h->m * sizeof(double), k, s));
  if (z && h->m) HIPCHK(hipMemcpyAsync(z, h->oz, B * h->m * sizeof(double), k, s));
  if (status) HIPCHK(hipMemcpyAsync(status, h->ostatus, B * sizeof(int), k, s));
  if (iters) HIPCHK(hipMemcpyAsync(iters, h->oiters, B * sizeof(int), k, s));
  if (info) HIPCHK(hipMemcpyAsync(info, h->oinfo, B * 4 * sizeof(double), k, s));
  if (mem != MPCQP_MEM_DEVICE) HIPCHK(hipStreamSynchronize(s));
  return MPCQP_OK;
}

int mpcqp_sync(mpcqp_handle *h) {
  if (!h) return fail(MPCQP_ERR_ARG, "null handle");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->last_stream));
  return MPCQP_OK;
}

void mpcqp_destroy(mpcqp_handle *h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->solved) (void)hipStreamSynchronize(h->last_stream);
  for (void *p : h->dev_allocs) (void)hipFree(p);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->ev_order) (void)hipEventDestroy(h->ev_order);
  for (int i = 0; i < mpcqp_handle::NPIPE; i++) if (h->pipe[i]) (void)hipStreamDestroy(h->pipe[i]);
  if (h->pipe_copy) (void)hipStreamDestroy(h->pipe_copy);
  for (hipEvent_t e : h->pipe_ev) (void)hipEventDestroy(e);
  delete h;
}

int mpcqp_last_kernel_ms(mpcqp_handle *h, float *ms) {
  if (!h || !ms) return fail(MPCQP_ERR_ARG, "null pointer");
  if (!h->solved) return fail(MPCQP_ERR_STATE, "no solve has been issued");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipEventSynchronize(h->ev1));
  HIPCHK(hipEventElapsedTime(ms, h->ev0, h->ev1));
  return MPCQP_OK;
}

int mpcqp_plan_info(const mpcqp_handle *h, long *o) {
  if (!h || !o) return fail(MPCQP_ERR_ARG, "null pointer");
  const Plan &pl = h->plan;
  o[0] = h->n; o[1] = h->m; o[2] = h->batch; o[3] = pl.npad; o[4] = pl.mpad; o[5] = pl.nb; o[6] = pl.nblk; o[7] = h->lds;
  o[8] = h->wl.stride * 8; o[9] = pl.ordering; o[10] = pl.nnzP_triu; o[11] = pl.nnzA_in; o[12] = pl.nT; o[13] = (long)pl.fac.size();
  o[14] = pl.A.slots() + pl.At.slots() + pl.P.slots(); o[15] = h->gblocks ? 100 + h->variant : h->variant;
  return MPCQP_OK;
}

int mpcqp_debug_scaling(mpcqp_handle *h, int b, double *D, double *E, double *c) {
  if (!h || b < 0 || b >= h->batch) return fail(MPCQP_ERR_ARG, "bad instance index");
  if (!h->solved) return fail(MPCQP_ERR_STATE, "no solve has been issued");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->last_stream));
  const Plan &pl = h->plan;
  std::vector<double> Dp(pl.npad);
  const double *base = h->ws + (size_t)b * h->wl.stride;
  if (D) { HIPCHK(hipMemcpy(Dp.data(), base + h->wl.D, pl.npad * sizeof(double), hipMemcpyDeviceToHost)); for (int j = 0; j < h->n; j++) D[j] = Dp[pl.pos[j]]; }
  if (E && h->m) HIPCHK(hipMemcpy(E, base + h->wl.E, h->m * sizeof(double), hipMemcpyDeviceToHost));
  if (c) HIPCHK(hipMemcpy(c, h->ocs + b, sizeof(double), hipMemcpyDeviceToHost));
  return MPCQP_OK;
}

#ifdef MPCQP_TIMING
// timing build only: per-QP cycle counts of the 16 instrumented segments (copied to host)
int mpcqp_debug_timing(mpcqp_handle *h, long long *out) {
  if (!h || !out || !h->odbg) return fail(MPCQP_ERR_ARG, "no timing data");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->last_stream));
  HIPCHK(hipMemcpy(out, h->odbg, ((size_t)h->batch * 16 + 128) * sizeof(long long), hipMemcpyDeviceToHost));
  return MPCQP_OK;
}
#endif

int mpcqp_debug_blockops(const double *A, const double *B, const double *C, const double *S, double *out_gemm, double *out_linv, int *potrf_fail) {
  if (!A || !B || !C || !S || !out_gemm || !out_linv || !potrf_fail) return fail(MPCQP_ERR_ARG, "null pointer");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(MPCQP_ERR_NO_GPU, "no device");
  double *d = nullptr; int *df = nullptr;
  HIPCHK(hipMalloc((void **)&d, 6 * BLK * sizeof(double)));
  HIPCHK(hipMalloc((void **)&df, sizeof(int)));
  HIPCHK(hipMemcpy(d, A, BLK * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(d + BLK, B, BLK * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d + 2 * BLK, C, BLK * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(d + 3 * BLK, S, BLK * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(mpcqp_blockops_kernel, dim3(1), dim3(WAVE), 0, 0, d, d + BLK, d + 2 * BLK, d + 3 * BLK, d + 4 * BLK, d + 5 * BLK, df);
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out_gemm, d + 4 * BLK, BLK * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(out_linv, d + 3 * BLK, BLK * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(potrf_fail, df, sizeof(int), hipMemcpyDeviceToHost));
  (void)hipFree(d); (void)hipFree(df);
  return MPCQP_OK;
}

}  // extern "C"
