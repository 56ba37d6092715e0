// kernels_common.hpp -- constants, device-side plan / IO structs and the device helpers shared by every kernel family: wave reductions, ELL sweeps, block primitives (MFMA GEMM, 16x16 inverse), Ruiz scaling, termination and infeasibility tests of the streaming kernel
// Part of the single translation unit mpcqp.hip (included there in order; not a stand-alone header).
#pragma once


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
  int no_touch;        // on-chip mode, experiment switch (MPCQP_NO_TOUCH): no L2 prefetch by the idle waves
  int no_remap;        // on-chip mode, experiment switch (MPCQP_NO_REMAP): waves keep the parts their index gives them
  int *queue; int count;   // two-kernel on-chip mode: resident workgroups draw instance tickets 0 .. count - 1 from *queue (kernel_oc_split.hpp); NULL = one workgroup per instance
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
// workgroup barrier of the multi-wave kernels
template <int NW> __device__ __forceinline__ void bsync() { __syncthreads(); }

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
