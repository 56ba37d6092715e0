// reduced.hpp -- the opt-in reduced form (mpcqp_create_reduced): variables fixed by equality singleton rows are eliminated before the
// solve and restored after it.  Host-side maps + the presolve / postsolve kernels.
// Part of the single translation unit mpcqp.hip (included there in order; not a stand-alone header).
#pragma once

// The reference carries the parameters p as QP variables with rows  p <= p + dp <= p, i.e. dp = 0 (reference
// src/sqp_solver/SQPOptimizationSolver.cpp:47-60,117), and pins the first frame through lbx = ubx
// (src/OptimalControlProblem.cpp:93-96).  OSQP solves that QP as it stands; so does the default form here.  In the reduced form the caller
// names those rows (singleton rows of A with l = u in every instance); for each, x_j = l_i / a_ij is substituted:
//     q_free += P[free, fixed] x_fixed,   l, u of the kept rows -= A[kept, fixed] x_fixed,
// the smaller QP (no parameter block: the reduced KKT matrix loses its arrow) is solved by an ordinary handle on the reduced
// pattern, and x, y, z come back in the caller's dimensions: x_fixed as substituted, z_fixed = l, and the multiplier of an
// eliminated row from stationarity of its variable,  y_i = -(q_j + (P x)_j + sum_kept a_kj y_k) / a_ij.
// The result is a solution of the same QP to the same tolerances, from a different (shorter) ADMM run: iteration counts and the
// last digits differ from the full form, which stays the default.
struct RedMaps {
  int n = 0, m = 0, nr = 0, mr = 0, nfix = 0, nnzP = 0, nnzA = 0;
  std::vector<int> Ppr, Pir, Apr, Air;              // reduced patterns (CSC)
  std::vector<int> Psrc, Asrc;                      // reduced value -> index into the caller's value array
  std::vector<int> fix_var, fix_row, fix_src;       // [nfix] eliminated variable, its singleton row, index of a_ij in A's values
  std::vector<int> free_var, kept_row;              // [nr], [mr] reduced index -> original
  std::vector<int> var_of, row_of;                  // [n], [m] original -> reduced index (free / kept) or -1 - k (the k-th eliminated)
  // corrections, CSR by destination: q_r[jr] += P[src] * xfix[k];  shift[ir] += A[src] * xfix[k]
  std::vector<int> qc_ptr, qc_k, qc_src, lc_ptr, lc_k, lc_src;
  // postsolve, per eliminated variable: (P x)_j over all x: entries (variable, P src); A' y over kept rows: (reduced row, A src)
  std::vector<int> yp_ptr, yp_var, yp_src, ya_ptr, ya_row, ya_src;
  std::string error;
};

inline RedMaps build_red_maps(int n, int m, const int *Pp, const int *Pi, const int *Ap, const int *Ai, int nfixed, const int *fixed_rows) {
  RedMaps r; r.n = n; r.m = m; r.nnzP = Pp[n]; r.nnzA = Ap[n];
  std::vector<int> rowcnt(m, 0), rowcol(m, -1), rowsrc(m, -1);
  for (int j = 0; j < n; j++) for (int k = Ap[j]; k < Ap[j + 1]; k++) { rowcnt[Ai[k]]++; rowcol[Ai[k]] = j; rowsrc[Ai[k]] = k; }
  r.var_of.assign(n, 0); r.row_of.assign(m, 0);
  std::vector<char> vfix(n, 0), rfix(m, 0);
  for (int t = 0; t < nfixed; t++) {
    const int i = fixed_rows[t];
    if (i < 0 || i >= m) { r.error = "fixed row out of range"; return r; }
    if (rowcnt[i] != 1) { r.error = "a fixed row must have exactly one entry in A"; return r; }
    if (rfix[i] || vfix[rowcol[i]]) { r.error = "fixed row named twice, or two fixed rows on one variable"; return r; }
    rfix[i] = 1; vfix[rowcol[i]] = 1;
    r.var_of[rowcol[i]] = -1 - (int)r.fix_var.size(); r.row_of[i] = -1 - (int)r.fix_var.size();
    r.fix_var.push_back(rowcol[i]); r.fix_row.push_back(i); r.fix_src.push_back(rowsrc[i]);
  }
  r.nfix = (int)r.fix_var.size();
  for (int j = 0; j < n; j++) if (!vfix[j]) { r.var_of[j] = (int)r.free_var.size(); r.free_var.push_back(j); }
  for (int i = 0; i < m; i++) if (!rfix[i]) { r.row_of[i] = (int)r.kept_row.size(); r.kept_row.push_back(i); }
  r.nr = (int)r.free_var.size(); r.mr = (int)r.kept_row.size();
  if (r.nr == 0) { r.error = "every variable is fixed"; return r; }
  // reduced patterns + corrections
  std::vector<std::vector<std::pair<int, int>>> qc(r.nr), lc(r.mr), yp(r.nfix), ya(r.nfix);
  r.Ppr.assign(1, 0); r.Apr.assign(1, 0);
  for (int jr = 0; jr < r.nr; jr++) {
    const int j = r.free_var[jr];
    for (int k = Pp[j]; k < Pp[j + 1]; k++) { const int i = Pi[k]; if (r.var_of[i] >= 0) { r.Pir.push_back(r.var_of[i]); r.Psrc.push_back(k); } }
    r.Ppr.push_back((int)r.Pir.size());
    for (int k = Ap[j]; k < Ap[j + 1]; k++) { const int i = Ai[k]; if (r.row_of[i] >= 0) { r.Air.push_back(r.row_of[i]); r.Asrc.push_back(k); } }
    r.Apr.push_back((int)r.Air.size());
  }
  // P: only entries with row <= col count (mpcqp_create); an off-diagonal entry (i, j), i < j, stands for both (i, j) and (j, i)
  for (int j = 0; j < n; j++) for (int k = Pp[j]; k < Pp[j + 1]; k++) {
    const int i = Pi[k];
    if (i > j) continue;
    const int vi = r.var_of[i], vj = r.var_of[j];
    if (vi >= 0 && vj < 0) qc[vi].push_back({-1 - vj, k});
    if (i != j && vj >= 0 && vi < 0) qc[vj].push_back({-1 - vi, k});
    if (vj < 0) yp[-1 - vj].push_back({i, k});                       // (P x)_j += P_ij x_i
    if (i != j && vi < 0) yp[-1 - vi].push_back({j, k});             // (P x)_i += P_ij x_j
  }
  for (int j = 0; j < n; j++) if (r.var_of[j] < 0) for (int k = Ap[j]; k < Ap[j + 1]; k++) {
    const int i = Ai[k], kf = -1 - r.var_of[j];
    if (r.row_of[i] >= 0) { lc[r.row_of[i]].push_back({kf, k}); ya[kf].push_back({r.row_of[i], k}); }
  }
  auto flat = [](const std::vector<std::vector<std::pair<int, int>>> &v, std::vector<int> &ptr, std::vector<int> &a, std::vector<int> &b) {
    ptr.assign(1, 0);
    for (auto &l : v) { for (auto &e : l) { a.push_back(e.first); b.push_back(e.second); } ptr.push_back((int)a.size()); }
  };
  flat(qc, r.qc_ptr, r.qc_k, r.qc_src); flat(lc, r.lc_ptr, r.lc_k, r.lc_src);
  flat(yp, r.yp_ptr, r.yp_var, r.yp_src); flat(ya, r.ya_ptr, r.ya_row, r.ya_src);
  return r;
}

struct DevRed {
  int n, m, nr, mr, nfix, nnzPr, nnzAr;
  const int *Psrc, *Asrc, *fix_var, *fix_row, *fix_src, *free_var, *kept_row, *var_of, *row_of;
  const int *qc_ptr, *qc_k, *qc_src, *lc_ptr, *lc_k, *lc_src, *yp_ptr, *yp_var, *yp_src, *ya_ptr, *ya_row, *ya_src;
  double *Pr, *qr, *Ar, *lr, *ur, *xfix; int *bad;      // reduced data [batch x ...], the substituted values [batch x nfix], contract violations [batch]
};

// one workgroup per instance.  vectors_only: P, A values of the reduced pattern are in place already (kept workspace)
__global__ void __launch_bounds__(256) mpcqp_presolve_kernel(const DevRed rd, const DevIO io, int vectors_only) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const double *P = io.P + (long)b * io.sP, *A = io.A + (long)b * io.sA, *q = io.q + (long)b * io.sq, *l = io.l + (long)b * io.sl, *u = io.u + (long)b * io.su;
  double *xf = rd.xfix + (long)b * rd.nfix;
  __shared__ int bad;
  if (tid == 0) bad = 0;
  __syncthreads();
  for (int k = tid; k < rd.nfix; k += 256) {
    const double lo = l[rd.fix_row[k]], up = u[rd.fix_row[k]], a = A[rd.fix_src[k]];
    if (!(fabs(up - lo) <= 1e-9 * fmax(1.0, fabs(lo))) || a == 0.0) bad = 1;       // the caller's promise: an equality row with a nonzero entry
    xf[k] = lo / a;
  }
  if (!vectors_only) {
    double *Pr = rd.Pr + (long)b * rd.nnzPr, *Ar = rd.Ar + (long)b * rd.nnzAr;
    for (int e = tid; e < rd.nnzPr; e += 256) Pr[e] = P[rd.Psrc[e]];
    for (int e = tid; e < rd.nnzAr; e += 256) Ar[e] = A[rd.Asrc[e]];
  }
  __syncthreads();
  for (int jr = tid; jr < rd.nr; jr += 256) {
    double v = q[rd.free_var[jr]];
    for (int e = rd.qc_ptr[jr]; e < rd.qc_ptr[jr + 1]; e++) v += P[rd.qc_src[e]] * xf[rd.qc_k[e]];
    rd.qr[(long)b * rd.nr + jr] = v;
  }
  for (int ir = tid; ir < rd.mr; ir += 256) {
    double s = 0.0;
    for (int e = rd.lc_ptr[ir]; e < rd.lc_ptr[ir + 1]; e++) s += A[rd.lc_src[e]] * xf[rd.lc_k[e]];
    const double lo = l[rd.kept_row[ir]], up = u[rd.kept_row[ir]];
    rd.lr[(long)b * rd.mr + ir] = lo <= -Q_INFTY ? lo : lo - s;                 // an infinite bound stays infinite
    rd.ur[(long)b * rd.mr + ir] = up >= Q_INFTY ? up : up - s;
  }
  if (tid == 0) rd.bad[b] = bad;
}

// xr, yr, zr: the reduced handle's outputs; x, y, z, status...: the outer handle's
__global__ void __launch_bounds__(256) mpcqp_postsolve_kernel(const DevRed rd, const DevIO io, const double *xr, const double *yr, const double *zr, const int *str,
                                                              const int *itr, const double *infr, double *x, double *y, double *z, int *status, int *iters, double *info) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const double *P = io.P + (long)b * io.sP, *A = io.A + (long)b * io.sA, *q = io.q + (long)b * io.sq, *l = io.l + (long)b * io.sl;
  const double *xf = rd.xfix + (long)b * rd.nfix;
  const bool refused = rd.bad[b] != 0;
  double *xo = x + (long)b * rd.n, *yo = y + (long)b * rd.m, *zo = z + (long)b * rd.m;
  for (int j = tid; j < rd.n; j += 256) { const int v = rd.var_of[j]; xo[j] = refused ? NAN : (v >= 0 ? xr[(long)b * rd.nr + v] : xf[-1 - v]); }
  for (int i = tid; i < rd.m; i += 256) {
    const int v = rd.row_of[i];
    if (v >= 0) {
      double sh = 0.0;                                         // z = A x in the caller's rows: the substituted part comes back
      for (int e = rd.lc_ptr[v]; e < rd.lc_ptr[v + 1]; e++) sh += A[rd.lc_src[e]] * xf[rd.lc_k[e]];
      yo[i] = refused ? NAN : yr[(long)b * rd.mr + v]; zo[i] = refused ? NAN : zr[(long)b * rd.mr + v] + sh;
    }
    else zo[i] = refused ? NAN : l[i];
  }
  __syncthreads();      // xo is complete (global memory written by this workgroup, read back below)
  for (int k = tid; k < rd.nfix; k += 256) {
    const int j = rd.fix_row[k];      // the eliminated row
    // stationarity of the eliminated variable: q_j + (P x)_j + sum_kept a_kj y_k + a_ij y_i = 0
    double s = q[rd.fix_var[k]];
    for (int e = rd.yp_ptr[k]; e < rd.yp_ptr[k + 1]; e++) s += P[rd.yp_src[e]] * xo[rd.yp_var[e]];
    for (int e = rd.ya_ptr[k]; e < rd.ya_ptr[k + 1]; e++) s += A[rd.ya_src[e]] * yr[(long)b * rd.mr + rd.ya_row[e]];
    yo[j] = refused ? NAN : -s / A[rd.fix_src[k]];
  }
  if (tid == 0) {
    status[b] = refused ? MPCQP_UNSOLVED : str[b]; iters[b] = refused ? 0 : itr[b];
    for (int t = 0; t < 4; t++) info[4L * b + t] = infr[4L * b + t];
  }
}

// warm start of a reduced handle: the caller's x0 [n], y0 [m] restricted to the free variables / kept rows
__global__ void __launch_bounds__(256) mpcqp_red_gather_kernel(const DevRed rd, const double *x0, const double *y0, double *xr, double *yr) {
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int jr = tid; jr < rd.nr; jr += 256) xr[(long)b * rd.nr + jr] = x0[(long)b * rd.n + rd.free_var[jr]];
  for (int ir = tid; ir < rd.mr; ir += 256) yr[(long)b * rd.mr + ir] = y0[(long)b * rd.m + rd.kept_row[ir]];
}
