/*
 * osqp_oracle.c -- TEST INFRASTRUCTURE ONLY.  See osqp_oracle.h for what this restates and why.
 *
 * Section map (each cites what it follows):
 *   [S1] problem plumbing        reference CuCaQP.cpp:43-103,271-288 (argument order P,q,A,l,u; upper
 *                                triangle of P), CuCaQP.h:105-152 (CSC semantics)
 *   [S2] ordering + symbolic     OSQP builtin backend = QDLDL sparse LDL' with a fill-reducing ordering;
 *                                elimination tree / up-looking numeric follow the published LDL algorithm
 *                                (T. Davis, "Algorithm 849: a concise sparse Cholesky factorization package")
 *   [S3] Ruiz equilibration      OSQP paper section 5.1 (modified Ruiz with cost scaling), 10 passes
 *   [S4] rho vector              OSQP paper section 5.2 (equality rows get 1e3 * rho)
 *   [S5] ADMM iteration          OSQP paper Algorithm 1
 *   [S6] residuals/termination   OSQP paper sections 3.3-3.4 (unscaled residuals, infeasibility certs)
 *   [S7] adaptive rho            OSQP paper section 5.2 (rho estimate from normalised residual ratio)
 */
#include "osqp_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ small helpers */
static double dmax(double a, double b) { return a > b ? a : b; }
static double dmin(double a, double b) { return a < b ? a : b; }
static double norm_inf(const double *v, int n) {
  double r = 0.0;
  for (int i = 0; i < n; i++) { double a = fabs(v[i]); if (a > r) r = a; }
  return r;
}
static double limit_scaling(double v) {
  v = v < ORC_MIN_SCALING ? 1.0 : v;
  v = v > ORC_MAX_SCALING ? ORC_MAX_SCALING : v;
  return v;
}

void orc_default_settings(orc_settings *s) {
  s->rho = 0.1; s->sigma = 1e-6; s->alpha = 1.6;
  s->eps_abs = 1e-3; s->eps_rel = 1e-3;          /* SQPOptimizationSolver.cpp:83-84 */
  s->eps_prim_inf = 1e-4; s->eps_dual_inf = 1e-4;
  s->adaptive_rho_tolerance = 5.0;
  s->max_iter = 10000;                            /* SQPOptimizationSolver.cpp:85 */
  s->check_termination = 25; s->scaling = 10;
  s->adaptive_rho = 1; s->adaptive_rho_interval = 0;
  s->scaled_termination = 0; s->warm_start = 0; s->linsys = 0;
}

/* ------------------------------------------------------------------ [S1]/[S2] pattern */
struct orc_pattern {
  int n, m;
  /* upper-triangular P in CSC; Pmap[k] = index into the caller's value array */
  int *Pp, *Pi, *Pmap; int nnzP;
  /* A in CSC (caller's order, identity map) */
  int *Ap, *Ai; int nnzA;
  /* KKT = [[P + sigma I, A'], [A, -diag(1/rho)]], upper triangle, symmetrically permuted */
  int N;            /* n + m */
  int *perm, *pinv; /* perm[new] = old */
  int *Kp, *Ki; int nnzK;
  int *slotP;       /* [nnzP] slot of each P entry in K values */
  int *slotA;       /* [nnzA] */
  int *slotD;       /* [N]   slot of each diagonal entry */
  /* symbolic LDL' */
  int *Parent, *Lp; int nnzL;
};

typedef struct { int r, c, kind, idx; } trip_t; /* kind 0=P 1=A 2=diag */
static int trip_cmp(const void *a, const void *b) {
  const trip_t *x = (const trip_t *)a, *y = (const trip_t *)b;
  if (x->c != y->c) return x->c - y->c;
  if (x->r != y->r) return x->r - y->r;
  return x->kind - y->kind;
}

/* exact minimum-degree ordering on a bitset adjacency matrix (N is a few thousand at most) */
static void min_degree(int N, const trip_t *T, int nT, int *perm) {
  int W = (N + 63) / 64;
  uint64_t *adj = (uint64_t *)calloc((size_t)N * W, sizeof(uint64_t));
  char *done = (char *)calloc(N, 1);
  int *deg = (int *)calloc(N, sizeof(int));
  for (int t = 0; t < nT; t++) {
    int r = T[t].r, c = T[t].c;
    if (r == c) continue;
    adj[(size_t)r * W + c / 64] |= 1ull << (c % 64);
    adj[(size_t)c * W + r / 64] |= 1ull << (r % 64);
  }
  for (int v = 0; v < N; v++) {
    int d = 0;
    for (int w = 0; w < W; w++) d += __builtin_popcountll(adj[(size_t)v * W + w]);
    deg[v] = d;
  }
  for (int step = 0; step < N; step++) {
    int best = -1;
    for (int v = 0; v < N; v++)
      if (!done[v] && (best < 0 || deg[v] < deg[best])) best = v;
    perm[step] = best; done[best] = 1;
    uint64_t *av = adj + (size_t)best * W;
    for (int w = 0; w < W; w++) {
      uint64_t bits = av[w];
      while (bits) {
        int b = __builtin_ctzll(bits); bits &= bits - 1;
        int a = w * 64 + b;
        uint64_t *aa = adj + (size_t)a * W;
        int d = 0;
        for (int k = 0; k < W; k++) aa[k] |= av[k];
        aa[best / 64] &= ~(1ull << (best % 64));
        aa[a / 64] &= ~(1ull << (a % 64));
        for (int k = 0; k < W; k++) d += __builtin_popcountll(aa[k]);
        deg[a] = d;
      }
    }
    /* remove eliminated node from everyone (already cleared in neighbours) */
  }
  free(adj); free(done); free(deg);
}

orc_pattern *orc_pattern_create(int n, int m, const int *Pp, const int *Pi, const int *Ap, const int *Ai) {
  orc_pattern *pt = (orc_pattern *)calloc(1, sizeof(orc_pattern));
  pt->n = n; pt->m = m; pt->N = n + m;
  int N = pt->N;
  /* upper triangle of P */
  int cnt = 0;
  for (int j = 0; j < n; j++) for (int k = Pp[j]; k < Pp[j + 1]; k++) if (Pi[k] <= j) cnt++;
  pt->nnzP = cnt;
  pt->Pp = (int *)malloc((n + 1) * sizeof(int));
  pt->Pi = (int *)malloc((cnt + 1) * sizeof(int));
  pt->Pmap = (int *)malloc((cnt + 1) * sizeof(int));
  cnt = 0;
  for (int j = 0; j < n; j++) {
    pt->Pp[j] = cnt;
    for (int k = Pp[j]; k < Pp[j + 1]; k++) if (Pi[k] <= j) { pt->Pi[cnt] = Pi[k]; pt->Pmap[cnt] = k; cnt++; }
  }
  pt->Pp[n] = cnt;
  pt->nnzA = Ap[n];
  pt->Ap = (int *)malloc((n + 1) * sizeof(int));
  pt->Ai = (int *)malloc((pt->nnzA + 1) * sizeof(int));
  memcpy(pt->Ap, Ap, (n + 1) * sizeof(int));
  memcpy(pt->Ai, Ai, pt->nnzA * sizeof(int));

  /* KKT triplets (unpermuted, upper) */
  int nT = pt->nnzP + pt->nnzA + N;
  trip_t *T = (trip_t *)malloc(nT * sizeof(trip_t));
  int t = 0;
  for (int j = 0; j < n; j++) for (int k = pt->Pp[j]; k < pt->Pp[j + 1]; k++) { T[t].r = pt->Pi[k]; T[t].c = j; T[t].kind = 0; T[t].idx = k; t++; }
  for (int j = 0; j < n; j++) for (int k = Ap[j]; k < Ap[j + 1]; k++) { T[t].r = j; T[t].c = n + Ai[k]; T[t].kind = 1; T[t].idx = k; t++; }
  for (int i = 0; i < N; i++) { T[t].r = i; T[t].c = i; T[t].kind = 2; T[t].idx = i; t++; }

  pt->perm = (int *)malloc(N * sizeof(int));
  pt->pinv = (int *)malloc(N * sizeof(int));
  min_degree(N, T, nT, pt->perm);
  for (int i = 0; i < N; i++) pt->pinv[pt->perm[i]] = i;

  for (int k = 0; k < nT; k++) {
    int r = pt->pinv[T[k].r], c = pt->pinv[T[k].c];
    if (r > c) { int s = r; r = c; c = s; }
    T[k].r = r; T[k].c = c;
  }
  qsort(T, nT, sizeof(trip_t), trip_cmp);
  pt->Kp = (int *)calloc(N + 1, sizeof(int));
  pt->Ki = (int *)malloc(nT * sizeof(int));
  pt->slotP = (int *)malloc((pt->nnzP + 1) * sizeof(int));
  pt->slotA = (int *)malloc((pt->nnzA + 1) * sizeof(int));
  pt->slotD = (int *)malloc(N * sizeof(int));
  int slot = -1, lr = -1, lc = -1;
  for (int k = 0; k < nT; k++) {
    if (T[k].r != lr || T[k].c != lc) { slot++; lr = T[k].r; lc = T[k].c; pt->Ki[slot] = lr; pt->Kp[lc + 1]++; }
    if (T[k].kind == 0) pt->slotP[T[k].idx] = slot;
    else if (T[k].kind == 1) pt->slotA[T[k].idx] = slot;
    else pt->slotD[T[k].idx] = slot;
  }
  pt->nnzK = slot + 1;
  for (int j = 0; j < N; j++) pt->Kp[j + 1] += pt->Kp[j];
  free(T);

  /* symbolic: elimination tree and column counts (LDL package, ldl_symbolic) */
  pt->Parent = (int *)malloc(N * sizeof(int));
  pt->Lp = (int *)malloc((N + 1) * sizeof(int));
  int *Lnz = (int *)calloc(N, sizeof(int)), *Flag = (int *)malloc(N * sizeof(int));
  for (int k = 0; k < N; k++) {
    pt->Parent[k] = -1; Flag[k] = k;
    for (int p = pt->Kp[k]; p < pt->Kp[k + 1]; p++) {
      int i = pt->Ki[p];
      if (i < k) for (; Flag[i] != k; i = pt->Parent[i]) {
        if (pt->Parent[i] == -1) pt->Parent[i] = k;
        Lnz[i]++; Flag[i] = k;
      }
    }
  }
  pt->Lp[0] = 0;
  for (int k = 0; k < N; k++) pt->Lp[k + 1] = pt->Lp[k] + Lnz[k];
  pt->nnzL = pt->Lp[N];
  free(Lnz); free(Flag);
  return pt;
}

void orc_pattern_destroy(orc_pattern *pt) {
  if (!pt) return;
  free(pt->Pp); free(pt->Pi); free(pt->Pmap); free(pt->Ap); free(pt->Ai);
  free(pt->perm); free(pt->pinv); free(pt->Kp); free(pt->Ki);
  free(pt->slotP); free(pt->slotA); free(pt->slotD); free(pt->Parent); free(pt->Lp);
  free(pt);
}
int orc_pattern_kkt_nnzL(const orc_pattern *pt) { return pt->nnzL; }

/* ------------------------------------------------------------------ per-QP workspace */
typedef struct {
  const orc_pattern *pt; const orc_settings *st;
  int n, m;
  double *P, *A, *q, *l, *u;           /* scaled data */
  double *D, *E, *Dinv, *Einv; double c, cinv;
  double *rho_vec, *rho_inv; int *ctype; double rho;
  double *x, *z, *y, *xprev, *zprev, *xt, *zt, *dx, *dy;
  double *Ax, *Px, *Aty, *tn, *tm;     /* products and temporaries */
  /* linsys 0 */
  double *Kx, *Lx, *Dg, *Dginv, *Y, *sol; int *Li, *Lnz, *Pattern, *Flag;
  /* linsys 1 */
  double *M;
  double prim_res, dual_res, obj;
  int status, iter;
} work_t;

static void spmv_A(const work_t *w, const double *x, double *out) { /* out = A x */
  const orc_pattern *pt = w->pt;
  for (int i = 0; i < w->m; i++) out[i] = 0.0;
  for (int j = 0; j < w->n; j++) { double xj = x[j]; for (int k = pt->Ap[j]; k < pt->Ap[j + 1]; k++) out[pt->Ai[k]] += w->A[k] * xj; }
}
static void spmv_At(const work_t *w, const double *y, double *out) { /* out = A' y */
  const orc_pattern *pt = w->pt;
  for (int j = 0; j < w->n; j++) { double s = 0.0; for (int k = pt->Ap[j]; k < pt->Ap[j + 1]; k++) s += w->A[k] * y[pt->Ai[k]]; out[j] = s; }
}
static void spmv_P(const work_t *w, const double *x, double *out) { /* out = P x, P symmetric from triu */
  const orc_pattern *pt = w->pt;
  for (int j = 0; j < w->n; j++) out[j] = 0.0;
  for (int j = 0; j < w->n; j++) for (int k = pt->Pp[j]; k < pt->Pp[j + 1]; k++) {
    int i = pt->Pi[k]; double v = w->P[k];
    out[i] += v * x[j];
    if (i != j) out[j] += v * x[i];
  }
}

/* [S3] modified Ruiz equilibration with cost normalisation, done in place on P, A, q like OSQP */
static void scale_data(work_t *w) {
  const orc_pattern *pt = w->pt; int n = w->n, m = w->m;
  for (int j = 0; j < n; j++) w->D[j] = 1.0;
  for (int i = 0; i < m; i++) w->E[i] = 1.0;
  w->c = 1.0;
  double *Dt = w->tn, *Et = w->tm;
  for (int it = 0; it < w->st->scaling; it++) {
    /* inf-norms of the columns of [[P, A'], [A, 0]] */
    for (int j = 0; j < n; j++) Dt[j] = 0.0;
    for (int i = 0; i < m; i++) Et[i] = 0.0;
    for (int j = 0; j < n; j++) for (int k = pt->Pp[j]; k < pt->Pp[j + 1]; k++) {
      int i = pt->Pi[k]; double a = fabs(w->P[k]);
      if (a > Dt[j]) Dt[j] = a;
      if (a > Dt[i]) Dt[i] = a;
    }
    for (int j = 0; j < n; j++) for (int k = pt->Ap[j]; k < pt->Ap[j + 1]; k++) {
      int i = pt->Ai[k]; double a = fabs(w->A[k]);
      if (a > Dt[j]) Dt[j] = a;
      if (a > Et[i]) Et[i] = a;
    }
    for (int j = 0; j < n; j++) Dt[j] = 1.0 / sqrt(limit_scaling(Dt[j]));
    for (int i = 0; i < m; i++) Et[i] = 1.0 / sqrt(limit_scaling(Et[i]));
    for (int j = 0; j < n; j++) for (int k = pt->Pp[j]; k < pt->Pp[j + 1]; k++) w->P[k] *= Dt[pt->Pi[k]] * Dt[j];
    for (int j = 0; j < n; j++) for (int k = pt->Ap[j]; k < pt->Ap[j + 1]; k++) w->A[k] *= Et[pt->Ai[k]] * Dt[j];
    for (int j = 0; j < n; j++) { w->q[j] *= Dt[j]; w->D[j] *= Dt[j]; }
    for (int i = 0; i < m; i++) w->E[i] *= Et[i];
    /* cost normalisation */
    for (int j = 0; j < n; j++) Dt[j] = 0.0;
    for (int j = 0; j < n; j++) for (int k = pt->Pp[j]; k < pt->Pp[j + 1]; k++) {
      int i = pt->Pi[k]; double a = fabs(w->P[k]);
      if (a > Dt[j]) Dt[j] = a;
      if (a > Dt[i]) Dt[i] = a;
    }
    double mean = 0.0;
    for (int j = 0; j < n; j++) mean += Dt[j];
    mean /= (double)n;
    double qn = limit_scaling(norm_inf(w->q, n));
    double ct = 1.0 / limit_scaling(dmax(mean, qn));
    for (int k = 0; k < pt->nnzP; k++) w->P[k] *= ct;
    for (int j = 0; j < n; j++) w->q[j] *= ct;
    w->c *= ct;
  }
  w->cinv = 1.0 / w->c;
  for (int j = 0; j < n; j++) w->Dinv[j] = 1.0 / w->D[j];
  for (int i = 0; i < m; i++) { w->Einv[i] = 1.0 / w->E[i]; w->l[i] *= w->E[i]; w->u[i] *= w->E[i]; }
}

/* [S4] */
static void set_rho_vec(work_t *w, int classify) {
  w->rho = dmin(dmax(w->rho, ORC_RHO_MIN), ORC_RHO_MAX);
  for (int i = 0; i < w->m; i++) {
    if (classify) {
      if (w->l[i] < -ORC_INFTY * ORC_MIN_SCALING && w->u[i] > ORC_INFTY * ORC_MIN_SCALING) w->ctype[i] = -1;
      else if (w->u[i] - w->l[i] < ORC_RHO_TOL) w->ctype[i] = 1;
      else w->ctype[i] = 0;
    }
    w->rho_vec[i] = w->ctype[i] < 0 ? ORC_RHO_MIN : (w->ctype[i] > 0 ? ORC_RHO_EQ_OVER_RHO_INEQ * w->rho : w->rho);
    w->rho_inv[i] = 1.0 / w->rho_vec[i];
  }
}

/* numeric factorisation; returns 0 ok, nonzero = not quasi-definite (non-convex) */
static int factor(work_t *w) {
  const orc_pattern *pt = w->pt; int n = w->n, m = w->m;
  if (w->st->linsys == 0) {
    int N = pt->N;
    memset(w->Kx, 0, pt->nnzK * sizeof(double));
    for (int k = 0; k < pt->nnzP; k++) w->Kx[pt->slotP[k]] += w->P[k];
    for (int k = 0; k < pt->nnzA; k++) w->Kx[pt->slotA[k]] += w->A[k];
    for (int j = 0; j < n; j++) w->Kx[pt->slotD[j]] += w->st->sigma;
    for (int i = 0; i < m; i++) w->Kx[pt->slotD[n + i]] += -w->rho_inv[i];
    /* up-looking LDL' (ldl_numeric) */
    int npos = 0;
    for (int k = 0; k < N; k++) {
      w->Y[k] = 0.0; int top = N; w->Flag[k] = k; w->Lnz[k] = 0;
      for (int p = pt->Kp[k]; p < pt->Kp[k + 1]; p++) {
        int i = pt->Ki[p];
        if (i <= k) {
          w->Y[i] += w->Kx[p];
          int len;
          for (len = 0; w->Flag[i] != k; i = pt->Parent[i]) { w->Pattern[len++] = i; w->Flag[i] = k; }
          while (len > 0) w->Pattern[--top] = w->Pattern[--len];
        }
      }
      w->Dg[k] = w->Y[k]; w->Y[k] = 0.0;
      for (; top < N; top++) {
        int i = w->Pattern[top]; double yi = w->Y[i]; w->Y[i] = 0.0;
        int p2 = pt->Lp[i] + w->Lnz[i], p;
        for (p = pt->Lp[i]; p < p2; p++) w->Y[w->Li[p]] -= w->Lx[p] * yi;
        double lki = yi / w->Dg[i];
        w->Dg[k] -= lki * yi;
        w->Li[p] = k; w->Lx[p] = lki; w->Lnz[i]++;
      }
      if (w->Dg[k] == 0.0 || w->Dg[k] != w->Dg[k]) return 1;
      if (w->Dg[k] > 0.0) npos++;
      w->Dginv[k] = 1.0 / w->Dg[k];
    }
    return npos == n ? 0 : 1;
  } else {
    double *M = w->M;
    memset(M, 0, (size_t)n * n * sizeof(double));
    for (int j = 0; j < n; j++) for (int k = pt->Pp[j]; k < pt->Pp[j + 1]; k++) { int i = pt->Pi[k]; M[(size_t)j * n + i] += w->P[k]; if (i != j) M[(size_t)i * n + j] += w->P[k]; }
    for (int j = 0; j < n; j++) M[(size_t)j * n + j] += w->st->sigma;
    /* A' R A : accumulate row by row using a CSR sweep built on the fly (columns pairs per row) */
    /* simple O(sum_i nnz_i^2) via CSC->row lists */
    int *rp = (int *)calloc(m + 1, sizeof(int));
    for (int k = 0; k < pt->nnzA; k++) rp[pt->Ai[k] + 1]++;
    for (int i = 0; i < m; i++) rp[i + 1] += rp[i];
    int *rc = (int *)malloc((pt->nnzA + 1) * sizeof(int)); double *rv = (double *)malloc((pt->nnzA + 1) * sizeof(double));
    int *fill = (int *)calloc(m, sizeof(int));
    for (int j = 0; j < n; j++) for (int k = pt->Ap[j]; k < pt->Ap[j + 1]; k++) { int i = pt->Ai[k]; int p = rp[i] + fill[i]++; rc[p] = j; rv[p] = w->A[k]; }
    for (int i = 0; i < m; i++) for (int a = rp[i]; a < rp[i + 1]; a++) for (int b = rp[i]; b < rp[i + 1]; b++)
      M[(size_t)rc[a] * n + rc[b]] += w->rho_vec[i] * rv[a] * rv[b];
    free(rp); free(rc); free(rv); free(fill);
    /* dense Cholesky, lower, row-major M[i*n+j] */
    for (int j = 0; j < n; j++) {
      double d = M[(size_t)j * n + j];
      for (int k = 0; k < j; k++) d -= M[(size_t)j * n + k] * M[(size_t)j * n + k];
      if (!(d > 0.0)) return 1;
      d = sqrt(d); M[(size_t)j * n + j] = d;
      for (int i = j + 1; i < n; i++) {
        double s = M[(size_t)i * n + j];
        for (int k = 0; k < j; k++) s -= M[(size_t)i * n + k] * M[(size_t)j * n + k];
        M[(size_t)i * n + j] = s / d;
      }
    }
    return 0;
  }
}

/* [S5] step 3 of Algorithm 1: solve the linear system for (xtilde, ztilde) */
static void solve_linsys(work_t *w) {
  const orc_pattern *pt = w->pt; int n = w->n, m = w->m;
  double sigma = w->st->sigma;
  if (w->st->linsys == 0) {
    int N = pt->N; double *b = w->sol;
    for (int j = 0; j < n; j++) b[pt->pinv[j]] = sigma * w->xprev[j] - w->q[j];
    for (int i = 0; i < m; i++) b[pt->pinv[n + i]] = w->zprev[i] - w->rho_inv[i] * w->y[i];
    for (int j = 0; j < N; j++) { int p2 = pt->Lp[j] + w->Lnz[j]; double bj = b[j]; for (int p = pt->Lp[j]; p < p2; p++) b[w->Li[p]] -= w->Lx[p] * bj; }
    for (int j = 0; j < N; j++) b[j] *= w->Dginv[j];
    for (int j = N - 1; j >= 0; j--) { int p2 = pt->Lp[j] + w->Lnz[j]; double s = b[j]; for (int p = pt->Lp[j]; p < p2; p++) s -= w->Lx[p] * b[w->Li[p]]; b[j] = s; }
    for (int j = 0; j < n; j++) w->xt[j] = b[pt->pinv[j]];
    for (int i = 0; i < m; i++) { double nu = b[pt->pinv[n + i]]; w->zt[i] = w->zprev[i] + w->rho_inv[i] * (nu - w->y[i]); }
  } else {
    for (int i = 0; i < m; i++) w->tm[i] = w->rho_vec[i] * w->zprev[i] - w->y[i];
    spmv_At(w, w->tm, w->xt);
    for (int j = 0; j < n; j++) w->xt[j] += sigma * w->xprev[j] - w->q[j];
    double *M = w->M;
    for (int i = 0; i < n; i++) { double s = w->xt[i]; for (int k = 0; k < i; k++) s -= M[(size_t)i * n + k] * w->xt[k]; w->xt[i] = s / M[(size_t)i * n + i]; }
    for (int i = n - 1; i >= 0; i--) { double s = w->xt[i]; for (int k = i + 1; k < n; k++) s -= M[(size_t)k * n + i] * w->xt[k]; w->xt[i] = s / M[(size_t)i * n + i]; }
    spmv_A(w, w->xt, w->zt);
  }
}

/* [S6] */
static void update_info(work_t *w) {
  int n = w->n, m = w->m; int unscale = w->st->scaling && !w->st->scaled_termination;
  spmv_A(w, w->x, w->Ax);
  spmv_P(w, w->x, w->Px);
  spmv_At(w, w->y, w->Aty);
  double r = 0.0;
  for (int i = 0; i < m; i++) { double v = w->Ax[i] - w->z[i]; if (unscale) v *= w->Einv[i]; v = fabs(v); if (v > r) r = v; }
  w->prim_res = r;
  r = 0.0;
  for (int j = 0; j < n; j++) { double v = w->q[j] + w->Px[j] + w->Aty[j]; if (unscale) v *= w->Dinv[j]; v = fabs(v); if (v > r) r = v; }
  w->dual_res = unscale ? w->cinv * r : r;
  double o = 0.0;
  for (int j = 0; j < n; j++) o += w->x[j] * (0.5 * w->Px[j] + w->q[j]);
  w->obj = w->st->scaling ? w->cinv * o : o;
}

static int is_primal_infeasible(work_t *w, double eps) {
  int n = w->n, m = w->m; int unscale = w->st->scaling && !w->st->scaled_termination;
  double *dy = w->tm;
  for (int i = 0; i < m; i++) {
    double v = w->dy[i];
    if (w->u[i] > ORC_INFTY * ORC_MIN_SCALING) { if (w->l[i] < -ORC_INFTY * ORC_MIN_SCALING) v = 0.0; else v = dmin(v, 0.0); }
    else if (w->l[i] < -ORC_INFTY * ORC_MIN_SCALING) v = dmax(v, 0.0);
    dy[i] = v;
  }
  double nrm = 0.0;
  for (int i = 0; i < m; i++) { double v = fabs(unscale ? w->E[i] * dy[i] : dy[i]); if (v > nrm) nrm = v; }
  if (nrm > eps) {
    double lhs = 0.0;
    for (int i = 0; i < m; i++) lhs += w->u[i] * dmax(dy[i], 0.0) + w->l[i] * dmin(dy[i], 0.0);
    if (lhs < -eps * nrm) {
      spmv_At(w, dy, w->tn);
      double a = 0.0;
      for (int j = 0; j < n; j++) { double v = fabs(unscale ? w->Dinv[j] * w->tn[j] : w->tn[j]); if (v > a) a = v; }
      return a < eps * nrm;
    }
  }
  return 0;
}

static int is_dual_infeasible(work_t *w, double eps) {
  int n = w->n, m = w->m; int unscale = w->st->scaling && !w->st->scaled_termination;
  double nrm = 0.0, cs = 1.0;
  if (unscale) { for (int j = 0; j < n; j++) { double v = fabs(w->D[j] * w->dx[j]); if (v > nrm) nrm = v; } cs = w->c; }
  else nrm = norm_inf(w->dx, n);
  if (nrm > eps) {
    double qdx = 0.0;
    for (int j = 0; j < n; j++) qdx += w->q[j] * w->dx[j];
    if (qdx < -cs * eps * nrm) {
      spmv_P(w, w->dx, w->tn);
      double a = 0.0;
      for (int j = 0; j < n; j++) { double v = fabs(unscale ? w->Dinv[j] * w->tn[j] : w->tn[j]); if (v > a) a = v; }
      if (a < cs * eps * nrm) {
        spmv_A(w, w->dx, w->tm);
        for (int i = 0; i < m; i++) {
          double v = unscale ? w->Einv[i] * w->tm[i] : w->tm[i];
          if ((w->u[i] < ORC_INFTY * ORC_MIN_SCALING && v > eps * nrm) || (w->l[i] > -ORC_INFTY * ORC_MIN_SCALING && v < -eps * nrm)) return 0;
        }
        return 1;
      }
    }
  }
  return 0;
}

static int check_termination(work_t *w, int approximate) {
  int n = w->n, m = w->m; const orc_settings *st = w->st;
  int unscale = st->scaling && !st->scaled_termination;
  double eps_abs = st->eps_abs, eps_rel = st->eps_rel, epi = st->eps_prim_inf, edi = st->eps_dual_inf;
  int pc = 0, dc = 0, pic = 0, dic = 0;
  if (w->prim_res > ORC_INFTY || w->dual_res > ORC_INFTY || w->prim_res != w->prim_res || w->dual_res != w->dual_res) { w->status = ORC_NON_CVX; w->obj = NAN; return 1; }
  if (approximate) { eps_abs *= 10; eps_rel *= 10; epi *= 10; edi *= 10; }
  if (m == 0) pc = 1;
  else {
    double nz = 0.0, na = 0.0;
    for (int i = 0; i < m; i++) { double s = unscale ? w->Einv[i] : 1.0; nz = dmax(nz, fabs(s * w->z[i])); na = dmax(na, fabs(s * w->Ax[i])); }
    double eps_prim = eps_abs + eps_rel * dmax(nz, na);
    if (w->prim_res < eps_prim) pc = 1; else pic = is_primal_infeasible(w, epi);
  }
  {
    double nq = 0.0, na = 0.0, np = 0.0;
    for (int j = 0; j < n; j++) { double s = unscale ? w->Dinv[j] : 1.0; nq = dmax(nq, fabs(s * w->q[j])); na = dmax(na, fabs(s * w->Aty[j])); np = dmax(np, fabs(s * w->Px[j])); }
    double mx = dmax(nq, dmax(na, np));
    if (unscale) mx *= w->cinv;
    double eps_dual = eps_abs + eps_rel * mx;
    if (w->dual_res < eps_dual) dc = 1; else dic = is_dual_infeasible(w, edi);
  }
  if (pc && dc) { w->status = approximate ? ORC_SOLVED_INACCURATE : ORC_SOLVED; return 1; }
  if (pic) { w->status = approximate ? ORC_PRIMAL_INFEASIBLE_INACCURATE : ORC_PRIMAL_INFEASIBLE; w->obj = ORC_INFTY; return 1; }
  if (dic) { w->status = approximate ? ORC_DUAL_INFEASIBLE_INACCURATE : ORC_DUAL_INFEASIBLE; w->obj = -ORC_INFTY; return 1; }
  return 0;
}

/* [S7]; uses Ax, Px, Aty left by update_info (scaled quantities, as OSQP does) */
static double rho_estimate(work_t *w) {
  int n = w->n, m = w->m;
  double pr = 0.0, dr = 0.0, nz = norm_inf(w->z, m), nax = norm_inf(w->Ax, m);
  for (int i = 0; i < m; i++) pr = dmax(pr, fabs(w->Ax[i] - w->z[i]));
  for (int j = 0; j < n; j++) dr = dmax(dr, fabs(w->q[j] + w->Px[j] + w->Aty[j]));
  pr /= (dmax(nz, nax) + ORC_DIVISION_TOL);
  dr /= (dmax(norm_inf(w->q, n), dmax(norm_inf(w->Aty, n), norm_inf(w->Px, n))) + ORC_DIVISION_TOL);
  double est = w->rho * sqrt(pr / (dr + ORC_DIVISION_TOL));
  return dmin(dmax(est, ORC_RHO_MIN), ORC_RHO_MAX);
}

/* reuse != 0: the workspace is kept from this instance's previous solve (osqp_update_data_vec semantics): P, A, D, E, c,
 * rho and the factor stay; q, l, u are replaced and scaled with the kept D, E, c; the factor is rebuilt only if a row's
 * class (loose / inequality / equality) changed, because rho_vec depends on it. */
static void solve_one(work_t *w, const double *Px, const double *q, const double *Ax, const double *l, const double *u,
                      const double *x0, const double *y0, double rho_start, int reuse) {
  const orc_pattern *pt = w->pt; const orc_settings *st = w->st; int n = w->n, m = w->m;
  if (reuse) {
    int prev_noncvx = (w->status == ORC_NON_CVX), changed = 0;
    for (int j = 0; j < n; j++) w->q[j] = w->c * w->D[j] * q[j];
    for (int i = 0; i < m; i++) { w->l[i] = w->E[i] * dmax(l[i], -ORC_INFTY); w->u[i] = w->E[i] * dmin(u[i], ORC_INFTY); }
    for (int i = 0; i < m; i++) {
      int ct = (w->l[i] < -ORC_INFTY * ORC_MIN_SCALING && w->u[i] > ORC_INFTY * ORC_MIN_SCALING) ? -1 : (w->u[i] - w->l[i] < ORC_RHO_TOL ? 1 : 0);
      if (ct != w->ctype[i]) changed = 1;
    }
    w->iter = 0; w->prim_res = w->dual_res = w->obj = 0.0;
    for (int j = 0; j < n; j++) w->x[j] = w->xprev[j] = w->dx[j] = 0.0;
    for (int i = 0; i < m; i++) w->z[i] = w->zprev[i] = w->y[i] = w->dy[i] = 0.0;
    if (prev_noncvx) return;
    w->status = ORC_UNSOLVED;
    if (changed) { set_rho_vec(w, 1); if (factor(w)) { w->status = ORC_NON_CVX; return; } }
  } else {
    /* [S1] load; clip bounds to +-OSQP_INFTY as osqp_setup does */
    for (int k = 0; k < pt->nnzP; k++) w->P[k] = Px[pt->Pmap[k]];
    for (int k = 0; k < pt->nnzA; k++) w->A[k] = Ax[k];
    for (int j = 0; j < n; j++) w->q[j] = q[j];
    for (int i = 0; i < m; i++) { w->l[i] = dmax(l[i], -ORC_INFTY); w->u[i] = dmin(u[i], ORC_INFTY); }
    w->c = w->cinv = 1.0;
    if (st->scaling) scale_data(w);
    else { for (int j = 0; j < n; j++) w->D[j] = w->Dinv[j] = 1.0; for (int i = 0; i < m; i++) w->E[i] = w->Einv[i] = 1.0; }
    w->rho = rho_start > 0.0 ? rho_start : st->rho;   /* a kept workspace carries rho over (osqp_update_* leave it alone) */
    set_rho_vec(w, 1);
    w->status = ORC_UNSOLVED; w->iter = 0; w->prim_res = w->dual_res = w->obj = 0.0;
    for (int j = 0; j < n; j++) w->x[j] = w->xprev[j] = w->dx[j] = 0.0;
    for (int i = 0; i < m; i++) w->z[i] = w->zprev[i] = w->y[i] = w->dy[i] = 0.0;
    if (factor(w)) { w->status = ORC_NON_CVX; return; }
  }
  if (st->warm_start && x0 && y0) { /* osqp_warm_start: scale x0 by Dinv, y0 by Einv * c, z = A x */
    for (int j = 0; j < n; j++) w->x[j] = x0[j] * w->Dinv[j];
    for (int i = 0; i < m; i++) w->y[i] = y0[i] * w->Einv[i] * w->c;
    spmv_A(w, w->x, w->z);
  }
  int interval = st->adaptive_rho_interval;
  if (st->adaptive_rho && interval == 0) interval = st->check_termination ? 4 * st->check_termination : 100;
  int iter, can_check = 0;
  for (iter = 1; iter <= st->max_iter; iter++) {
    double *t;
    t = w->x; w->x = w->xprev; w->xprev = t;
    t = w->z; w->z = w->zprev; w->zprev = t;
    solve_linsys(w);
    for (int j = 0; j < n; j++) { w->x[j] = st->alpha * w->xt[j] + (1.0 - st->alpha) * w->xprev[j]; w->dx[j] = w->x[j] - w->xprev[j]; }
    for (int i = 0; i < m; i++) {
      double zr = st->alpha * w->zt[i] + (1.0 - st->alpha) * w->zprev[i];
      double zn = dmin(dmax(zr + w->rho_inv[i] * w->y[i], w->l[i]), w->u[i]);
      w->z[i] = zn;
      w->dy[i] = w->rho_vec[i] * (zr - zn);
      w->y[i] += w->dy[i];
    }
    w->iter = iter;
    can_check = st->check_termination && (iter % st->check_termination == 0);
    if (can_check) { update_info(w); if (check_termination(w, 0)) break; }
    if (st->adaptive_rho && interval && (iter % interval == 0)) {
      if (!can_check) update_info(w);
      double rn = rho_estimate(w);
      if (rn > w->rho * st->adaptive_rho_tolerance || rn < w->rho / st->adaptive_rho_tolerance) {
        w->rho = rn; set_rho_vec(w, 0);
        if (factor(w)) { w->status = ORC_NON_CVX; return; }
      }
    }
  }
  if (iter > st->max_iter) w->iter = st->max_iter;
  if (!can_check) { update_info(w); check_termination(w, 0); }
  if (w->status == ORC_UNSOLVED) { if (!check_termination(w, 1)) w->status = ORC_MAX_ITER_REACHED; }
}

static work_t *work_alloc(const orc_pattern *pt, const orc_settings *st) {
  work_t *w = (work_t *)calloc(1, sizeof(work_t));
  int n = pt->n, m = pt->m, N = pt->N;
  w->pt = pt; w->st = st; w->n = n; w->m = m;
#define DV(k) ((double *)calloc((size_t)(k) + 1, sizeof(double)))
  w->P = DV(pt->nnzP); w->A = DV(pt->nnzA); w->q = DV(n); w->l = DV(m); w->u = DV(m);
  w->D = DV(n); w->E = DV(m); w->Dinv = DV(n); w->Einv = DV(m);
  w->rho_vec = DV(m); w->rho_inv = DV(m); w->ctype = (int *)calloc(m + 1, sizeof(int));
  w->x = DV(n); w->z = DV(m); w->y = DV(m); w->xprev = DV(n); w->zprev = DV(m); w->xt = DV(n); w->zt = DV(m); w->dx = DV(n); w->dy = DV(m);
  w->Ax = DV(m); w->Px = DV(n); w->Aty = DV(n); w->tn = DV(n); w->tm = DV(m);
  if (st->linsys == 0) {
    w->Kx = DV(pt->nnzK); w->Lx = DV(pt->nnzL); w->Dg = DV(N); w->Dginv = DV(N); w->Y = DV(N); w->sol = DV(N);
    w->Li = (int *)calloc(pt->nnzL + 1, sizeof(int)); w->Lnz = (int *)calloc(N + 1, sizeof(int));
    w->Pattern = (int *)calloc(N + 1, sizeof(int)); w->Flag = (int *)calloc(N + 1, sizeof(int));
  } else w->M = DV((size_t)n * n);
#undef DV
  return w;
}
static void work_free(work_t *w) {
  free(w->P); free(w->A); free(w->q); free(w->l); free(w->u); free(w->D); free(w->E); free(w->Dinv); free(w->Einv);
  free(w->rho_vec); free(w->rho_inv); free(w->ctype);
  free(w->x); free(w->z); free(w->y); free(w->xprev); free(w->zprev); free(w->xt); free(w->zt); free(w->dx); free(w->dy);
  free(w->Ax); free(w->Px); free(w->Aty); free(w->tn); free(w->tm);
  free(w->Kx); free(w->Lx); free(w->Dg); free(w->Dginv); free(w->Y); free(w->sol); free(w->Li); free(w->Lnz); free(w->Pattern); free(w->Flag); free(w->M);
  free(w);
}

static void store(work_t *w, int b, double *x, double *y, double *z, int *status, int *iters, double *info) {
  int n = w->n, m = w->m;
  int bad = (w->status == ORC_PRIMAL_INFEASIBLE || w->status == ORC_PRIMAL_INFEASIBLE_INACCURATE ||
             w->status == ORC_DUAL_INFEASIBLE || w->status == ORC_DUAL_INFEASIBLE_INACCURATE || w->status == ORC_NON_CVX);
  /* store_solution: x = D x, y = E y / c, z = Einv z; NaN when no solution is defined */
  if (x) for (int j = 0; j < n; j++) x[(size_t)b * n + j] = bad ? NAN : w->D[j] * w->x[j];
  if (y) for (int i = 0; i < m; i++) y[(size_t)b * m + i] = bad ? NAN : w->cinv * w->E[i] * w->y[i];
  if (z) for (int i = 0; i < m; i++) z[(size_t)b * m + i] = bad ? NAN : w->Einv[i] * w->z[i];
  if (status) status[b] = w->status;
  if (iters) iters[b] = w->iter;
  if (info) { info[4 * b] = w->obj; info[4 * b + 1] = w->prim_res; info[4 * b + 2] = w->dual_res; info[4 * b + 3] = w->rho; }
}

/* OSQP validates the data at setup and refuses a problem with l_i > u_i (osqp_setup returns OSQP_DATA_VALIDATION_ERROR, so
 * OsqpEigen's initSolver -- reference src/sqp_solver/CuCaQP.cpp:183-197 -- returns false and nothing is solved).  Batched form of
 * that refusal: the instance reports ORC_UNSOLVED with 0 iterations and NaN in x, y, z and the residuals. */
static void refuse_crossed_bounds(int n, int m, const double *l, const double *u, int b, double *x, double *y, double *z, int *status,
                                  int *iters, double *info) {
  int crossed = 0;
  for (int i = 0; i < m; i++) crossed |= l[i] > u[i];
  if (!crossed) return;
  if (x) for (int j = 0; j < n; j++) x[(size_t)b * n + j] = NAN;
  if (y) for (int i = 0; i < m; i++) y[(size_t)b * m + i] = NAN;
  if (z) for (int i = 0; i < m; i++) z[(size_t)b * m + i] = NAN;
  if (status) status[b] = ORC_UNSOLVED;
  if (iters) iters[b] = 0;
  if (info) info[4 * b] = info[4 * b + 1] = info[4 * b + 2] = NAN;
}

int orc_solve_batch(const orc_pattern *pt, const orc_settings *st, int batch,
                    const double *Px, long sP, const double *q, long sq, const double *Ax, long sA,
                    const double *l, long sl, const double *u, long su,
                    const double *x0, const double *y0,
                    double *x, double *y, double *z, int *status, int *iters, double *info, int nthreads) {
  return orc_solve_batch_rho(pt, st, batch, Px, sP, q, sq, Ax, sA, l, sl, u, su, x0, y0, NULL, x, y, z, status, iters, info, nthreads);
}

int orc_solve_batch_rho(const orc_pattern *pt, const orc_settings *st, int batch,
                        const double *Px, long sP, const double *q, long sq, const double *Ax, long sA,
                        const double *l, long sl, const double *u, long su,
                        const double *x0, const double *y0, const double *rho0,
                        double *x, double *y, double *z, int *status, int *iters, double *info, int nthreads) {
  if (!pt || !st || batch < 0) return 1;
  if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
  {
    work_t *w = work_alloc(pt, st);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
    for (int b = 0; b < batch; b++) {
      solve_one(w, Px + (size_t)b * sP, q + (size_t)b * sq, Ax + (size_t)b * sA, l + (size_t)b * sl, u + (size_t)b * su,
                x0 ? x0 + (size_t)b * pt->n : NULL, y0 ? y0 + (size_t)b * pt->m : NULL, rho0 ? rho0[b] : 0.0, 0);
      store(w, b, x, y, z, status, iters, info);
      refuse_crossed_bounds(pt->n, pt->m, l + (size_t)b * sl, u + (size_t)b * su, b, x, y, z, status, iters, info);
    }
    work_free(w);
  }
  return 0;
}

/* ---- kept workspaces: one per instance, alive across solves (what OsqpEigen's solver object is between updates) */
struct orc_state { const orc_pattern *pt; orc_settings st; int batch; work_t **w; int solved; };

orc_state *orc_state_create(const orc_pattern *pt, const orc_settings *st, int batch) {
  if (!pt || !st || batch <= 0) return NULL;
  orc_state *s = (orc_state *)calloc(1, sizeof(orc_state));
  s->pt = pt; s->st = *st; s->batch = batch; s->solved = 0;
  s->w = (work_t **)calloc((size_t)batch, sizeof(work_t *));
  for (int b = 0; b < batch; b++) s->w[b] = work_alloc(pt, &s->st);
  return s;
}

void orc_state_destroy(orc_state *s) {
  if (!s) return;
  for (int b = 0; b < s->batch; b++) work_free(s->w[b]);
  free(s->w); free(s);
}

int orc_state_solve(orc_state *s, int vectors_only,
                    const double *Px, long sP, const double *q, long sq, const double *Ax, long sA,
                    const double *l, long sl, const double *u, long su, const double *x0, const double *y0,
                    double *x, double *y, double *z, int *status, int *iters, double *info, int nthreads) {
  if (!s || (vectors_only && !s->solved)) return 1;
  if (nthreads < 1) nthreads = 1;
  const orc_pattern *pt = s->pt;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 4)
#endif
  for (int b = 0; b < s->batch; b++) {
    work_t *w = s->w[b];
    solve_one(w, vectors_only ? NULL : Px + (size_t)b * sP, q + (size_t)b * sq, vectors_only ? NULL : Ax + (size_t)b * sA,
              l + (size_t)b * sl, u + (size_t)b * su, x0 ? x0 + (size_t)b * pt->n : NULL, y0 ? y0 + (size_t)b * pt->m : NULL, 0.0, vectors_only);
    store(w, b, x, y, z, status, iters, info);
    refuse_crossed_bounds(pt->n, pt->m, l + (size_t)b * sl, u + (size_t)b * su, b, x, y, z, status, iters, info);
  }
  s->solved = 1;
  return 0;
}
