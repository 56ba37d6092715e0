/*
 * osqp_oracle.h -- TEST INFRASTRUCTURE ONLY (parity oracle + timed CPU baseline).
 *
 * CPU restatement, in plain C / fp64, of the algorithm the reference's hot path runs:
 *   SQPOptimizationSolver::getOptimalSolution  (reference src/sqp_solver/SQPOptimizationSolver.cpp:137-198)
 *     -> CuCaQP::setSystem / initSolver / solve (reference src/sqp_solver/CuCaQP.cpp:271-288,183-211)
 *       -> OsqpEigen 0.9.0 -> OSQP v1.0.0.beta1 osqp_setup + osqp_solve   (cpu_install.sh:4-6,34-44)
 * OSQP is an un-vendored third-party dependency of the reference (not under /root/reference), so its
 * published algorithm (Stellato et al., "OSQP: an operator splitting solver for quadratic programs",
 * Math. Prog. Comp. 2020) is restated here with the settings the reference fixes
 * (SQPOptimizationSolver.cpp:81-85: eps_abs = eps_rel = 1e-3, max_iter = 10000, warm start moot because
 * CuCaQP::setSystem clears the solver, i.e. every QP is a cold start) and OSQP defaults otherwise.
 *
 * PARITY UNPINNED at the QP boundary: the reference holds no machine-checkable (P,q,A,l,u)->x vectors
 * (SURVEY.md section 8c). The oracle is pinned instead on (i) the 7 convex known answers of the
 * reference's test/test.cpp:13-185 and (ii) KKT-verified high-accuracy optima (tests/golden/).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#ifndef OSQP_ORACLE_H
#define OSQP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_INFTY 1e30          /* OSQP_INFTY */
#define ORC_MIN_SCALING 1e-4    /* MIN_SCALING */
#define ORC_MAX_SCALING 1e4     /* MAX_SCALING */
#define ORC_RHO_MIN 1e-6
#define ORC_RHO_MAX 1e6
#define ORC_RHO_TOL 1e-4
#define ORC_RHO_EQ_OVER_RHO_INEQ 1e3
#define ORC_DIVISION_TOL 1e-10  /* guard used when normalising residuals for the rho estimate */

/* status codes follow OSQP 1.0's enum osqp_status_type */
enum {
  ORC_SOLVED = 1,
  ORC_SOLVED_INACCURATE = 2,
  ORC_PRIMAL_INFEASIBLE = 3,
  ORC_PRIMAL_INFEASIBLE_INACCURATE = 4,
  ORC_DUAL_INFEASIBLE = 5,
  ORC_DUAL_INFEASIBLE_INACCURATE = 6,
  ORC_MAX_ITER_REACHED = 7,
  ORC_NON_CVX = 9,
  ORC_UNSOLVED = 11
};

typedef struct {
  double rho;                  /* 0.1 */
  double sigma;                /* 1e-6 */
  double alpha;                /* 1.6 */
  double eps_abs;              /* reference: 1e-3 (SQPOptimizationSolver.cpp:83) */
  double eps_rel;              /* reference: 1e-3 (SQPOptimizationSolver.cpp:84) */
  double eps_prim_inf;         /* 1e-4 */
  double eps_dual_inf;         /* 1e-4 */
  double adaptive_rho_tolerance; /* 5 */
  int max_iter;                /* reference: 10000 (SQPOptimizationSolver.cpp:85) */
  int check_termination;       /* 25 */
  int scaling;                 /* 10 Ruiz passes */
  int adaptive_rho;            /* 1 */
  int adaptive_rho_interval;   /* 0 = OSQP's rule without wall-clock profiling: 4 * check_termination */
  int scaled_termination;      /* 0 */
  int warm_start;              /* 0 = cold start (what the reference effectively does) */
  int linsys;                  /* 0 = sparse LDL' of the quasi-definite KKT (OSQP builtin/QDLDL),
                                  1 = dense Cholesky of P + sigma I + A' diag(rho) A (same iterates
                                      in exact arithmetic; what the HIP kernels factorise) */
} orc_settings;

void orc_default_settings(orc_settings *s);

/* Opaque pattern-level workspace: ordering + symbolic factorisation, shared by every QP with the same
 * sparsity (the reference redoes this per QP, CuCaQP.cpp:183-197; sharing it only makes the timed CPU
 * baseline faster, i.e. harder to beat). P is given in CSC; only entries with row <= col are used
 * (OsqpEigen takes the upper triangle of the both-triangles Hessian CasADi produces). */
typedef struct orc_pattern orc_pattern;

orc_pattern *orc_pattern_create(int n, int m,
                                const int *Pp, const int *Pi,  /* CSC of P (n x n), nnzP = Pp[n] */
                                const int *Ap, const int *Ai); /* CSC of A (m x n), nnzA = Ap[n] */
void orc_pattern_destroy(orc_pattern *pat);
int orc_pattern_kkt_nnzL(const orc_pattern *pat);

/* Solve `batch` QPs sharing the pattern.  Value arrays are instance-major: QP b reads Px + b*strideP
 * etc.; a stride of 0 shares the array across the batch.  x0/y0 (may be NULL) are used only when
 * settings->warm_start != 0.  Outputs (any may be NULL): x [batch*n], y [batch*m], z [batch*m],
 * status/iters [batch], info [batch*4] = {obj, prim_res, dual_res, rho_final}.
 * nthreads <= 1 runs serially; otherwise OpenMP over the batch.  Returns 0 on success. */
int orc_solve_batch(const orc_pattern *pat, const orc_settings *settings, int batch,
                    const double *Px, long strideP, const double *q, long strideq,
                    const double *Ax, long strideA, const double *l, long stridel,
                    const double *u, long strideu,
                    const double *x0, const double *y0,
                    double *x, double *y, double *z, int *status, int *iters, double *info,
                    int nthreads);
/* same, with a per-instance starting rho (rho0 [batch], entries <= 0 or rho0 == NULL mean settings->rho): what a kept
 * OSQP workspace does across osqp_update_* calls -- the fast path the reference's unused CuCaQP::update* members
 * (reference src/sqp_solver/CuCaQP.cpp:106-161) were written for */
int orc_solve_batch_rho(const orc_pattern *pat, const orc_settings *settings, int batch,
                        const double *Px, long strideP, const double *q, long strideq,
                        const double *Ax, long strideA, const double *l, long stridel,
                        const double *u, long strideu,
                        const double *x0, const double *y0, const double *rho0,
                        double *x, double *y, double *z, int *status, int *iters, double *info,
                        int nthreads);

/* Kept workspaces (one per instance): orc_state_solve(vectors_only = 0) is a full setup + solve that keeps every instance's
 * scaled data, D, E, c, rho and factor; vectors_only = 1 replaces q, l, u only, as OSQP's osqp_update_data_vec does on a
 * kept workspace (Px / Ax are ignored) -- the fast path of the reference's unused CuCaQP::update* members
 * (reference src/sqp_solver/CuCaQP.cpp:106-161).  Returns nonzero if vectors_only is asked for before any full solve. */
typedef struct orc_state orc_state;
orc_state *orc_state_create(const orc_pattern *pat, const orc_settings *settings, int batch);
void orc_state_destroy(orc_state *s);
int orc_state_solve(orc_state *s, int vectors_only,
                    const double *Px, long strideP, const double *q, long strideq, const double *Ax, long strideA,
                    const double *l, long stridel, const double *u, long strideu, const double *x0, const double *y0,
                    double *x, double *y, double *z, int *status, int *iters, double *info, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
