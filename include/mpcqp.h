/*
 * mpcqp.h -- C ABI of the MI355X-native batched QP engine (libmpcqp.so).
 *
 * Drop-in boundary.  The reference has no FFI; its seam is the C++ class CuCaQP
 * (reference include/optimal_control_problem/sqp_solver/CuCaQP.h:27-102), driven from one place in the
 * order setDimension -> settings -> [ setSystem(P,q,A,l,u) -> initSolver -> solve -> getSolution ]
 * (reference src/sqp_solver/SQPOptimizationSolver.cpp:80-85,155-167).  Each entry point below names the
 * CuCaQP member(s) it replaces.  One handle owns one batch of QPs that share a sparsity pattern (the
 * reference's batch is 1); distinct handles are independent (one per GPU when sharding a batch).
 *
 * Plain C: no exceptions cross the ABI, integer status codes, caller owns every I/O buffer, the library
 * owns its device workspace.  A handle is not thread-safe (neither is CuCaQP).
 */
#ifndef MPCQP_H
#define MPCQP_H

#ifdef __cplusplus
extern "C" {
#endif

#define MPCQP_OK 0
#define MPCQP_ERR_ARG 1       /* bad dimensions / null pointers / malformed CSC   (CuCaQP.cpp:23-27,49-52) */
#define MPCQP_ERR_HIP 2       /* a HIP runtime call failed; mpcqp_strerror() carries the HIP message        */
#define MPCQP_ERR_NO_GPU 3    /* no gfx950 device / code object missing: the product path has no CPU fallback */
#define MPCQP_ERR_STATE 4     /* call order violated (solve before update, ...)   (CuCaQP.cpp:199-203)      */
#define MPCQP_ERR_LIMIT 5     /* problem does not fit the kernel's on-chip budget                          */

/* per-QP solver status, numerically identical to OSQP 1.0's osqp_status_type */
#define MPCQP_SOLVED 1
#define MPCQP_SOLVED_INACCURATE 2
#define MPCQP_PRIMAL_INFEASIBLE 3
#define MPCQP_PRIMAL_INFEASIBLE_INACCURATE 4
#define MPCQP_DUAL_INFEASIBLE 5
#define MPCQP_DUAL_INFEASIBLE_INACCURATE 6
#define MPCQP_MAX_ITER_REACHED 7
#define MPCQP_NON_CVX 9
#define MPCQP_UNSOLVED 11      /* also: data refused -- an instance with l_i > u_i on some row is not solved (OSQP's setup validation,
                                * which makes CuCaQP::initSolver return false, CuCaQP.cpp:183-197): 0 iterations, NaN in x, y, z */

#define MPCQP_MEM_HOST 0      /* pointer is host memory (pageable or pinned) */
#define MPCQP_MEM_DEVICE 1    /* pointer is device memory on the handle's GPU */

/* Replaces CuCaQP::setVerbosity/setWarmStart/setAbsoluteTolerance/setRelativeTolerance/setMaxIteration
 * (reference src/sqp_solver/CuCaQP.cpp:163-181).  Defaults = OSQP defaults with the values the reference
 * fixes at src/sqp_solver/SQPOptimizationSolver.cpp:81-85 (eps_abs = eps_rel = 1e-3, max_iter = 10000). */
typedef struct mpcqp_settings {
  double rho;                    /* 0.1   */
  double sigma;                  /* 1e-6  */
  double alpha;                  /* 1.6   */
  double eps_abs;                /* 1e-3  */
  double eps_rel;                /* 1e-3  */
  double eps_prim_inf;           /* 1e-4  */
  double eps_dual_inf;           /* 1e-4  */
  double adaptive_rho_tolerance; /* 5     */
  int max_iter;                  /* 10000 */
  int check_termination;         /* 25    */
  int scaling;                   /* 10 Ruiz passes */
  int adaptive_rho;              /* 1     */
  int adaptive_rho_interval;     /* 0 -> 4 * check_termination (deterministic; OSQP's wall-clock rule is not reproducible) */
  int scaled_termination;        /* 0     */
  int warm_start;                /* 0 = cold start from x = z = y = 0, which is what the reference does
                                    because CuCaQP::setSystem clears the solver (CuCaQP.cpp:271-288) */
  int device;                    /* HIP device ordinal, -1 = current device */
} mpcqp_settings;

typedef struct mpcqp_handle mpcqp_handle;

void mpcqp_default_settings(mpcqp_settings *s);

/* Replaces CuCaQP::CuCaQP + setDimension (CuCaQP.cpp:5-41) and the pattern-dependent part of initSolver
 * (CuCaQP.cpp:183-197): n variables, m constraint rows, `batch` QP instances sharing one sparsity.
 * P is CSC n x n -- either triangle or both (CasADi hands the reference both triangles; like OsqpEigen,
 * only entries with row <= col are used); A is CSC m x n.  Index arrays are copied. */
int mpcqp_create(int n, int m, int batch,
                 const int *P_colptr, const int *P_rowidx,
                 const int *A_colptr, const int *A_rowidx,
                 const mpcqp_settings *settings, mpcqp_handle **out);

/* Opt-in: mpcqp_create with the kernel family chosen by MEASUREMENT instead of by the rules tuned on the MPC workloads (for patterns those
 * rules have never seen).  Every family that takes the pattern and size (the rule's own choice, the on-chip modes, the LDS-resident and the
 * global-block kernels) gets a handle, runs a synthetic QP on this pattern (unit diagonal P, pseudo-random A, box around the origin, a fixed
 * number of ADMM iterations), and the fastest one is returned; the others are destroyed.  The handle that comes back IS a handle of that
 * family: results are bitwise what MPCQP_VARIANT=<family> gives.  The choice is cached per (pattern, batch) for the life of the process.
 * Costs a few hundred milliseconds at create -- the pattern-dependent part of what the reference pays in initSolver for every QP
 * (CuCaQP.cpp:183-197).  The environment variable MPCQP_AUTOTUNE=1 makes mpcqp_create behave like this entry point; default: off. */
int mpcqp_create_tuned(int n, int m, int batch,
                       const int *P_colptr, const int *P_rowidx,
                       const int *A_colptr, const int *A_rowidx,
                       const mpcqp_settings *settings, mpcqp_handle **out);

/* Opt-in reduced form.  The reference's QP carries variables that are fixed by construction: the parameter block p, whose rows
 * read p <= p + dp <= p (SQPOptimizationSolver.cpp:47-60,117), and the first frame, pinned through lbx = ubx
 * (src/OptimalControlProblem.cpp:93-96).  OSQP iterates on them like on any other variable, and so does a handle from
 * mpcqp_create.  Here the caller names `nfixed` rows of A that have a single entry and l = u in every instance and every
 * update; their variables are substituted (x_j = l_i / a_ij) before the solve, the remaining QP -- smaller, and without the
 * parameter block's coupling to every stage -- is solved on its own pattern, and x, y, z are returned in the caller's
 * dimensions (the multiplier of an eliminated row from stationarity of its variable).  The handle is used like any other
 * (update / update_vectors / warm_start / set_rho / solve / get; mpcqp_solve_host and mpcqp_debug_scaling are not available).
 * It is a different, shorter ADMM run on an equivalent QP: x agrees with the full form within the termination tolerance, not
 * in the last digits, and status / iteration counts / info are the reduced run's (info[0] without the constant the eliminated
 * variables contribute).  An instance that breaks the promise (l != u on a named row) is refused like one with crossed
 * bounds (MPCQP_UNSOLVED, NaN).  mpcqp_update_vectors reads the P and A arrays of the last mpcqp_update again (the eliminated
 * columns enter q, l, u): device pointers borrowed there must still be valid.  The default stays the full form. */
int mpcqp_create_reduced(int n, int m, int batch,
                         const int *P_colptr, const int *P_rowidx,
                         const int *A_colptr, const int *A_rowidx,
                         int nfixed, const int *fixed_rows,
                         const mpcqp_settings *settings, mpcqp_handle **out);

/* The same without the caller naming rows: the rows to eliminate are FOUND -- every row of A with a single entry whose bounds coincide in every
 * instance of (l, u), the bounds of the first update (host or device arrays, strides as in mpcqp_update; read once, here) -- which is what the
 * reference's formulation produces by itself: 0 <= dp <= 0 on the parameter block (SQPOptimizationSolver.cpp:117) and the first frame pinned
 * through lbx = ubx (src/OptimalControlProblem.cpp:93-96).  A caller that arrives through CuCaQP / OptimalControlProblem and cannot name rows
 * gets the reduced form this way (CuCaQP::setPresolveFixedRows(true) in cpp/CuCaQP.hpp).  Returns the handle of mpcqp_create_reduced on those rows
 * (an ordinary handle when there are none) and their number in *nfixed (may be NULL).  The promise "l = u on these rows in every later update" is
 * checked per instance as there (MPCQP_UNSOLVED, NaN for an instance that breaks it).  The default stays the full form. */
int mpcqp_create_presolved(int n, int m, int batch, const int *Pp, const int *Pi, const int *Ap, const int *Ai,
                           const double *l, long sl, const double *u, long su, int mem,
                           const mpcqp_settings *settings, mpcqp_handle **out, int *nfixed);

/* Replaces CuCaQP::setSystem -> setHessianMatrix/setGradient/setLinearConstraintsMatrix/setLowerBound/
 * setUpperBound (CuCaQP.cpp:43-103,271-288), argument order P,q,A,l,u as at CuCaQP.cpp:283-287.
 * Value arrays are instance-major: QP b reads P + b*strideP (in doubles) ... ; stride 0 shares one array
 * across the batch.  mem = MPCQP_MEM_HOST: values are copied (caller may free at once, as with CuCaQP,
 * which copies into its members CuCaQP.h:83-87).  mem = MPCQP_MEM_DEVICE: pointers are borrowed until
 * the next mpcqp_solve on this handle has completed. */
int mpcqp_update(mpcqp_handle *h,
                 const double *P, long strideP, const double *q, long strideq,
                 const double *A, long strideA, const double *l, long stridel,
                 const double *u, long strideu, int mem);

/* The (unused, private) CuCaQP::update* fast path made real (CuCaQP.cpp:106-161): primal/dual start
 * x0 [batch*n], y0 [batch*m]; honoured by the next solve when settings.warm_start != 0. */
int mpcqp_warm_start(mpcqp_handle *h, const double *x0, const double *y0, int mem);

/* Kept workspace -- the fast path the reference's private, never-called CuCaQP::updateGradient / updateLowerBound /
 * updateUpperBound were written for (CuCaQP.cpp:117-161), with the semantics of OSQP's osqp_update_data_vec: P and A, their
 * scaling (D, E, c), the KKT factorisation and every instance's current (possibly adapted) rho stay from the previous solve on
 * this handle; only q, l, u are replaced.  The next mpcqp_solve skips equilibration and factorisation; an instance is
 * re-factorised only if one of its rows moved between loose / inequality / equality (its rho_i changes then), and an instance
 * whose matrices were found non-convex stays MPCQP_NON_CVX.  x / y start from zero, or from mpcqp_warm_start when
 * settings.warm_start is set.  Call order: mpcqp_keep_workspace(h, 1) -> mpcqp_update -> mpcqp_solve ->
 * { mpcqp_update_vectors -> mpcqp_solve }*; mpcqp_update returns to a full setup.  Strides and mem as in mpcqp_update.
 * MPCQP_ERR_STATE when there is no kept solve yet; MPCQP_ERR_LIMIT on the streaming kernel variant. */
int mpcqp_keep_workspace(mpcqp_handle *h, int enable);
int mpcqp_update_vectors(mpcqp_handle *h, const double *q, long strideq, const double *l, long stridel,
                         const double *u, long strideu, int mem);

/* Per-instance starting rho for the following solves (rho0 [batch]; entries <= 0 mean settings.rho; NULL returns to
 * settings.rho for all).  A kept OSQP workspace carries its adapted rho from one problem to the next
 * (osqp_update_* do not reset it) -- the behaviour the reference's unused update* members would have had
 * (CuCaQP.cpp:106-161); feed info[3] of the previous solve back in. */
int mpcqp_set_rho(mpcqp_handle *h, const double *rho0, int mem);

/* Replaces CuCaQP::initSolver + CuCaQP::solve (CuCaQP.cpp:183-211): per QP, Ruiz scaling, KKT
 * factorisation and the ADMM loop run in one launch on `stream` (a hipStream_t, NULL = default stream).
 * Asynchronous: returns after the launch; results are ordered on `stream`. */
int mpcqp_solve(mpcqp_handle *h, void *stream);

/* Fused host-buffer step for large batches: CuCaQP::setSystem + initSolver + solve + getSolution (CuCaQP.cpp:183-224,271-288) in
 * one call, pipelined -- the batch is cut into `chunks` slices (0 = 6); all host-to-device copies queue on one internal stream
 * in slice order, each slice's kernel and its device-to-host copies run on one of several compute streams as soon as its inputs
 * have landed, so transfers overlap the kernels of earlier slices ("streams matrices via pinned hipMemcpyAsync").  Arrays are dense and instance-major (stride = width; strideP / strideA may be 0 for matrices shared by the
 * batch); x [batch*n], y [batch*m], status, iters [batch] may be NULL.  The copies are asynchronous only from / to pinned host
 * memory (hipHostMalloc, hipHostRegister); pageable memory works but serialises.  Returns when everything has completed.
 * Cold start, full setup, batch-order dispatch; the results also stay on the device for mpcqp_get. */
int mpcqp_solve_host(mpcqp_handle *h, const double *P, long strideP, const double *q, long strideq,
                     const double *A, long strideA, const double *l, long stridel, const double *u, long strideu,
                     double *x, double *y, int *status, int *iters, int chunks);

/* Scheduling hint (on by default; MPCQP_NO_LPT=1 in the environment turns it off at create): after every solve the instances
 * are ranked by their ADMM iteration count, and the next solve on the handle hands them to workgroups in that order, longest
 * first.  Instances are independent, so no output changes by a bit; what changes is the tail of the launch -- with one QP per
 * workgroup and 25- and 100-iteration instances mixed, in-order dispatch leaves CUs idle while the last long instance
 * finishes.  The predictor is exact when the same batch is solved again and good in an MPC loop (consecutive solves of the
 * same plants); a stale hint is harmless.  No reference counterpart (the reference solves one QP at a time). */
int mpcqp_set_dispatch_hint(mpcqp_handle *h, int enable);

/* Replaces CuCaQP::getSolution / getSolutionAsDM (CuCaQP.cpp:213-224), and additionally surfaces what the
 * reference drops: duals y, row activities z, per-QP status, iteration count and
 * info[4] = {objective, primal residual, dual residual, final rho}.  Any output pointer may be NULL.
 * Copies are issued on the stream of the last solve; with MPCQP_MEM_HOST the call returns after they
 * have completed. */
int mpcqp_get(mpcqp_handle *h, double *x, double *y, double *z, int *status, int *iters, double *info, int mem);

int mpcqp_sync(mpcqp_handle *h);                 /* wait for the last solve / copies */
void mpcqp_destroy(mpcqp_handle *h);             /* replaces CuCaQP::~CuCaQP (CuCaQP.cpp:16-21) */
const char *mpcqp_strerror(int code);            /* replaces the std::cerr messages of CuCaQP.cpp */

/* Measurement hooks (no reference counterpart; the reference times setSystem+initSolver+solve with
 * std::chrono, SQPOptimizationSolver.cpp:153-160).  Duration of the last solve kernel from HIP events
 * recorded around the launch on its stream (waits for the kernel). */
int mpcqp_last_kernel_ms(mpcqp_handle *h, float *ms);
/* The same split by the reference's two calls, where the handle runs them as two kernels (the on-chip mode: variant 200 + NW):
 * setup_ms = the set-up kernel (CuCaQP::initSolver, CuCaQP.cpp:183-197), solve_ms = the iteration kernel (CuCaQP::solve,
 * CuCaQP.cpp:199-211).  Handles that do both in one kernel report setup_ms = 0 and the whole kernel in solve_ms. */
int mpcqp_last_phase_ms(mpcqp_handle *h, float *setup_ms, float *solve_ms);
/* info[0..15]: n, m, batch, npad, mpad, n_blocks(n/16), L_blocks, lds_bytes_per_qp, workspace_bytes_per_qp,
 * ordering(0 natural,1 hubs-last,2 twisted), nnzP_triu, nnzA, T_blocks, factor_ops (on-chip kernels: the number of dense 16 x 16 tiles of A the
 * iteration's two sweeps run on, 0 = ELL only), ell_slots_total,
 * variant (0 = streaming kernel, NW > 0 = LDS-resident factor with NW waves per QP, 100 + NW = same kernels with
 * the factor blocks left in the HBM slab, 200 + NW = on-chip mode: factor in LDS + registers) */
int mpcqp_plan_info(const mpcqp_handle *h, long *info16);

/* What bench.py prices a launch of the on-chip mode with (variant 200 + NW; info[0..7] are zero for other kernel families).  info[0..7]: chain blocks of the factor, 1 if
 * the pattern has an arrow head (the parameter block), lengths of the two elimination chains (the critical path of each triangular sweep: one
 * dependent 16 x 16 mat-vec per position), factor blocks resident in LDS, positions per wave, hub blocks per wave in registers, and -- two-kernel
 * form -- the number of {re-factorisation, iteration} launch pairs queued behind a solve for adaptive-rho steps (0 = single kernel);
 * info[8..10]: 64-lane ELL slots of A (by row), A' (by variable) and P that a sweep walks; info[11]: twisted pairs of chains (1: plain or twisted order; more: the dissected order, whose chains info[2], info[3] describe pair 0).  No reference counterpart (the
 * reference's instruments are the two timers of SQPOptimizationSolver.cpp:133-164). */
int mpcqp_oc_info(const mpcqp_handle *h, long *info12);

/* Replaces CuCaQP::printSolverData (CuCaQP.cpp:226-269): copies the scaled problem data the kernel holds
 * for instance b back to the host (D [n], E [m], c; any pointer may be NULL). */
int mpcqp_debug_scaling(mpcqp_handle *h, int b, double *D, double *E, double *c);

/* Kernel self-test of the 16x16 block primitives (MFMA f64 A*B^T, Cholesky+inverse) on caller data:
 * A, B, C are row-major 16x16; out_gemm = C - A*B^T; out_linv = inverse of chol(S) for SPD S (lower). */
int mpcqp_debug_blockops(const double *A, const double *B, const double *C, const double *S,
                         double *out_gemm, double *out_linv, int *potrf_fail);

/* ---------------------------------------------------------------------------------------------------------------
 * Local-system evaluation on device.
 *
 * Replaces SQPOptimizationSolver::getLocalSystem (reference src/sqp_solver/SQPOptimizationSolver.cpp:100-120): the
 * CasADi function localSystemFunction_(p, x, l, u) -> (H, grad f, J_c, l - c, u - c) built at
 * SQPOptimizationSolver.cpp:47-77 with augmented variables w = [p; x] and rows c = [p; x; g(p, x)], evaluated at the
 * current SQP iterate.  Here the evaluation is a HIP kernel over a batch of instances of one stage OCP
 *     min sum_k (s_k - p)' Q (s_k - p) + u_k' R u_k   s.t.  s_{k+1} = F(s_k, u_k),   frame_k = [s_k; u_k], k < horizon
 * (cost as OptimalControlProblem::addVectorCost sums it, reference src/OptimalControlProblem.cpp:574-600; dynamics rows
 * as addEquationConstraint stacks them, :448-470; stage-interleaved frames, src/OCP_config/OCPConfig.cpp:29-46,102),
 * and its outputs are written straight into the device arrays mpcqp_update borrows (MPCQP_MEM_DEVICE), in the CSC value
 * order of mpcqp_stage_pattern -- so an SQP iteration never leaves the GPU.  Jacobians of F come from forward-mode dual
 * numbers inside the kernel (one thread per instance and QP column), not from finite differences.
 * np = nx (p is the reference state), n = np + horizon * (nx + nu), m = n + (horizon - 1) * nx + horizon * nh, where nh is the
 * number of rows of an optional per-frame path constraint lo <= h(s_k, u_k) <= hi (generated libraries only; rows [p; x; g; h],
 * their bounds travel in lbg / ubg behind the dynamics rows). */
#define MPCQP_MODEL_DOUBLE_INTEGRATOR 0   /* nx 2, nu 1, exact discrete map; no parameters                       */
#define MPCQP_MODEL_QUADROTOR 1           /* nx 12, nu 4, RK4; par = {mass, grav, arm, kappa, Jx, Jy, Jz}        */
#define MPCQP_MODEL_CARTPOLE 2            /* nx 4, nu 1, RK4; par = {m_cart, m_pole, length, grav}               */
#define MPCQP_MODEL_USER 3                /* dynamics from a generated library, see mpcqp_stage_create_user       */

typedef struct mpcqp_stage_desc {
  int model;        /* MPCQP_MODEL_*            */
  int horizon;      /* number of frames N >= 2  */
  double dt;        /* step of the discrete map */
  double Q[16];     /* diagonal state weights (first nx used)  */
  double R[8];      /* diagonal input weights (first nu used)  */
  double par[8];    /* model parameters, see MPCQP_MODEL_*     */
  int device;       /* HIP device ordinal, -1 = current device */
} mpcqp_stage_desc;

typedef struct mpcqp_stage mpcqp_stage;

/* fills `d` with the zoo's defaults for `model` (weights, parameters and dt of SURVEY.md section 8d) */
int mpcqp_stage_default(int model, int horizon, mpcqp_stage_desc *d);
/* the once-per-problem part (the reference's constructor builds the symbolic function once, SQPOptimizationSolver.cpp:12-92) */
int mpcqp_stage_create(const mpcqp_stage_desc *d, mpcqp_stage **out);
/* User-defined dynamics: the native form of the reference's gen_code / load_lib flow (solver_settings.gen_code writes the
 * CasADi function out as C, gcc builds a shared library, load_lib loads it; reference src/OptimalControlProblem.cpp:263-287,
 * 602-640).  `library_path` is a shared library generated by optimal_control_problem_amd/codegen.py from a traced discrete
 * map s_{k+1} = F(s_k, u_k): a scalar-generic functor instantiated into the same evaluation kernels (hipcc, gfx950), exporting
 * mpcqp_user_abi / _dims / _eval / _merit.  d->model and d->par are ignored (constants are baked into the generated code);
 * nx <= 16, nu <= 8.  A library generated with a stage cost l(s, u, r) (and optionally a terminal one) -- the general form of the
 * SX cost terms the reference sums in addScalarCost, src/OptimalControlProblem.cpp:491-497 -- also exports mpcqp_user_cost: the
 * objective is then sum_k l(s_k, u_k, p) with its exact Hessian (the reference's hessian(f, w), SQPOptimizationSolver.cpp:55-60) in
 * the structure the generated code reports (mpcqp_stage_pattern), and d->Q, d->R, mpcqp_stage_set_weights do not apply. */
int mpcqp_stage_create_user(const mpcqp_stage_desc *d, const char *library_path, mpcqp_stage **out);
void mpcqp_stage_destroy(mpcqp_stage *s);
/* Per-frame diagonal weights (terminal costs, ramps): Qk [horizon * nx], Rk [horizon * nu], host pointers, copied; frame k is
 * weighted by Qk[k*nx ...], Rk[k*nu ...] instead of desc.Q, desc.R (the reference calls addVectorCost once per step, so weights may
 * differ by step, reference readme.md:121-128).  NULL, NULL returns to desc.Q, desc.R.  MPCQP_ERR_ARG for an evaluator generated
 * with its own stage cost. */
int mpcqp_stage_set_weights(mpcqp_stage *s, const double *Qk, const double *Rk);
/* Per-frame bounds of the path constraint, lo / hi [horizon * nh] host pointers, copied: what mpcqp_stage_merit measures violations
 * against when the bounds differ by frame -- a terminal constraint is a path constraint that is loose (-inf, +inf) on every frame
 * but the last (addInequalityConstraint is called per frame in the reference, src/OptimalControlProblem.cpp:448-470, so bounds may
 * differ by frame).  The QP itself takes its bounds from lbg / ubg of mpcqp_stage_eval either way.  NULL, NULL returns to the
 * bounds baked into the generated library. */
int mpcqp_stage_set_path_bounds(mpcqp_stage *s, const double *lo, const double *hi);
/* dims[8] = {nx, nu, np, n, m, nnz(P), nnz(A), horizon * (nx + nu)} */
int mpcqp_stage_dims(const mpcqp_stage *s, int *dims8);
/* 1 when the evaluator was generated with its own stage cost (mpcqp_stage_create_user above), 0 for diagonal tracking weights */
int mpcqp_stage_has_cost(const mpcqp_stage *s);
/* CSC sparsity of P (n x n, both triangles, as CasADi hands it to CuCaQP) and A = [I; dg/dw] (m x n): the arrays
 * mpcqp_create takes.  Pp, Ap: n + 1 entries; Pi: nnz(P); Ai: nnz(A).  Host pointers. */
int mpcqp_stage_pattern(const mpcqp_stage *s, int *Pp, int *Pi, int *Ap, int *Ai);
/* getLocalSystem for `batch` instances.  All pointers are device memory, instance-major and dense: p [batch*np],
 * x, lbx, ubx [batch*horizon*(nx+nu)], lbg, ubg [batch*(horizon-1)*nx]  ->  P [batch*nnzP], q [batch*n],
 * A [batch*nnzA], l, u [batch*m] with l = [p; lbx; lbg] - c, u = [p; ubx; ubg] - c.  Asynchronous on `stream`. */
int mpcqp_stage_eval(mpcqp_stage *s, int batch, const double *p, const double *x,
                     const double *lbx, const double *ubx, const double *lbg, const double *ubg,
                     double *P, double *q, double *A, double *l, double *u, void *stream);
/* objective f [batch] (SQPOptimizationSolver.cpp:180-181) and max-norm of the dynamics violation gmax [batch]
 * (either may be NULL) at iterate x; device pointers */
int mpcqp_stage_merit(mpcqp_stage *s, int batch, const double *p, const double *x, double *f, double *gmax, void *stream);
/* the damped update result.x += alpha * solution[pSize:] (SQPOptimizationSolver.cpp:171-177): x [batch*nvar] +=
 * alpha * dw[b*n + np ...]; device pointers.  step_max [batch] (may be NULL) receives max|alpha * dx| per instance.
 * status [batch] (may be NULL = the reference's behaviour: every step is taken, NaN from an infeasible QP included):
 * when given, instances whose QP status is not MPCQP_SOLVED / _SOLVED_INACCURATE / _MAX_ITER_REACHED keep their x. */
int mpcqp_stage_step(mpcqp_stage *s, int batch, double alpha, const double *dw, double *x, double *step_max,
                     const int *status, void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * Structured stage form: the same QP, given by its stage blocks instead of CSC value arrays.
 *
 * The reference only ever produces OCP-structured QPs: the decision vector is the stage-interleaved list of frames
 * [s_k; u_k] (src/OCP_config/OCPConfig.cpp:29-46,102) behind a parameter block p, and the local QP has w = [p; frames] and rows
 * [p; frames; g] (src/sqp_solver/SQPOptimizationSolver.cpp:47-77).  A caller that has the blocks -- an LTV / linearised MPC that
 * never went through CasADi -- would have to scatter them into CSC arrays itself to call mpcqp_update.  Here it names the stage
 * dimensions once and hands over, per instance (instance-major, dense, row-major):
 *     H   [N][f][f]       Hessian block of frame k, f = nx + nu (symmetric; both triangles are read as given)
 *     Hp  [N][np][f]      P[p, frame_k]  (its transpose fills P[frame_k, p]);   Hpp [np][np]  P[p, p]      (both NULL when np = 0)
 *     AB  [N-1][nx][f]    [A_k B_k] of  s_{k+1} = A_k s_k + B_k u_k + c_k : dynamics row block k reads
 *                         lg_k <= s_{k+1} - A_k s_k - B_k u_k <= ug_k  (an equality when lg_k = ug_k = c_k), the sign convention of the
 *                         reference's rows (+1 on s_{k+1}, -dF/d[s_k; u_k]; compare mpcqp_stage_eval above)
 *     q [n], l [m], u [m] as in mpcqp_update: n = np + N f in the order [p; frames], m = n + (N - 1) nx in the order
 *                         [p; frames; dynamics] (the identity rows carry the box bounds; the reference pins p with l = u)
 * The QP is the one mpcqp_create / mpcqp_update would get for the CSC pattern mpcqp_stageqp_pattern returns -- for np = nx and dense
 * masks exactly mpcqp_stage_pattern's -- and it runs on an ordinary handle on that pattern: blocks and CSC arrays holding the same
 * numbers give bitwise the same x, y, z, status and iteration counts (tests/test_gpu_stageqp.py).  One gather kernel per update writes
 * the CSC value arrays (algorithmic bytes: blocks in, nnz(P) + nnz(A) values out); the ADMM kernels are not aware of this form.
 * Optional masks keep structural zeros out of the pattern (a diagonal tracking cost, a sparse Jacobian): entries masked out are not
 * read.  np = 0: a plain LQ-structured QP without the reference's parameter block. */
typedef struct mpcqp_stageqp_dims {
  int N;                            /* frames, >= 2 */
  int nx, nu;                       /* frame k = [s_k (nx); u_k (nu)] */
  int np;                           /* leading parameter block (0 = none) */
  const unsigned char *cost_mask;   /* optional [(f + np)^2] row-major over the local variables [s; u; p]: which entries of H_k, Hp_k, Hpp exist
                                       (symmetric, diagonal set; the same for every frame); NULL = dense blocks */
  const unsigned char *dyn_mask;    /* optional [nx * f] row-major: which entries of [A_k B_k] exist; NULL = dense */
} mpcqp_stageqp_dims;

typedef struct mpcqp_stageqp mpcqp_stageqp;

/* CSC pattern of the stage form (host only; no GPU needed): sizes4 = {n, m, nnz(P), nnz(A)}; any of the arrays may be NULL (query the
 * sizes first).  P has both triangles, rows ascending inside a column -- what CasADi hands CuCaQP::setHessianMatrix. */
int mpcqp_stageqp_pattern(const mpcqp_stageqp_dims *d, int *sizes4, int *P_colptr, int *P_rowidx, int *A_colptr, int *A_rowidx);
/* mpcqp_create on that pattern plus the maps and value arrays of the gather kernel */
int mpcqp_stageqp_create(const mpcqp_stageqp_dims *d, int batch, const mpcqp_settings *settings, mpcqp_stageqp **out);
/* the ordinary handle underneath: mpcqp_solve / _get / _warm_start / _set_rho / _keep_workspace / _update_vectors / _plan_info apply to it
 * (x, y, z come back in the orders given above).  Owned by the mpcqp_stageqp: do not destroy it. */
mpcqp_handle *mpcqp_stageqp_handle(mpcqp_stageqp *s);
/* mpcqp_update in blocks.  mem = MPCQP_MEM_DEVICE: the gather kernel is queued on `stream` -- the stream of the mpcqp_solve calls on this
 * handle -- and q, l, u are borrowed until that solve has completed, like in mpcqp_update; the block arrays are free once the kernel
 * has run.  mem = MPCQP_MEM_HOST: everything is copied before the call returns. */
int mpcqp_stageqp_update(mpcqp_stageqp *s, const double *H, const double *Hp, const double *Hpp, const double *AB,
                         const double *q, const double *l, const double *u, int mem, void *stream);
void mpcqp_stageqp_destroy(mpcqp_stageqp *s);

#ifdef __cplusplus
}
#endif
#endif
