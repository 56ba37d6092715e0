// CuCaQP.hpp -- C++ facade over the C ABI (include/mpcqp.h) with the reference class's public surface.
//
// Mirrors reference include/optimal_control_problem/sqp_solver/CuCaQP.h:27-102 so that
// SQPOptimizationSolver's call sequence (reference src/sqp_solver/SQPOptimizationSolver.cpp:80-85,155-167)
//     setDimension -> setVerbosity/WarmStart/AbsoluteTolerance/RelativeTolerance/MaxIteration
//     -> [ setSystem(P,q,A,l,u) -> initSolver -> solve -> getSolution ]  per SQP iteration
// compiles unchanged against it.  Same bool-return + std::cerr error behaviour (reference
// src/sqp_solver/CuCaQP.cpp); values stay fp64 (the reference's OSQP build is float, cpu_install.sh:44).
// The always-available overloads take raw CSC (colptr, rowidx, values) and dense pointers; the CasADi and the Eigen
// overloads of the reference are compiled only where those headers exist (they do not in this image: the tests compile them against
// tests/support/casadi_mock and tests/support/eigen_mock).
// One object may hold a batch of QPs of one sparsity (value arrays instance-major); batch = 1 is the drop-in case.
#pragma once
#include <algorithm>
#include <cstddef>
#include <iostream>
#include <vector>

#include "mpcqp.h"

#if defined(__has_include)
#if __has_include(<casadi/casadi.hpp>)
#include <casadi/casadi.hpp>
#define MPCQP_HAVE_CASADI 1
#endif
#if __has_include(<Eigen/Sparse>)
#include <Eigen/Dense>
#include <Eigen/Sparse>
#define MPCQP_HAVE_EIGEN 1
#endif
#endif

class CuCaQP {
 public:
  struct CscView { int rows, cols; const int *colptr, *rowidx; const double *values; };

  explicit CuCaQP(int batch = 1) : batch_(batch) { mpcqp_default_settings(&settings_); }
  ~CuCaQP() { clearSolver(); }                                   // reference CuCaQP.cpp:16-21
  CuCaQP(const CuCaQP &) = delete;
  CuCaQP &operator=(const CuCaQP &) = delete;

  bool setDimension(int numOfVariables, int numOfConstraints) {  // reference CuCaQP.cpp:23-41
    if (numOfVariables <= 0 || numOfConstraints <= 0) { std::cerr << "Error: Invalid dimensions." << std::endl; return false; }
    clearSolver();
    numOfVariables_ = numOfVariables; numOfConstraints_ = numOfConstraints;
    return true;
  }

  // settings pass-through, reference CuCaQP.cpp:163-181
  void setVerbosity(bool verbosity) { verbose_ = verbosity; }
  void setWarmStart(bool) {}  // moot in the reference too: setSystem clears the solver (CuCaQP.cpp:271-280)
  void setAbsoluteTolerance(double tolerance) { settings_.eps_abs = tolerance; dirty_ = true; }
  void setRelativeTolerance(double tolerance) { settings_.eps_rel = tolerance; dirty_ = true; }
  void setMaxIteration(int maxIteration) { settings_.max_iter = maxIteration; dirty_ = true; }
  mpcqp_settings &settings() { dirty_ = true; return settings_; }
  // opt-in: variables that a singleton row pins (l = u in every instance: the reference's 0 <= dp <= 0 rows, SQPOptimizationSolver.cpp:117, and the
  // first frame, OptimalControlProblem.cpp:93-96) are found from the bounds at initSolver() and substituted before the solve
  // (mpcqp_create_presolved).  An equivalent QP, a shorter ADMM run; the default (false) solves the QP as OSQP does.
  void setPresolveFixedRows(bool on) { if (on != presolve_) { presolve_ = on; dirty_ = true; } }
  int presolvedRows() const { return nfixed_; }

  // data, reference CuCaQP.cpp:43-103 (values are copied, like the reference copies into its members CuCaQP.h:83-87)
  bool setHessianMatrix(const CscView &P) {
    if (P.rows != numOfVariables_ || P.cols != numOfVariables_) {
      std::cerr << "Error: Hessian matrix dimensions mismatch. Expected " << numOfVariables_ << "x" << numOfVariables_ << std::endl;
      return false;
    }
    patternChanged_ |= assignPattern(P, Pp_, Pi_);
    Pv_.assign(P.values, P.values + (size_t)batch_ * Pi_.size());
    return true;
  }
  bool setGradient(const double *q, int size) {
    if (size != numOfVariables_) { std::cerr << "Error: Gradient vector size mismatch. Expected " << numOfVariables_ << std::endl; return false; }
    q_.assign(q, q + (size_t)batch_ * size);
    return true;
  }
  bool setLinearConstraintsMatrix(const CscView &A) {
    if (A.rows != numOfConstraints_ || A.cols != numOfVariables_) {
      std::cerr << "Error: Constraint matrix dimensions mismatch. Expected " << numOfConstraints_ << "x" << numOfVariables_ << std::endl;
      return false;
    }
    patternChanged_ |= assignPattern(A, Ap_, Ai_);
    Av_.assign(A.values, A.values + (size_t)batch_ * Ai_.size());
    return true;
  }
  bool setLowerBound(const double *l, int size) {
    if (size != numOfConstraints_) { std::cerr << "Error: Lower bound vector size mismatch. Expected " << numOfConstraints_ << std::endl; return false; }
    l_.assign(l, l + (size_t)batch_ * size);
    return true;
  }
  bool setUpperBound(const double *u, int size) {
    if (size != numOfConstraints_) { std::cerr << "Error: Upper bound vector size mismatch. Expected " << numOfConstraints_ << std::endl; return false; }
    u_.assign(u, u + (size_t)batch_ * size);
    return true;
  }
  // reference CuCaQP.cpp:271-288 (order P, q, A, l, u); return values dropped like the reference
  void setSystem(const CscView &P, const double *q, const CscView &A, const double *l, const double *u) {
    isInitialized_ = false;
    setHessianMatrix(P); setGradient(q, numOfVariables_); setLinearConstraintsMatrix(A);
    setLowerBound(l, numOfConstraints_); setUpperBound(u, numOfConstraints_);
  }

  // The reference's private update* members (CuCaQP.h:93-101, CuCaQP.cpp:106-161; never called there).  Vectors go through the
  // kept workspace (mpcqp_update_vectors = OSQP's osqp_update_data_vec: scaling, factorisation and rho stay, the next solve()
  // skips the setup); a matrix update falls back to a full setup at the next solve().
  bool updateHessianMatrix(const CscView &P) {
    if (!isInitialized_) { std::cerr << "Error: Solver not initialized. Call initSolver() first." << std::endl; return false; }
    const bool same = samePattern(P, Pp_, Pi_);
    return setHessianMatrix(P) && updateMatrix(same, "Hessian");
  }
  bool updateLinearConstraintsMatrix(const CscView &A) {
    if (!isInitialized_) { std::cerr << "Error: Solver not initialized. Call initSolver() first." << std::endl; return false; }
    const bool same = samePattern(A, Ap_, Ai_);
    return setLinearConstraintsMatrix(A) && updateMatrix(same, "Constraint");
  }
  bool updateGradient(const double *q, int size) { return updateVector(setGradient(q, size)); }
  bool updateLowerBound(const double *l, int size) { return updateVector(setLowerBound(l, size)); }
  bool updateUpperBound(const double *u, int size) { return updateVector(setUpperBound(u, size)); }

  bool initSolver() {                                            // reference CuCaQP.cpp:183-197
    isInitialized_ = false;
    if (Pp_.empty() || Ap_.empty() || q_.empty() || l_.empty() || u_.empty()) { std::cerr << "Error: Failed to initialize solver." << std::endl; return false; }
    // osqp_setup refuses l_i > u_i, so OsqpEigen's initSolver fails (reference CuCaQP.cpp:183-197); with a batch, instances are refused
    // one by one (status MPCQP_UNSOLVED) and only a batch without any valid instance fails here
    bool anyValid = numOfConstraints_ == 0;
    for (int b = 0; b < batch_ && !anyValid; b++) {
      bool crossed = false;
      for (int i = 0; i < numOfConstraints_; i++) crossed |= l_[(size_t)b * numOfConstraints_ + i] > u_[(size_t)b * numOfConstraints_ + i];
      anyValid = !crossed;
    }
    if (!anyValid) { std::cerr << "Error: Failed to initialize solver. (lower bound greater than upper bound)" << std::endl; return false; }
    int rc = MPCQP_OK;
    if (!handle_ || patternChanged_ || dirty_) {
      clearSolver();
      nfixed_ = 0;
      rc = presolve_ ? mpcqp_create_presolved(numOfVariables_, numOfConstraints_, batch_, Pp_.data(), Pi_.data(), Ap_.data(), Ai_.data(), l_.data(), numOfConstraints_,
                                              u_.data(), numOfConstraints_, MPCQP_MEM_HOST, &settings_, &handle_, &nfixed_)
                     : mpcqp_create(numOfVariables_, numOfConstraints_, batch_, Pp_.data(), Pi_.data(), Ap_.data(), Ai_.data(), &settings_, &handle_);
      patternChanged_ = dirty_ = false;
      kept_ = rc == MPCQP_OK && mpcqp_keep_workspace(handle_, 1) == MPCQP_OK;   // not on the streaming kernel variant
    }
    vectorsOnly_ = matricesDirty_ = false; solvedOnce_ = false;
    if (rc == MPCQP_OK)
      rc = mpcqp_update(handle_, Pv_.data(), (long)Pi_.size(), q_.data(), numOfVariables_, Av_.data(), (long)Ai_.size(),
                        l_.data(), numOfConstraints_, u_.data(), numOfConstraints_, MPCQP_MEM_HOST);
    if (rc != MPCQP_OK) { std::cerr << "Error: Failed to initialize solver. (" << mpcqp_strerror(rc) << ")" << std::endl; return false; }
    isInitialized_ = true;
    return true;
  }

  bool solve() {                                                 // reference CuCaQP.cpp:199-211
    if (!isInitialized_) { std::cerr << "Error: Solver not initialized. Call initSolver() first." << std::endl; return false; }
    int rc = MPCQP_OK;
    if (vectorsOnly_ && kept_ && solvedOnce_ && !matricesDirty_)
      rc = mpcqp_update_vectors(handle_, q_.data(), numOfVariables_, l_.data(), numOfConstraints_, u_.data(), numOfConstraints_, MPCQP_MEM_HOST);
    else if (vectorsOnly_ || matricesDirty_)
      rc = mpcqp_update(handle_, Pv_.data(), (long)Pi_.size(), q_.data(), numOfVariables_, Av_.data(), (long)Ai_.size(),
                        l_.data(), numOfConstraints_, u_.data(), numOfConstraints_, MPCQP_MEM_HOST);
    vectorsOnly_ = matricesDirty_ = false;
    if (rc == MPCQP_OK) rc = mpcqp_solve(handle_, nullptr);
    solvedOnce_ = rc == MPCQP_OK;
    solution_.resize((size_t)batch_ * numOfVariables_); status_.resize(batch_); iters_.resize(batch_);
    if (rc == MPCQP_OK) rc = mpcqp_get(handle_, solution_.data(), nullptr, nullptr, status_.data(), iters_.data(), nullptr, MPCQP_MEM_HOST);
    if (rc != MPCQP_OK) { std::cerr << "Error: Failed to solve problem. Error code: " << rc << std::endl; return false; }
    if (verbose_) std::cout << "mpcqp: status " << status_[0] << " after " << iters_[0] << " ADMM iterations" << std::endl;
    return true;
  }

#ifdef MPCQP_HAVE_EIGEN
  // reference CuCaQP.h:76 / CuCaQP.cpp:213-215: the primal solution as an Eigen column (instance 0; fp64 where the reference's OSQP build is float)
  Eigen::Matrix<double, Eigen::Dynamic, 1> getSolution() const {
    Eigen::Matrix<double, Eigen::Dynamic, 1> x((long)numOfVariables_);
    for (int j = 0; j < numOfVariables_ && (size_t)j < solution_.size(); j++) x[j] = solution_[j];
    return x;
  }
#else
  const std::vector<double> &getSolution() const { return solution_; }   // reference CuCaQP.cpp:213-215 (an Eigen column where <Eigen/Sparse> exists)
#endif
  const std::vector<double> &getSolutionVector() const { return solution_; }      // every instance of the batch, instance-major
  const std::vector<int> &getStatus() const { return status_; }
  const std::vector<int> &getIterations() const { return iters_; }

  void printSolverData() const {                                 // reference CuCaQP.cpp:226-269
    std::cout << "q:"; for (int i = 0; i < numOfVariables_; i++) std::cout << " " << q_[i]; std::cout << std::endl;
    std::cout << "l:"; for (int i = 0; i < numOfConstraints_; i++) std::cout << " " << l_[i]; std::cout << std::endl;
    std::cout << "u:"; for (int i = 0; i < numOfConstraints_; i++) std::cout << " " << u_[i]; std::cout << std::endl;
    for (int j = 0; j < numOfVariables_; j++) for (int p = Pp_[j]; p < Pp_[j + 1]; p++) std::cout << "P(" << Pi_[p] << "," << j << "): " << Pv_[p] << std::endl;
    for (int j = 0; j < numOfVariables_; j++) for (int p = Ap_[j]; p < Ap_[j + 1]; p++) std::cout << "A(" << Ai_[p] << "," << j << "): " << Av_[p] << std::endl;
  }

#ifdef MPCQP_HAVE_EIGEN
  // Eigen overloads of the reference (CuCaQP.h:37, 49, 51, 53, 55), for any scalar type (the reference's OSQPFloat is float, cpu_install.sh:44; values are
  // widened to fp64): same checks, same messages, same bool returns as the raw-pointer forms they forward to.  A sparse matrix must be compressed
  // column-major storage (what the reference's converter produces with makeCompressed, CuCaQP.h:105-137).
  template <class T>
  bool setHessianMatrix(const Eigen::SparseMatrix<T> &P) { EigenCsc<T> c(P); return c.ok && setHessianMatrix(c.view()); }
  template <class T>
  bool setLinearConstraintsMatrix(const Eigen::SparseMatrix<T> &A) { EigenCsc<T> c(A); return c.ok && setLinearConstraintsMatrix(c.view()); }
  template <class T>
  bool setGradient(const Eigen::Matrix<T, Eigen::Dynamic, 1> &q) { const std::vector<double> v(q.data(), q.data() + q.size()); return setGradient(v.data(), (int)v.size()); }
  template <class T>
  bool setLowerBound(const Eigen::Matrix<T, Eigen::Dynamic, 1> &l) { const std::vector<double> v(l.data(), l.data() + l.size()); return setLowerBound(v.data(), (int)v.size()); }
  template <class T>
  bool setUpperBound(const Eigen::Matrix<T, Eigen::Dynamic, 1> &u) { const std::vector<double> v(u.data(), u.data() + u.size()); return setUpperBound(v.data(), (int)v.size()); }
  template <class T>
  struct EigenCsc {      // index arrays as int, values as double, for the life of one call (the setters copy)
    bool ok; int rows, cols; std::vector<int> cp, ri; std::vector<double> val;
    explicit EigenCsc(const Eigen::SparseMatrix<T> &M) : ok(M.isCompressed()), rows((int)M.rows()), cols((int)M.cols()) {
      if (!ok) { std::cerr << "Error: sparse matrix is not in compressed storage. Call makeCompressed() first." << std::endl; return; }
      cp.assign(M.outerIndexPtr(), M.outerIndexPtr() + M.cols() + 1); ri.assign(M.innerIndexPtr(), M.innerIndexPtr() + M.nonZeros());
      val.assign(M.valuePtr(), M.valuePtr() + M.nonZeros());
    }
    CscView view() const { return CscView{rows, cols, cp.data(), ri.data(), val.data()}; }
  };
#endif

#ifdef MPCQP_HAVE_CASADI
  // CasADi overloads of the reference (CuCaQP.h:38-48, converters CuCaQP.h:105-152): DM is CSC already.
  static std::vector<int> toInt(const casadi_int *p, casadi_int n) { return std::vector<int>(p, p + n); }
  void setSystem(casadi::DMVector sys) {
    const casadi::DM &P = sys[0], &A = sys[2];
    pc_ = toInt(P.sparsity().colind(), P.size2() + 1); pr_ = toInt(P.sparsity().row(), P.nnz());
    ac_ = toInt(A.sparsity().colind(), A.size2() + 1); ar_ = toInt(A.sparsity().row(), A.nnz());
    setSystem(CscView{(int)P.size1(), (int)P.size2(), pc_.data(), pr_.data(), P.ptr()}, sys[1].ptr(),
              CscView{(int)A.size1(), (int)A.size2(), ac_.data(), ar_.data(), A.ptr()}, sys[3].ptr(), sys[4].ptr());
  }
  casadi::DM getSolutionAsDM() const { return casadi::DM(std::vector<double>(solution_.begin(), solution_.begin() + numOfVariables_)); }
  // the per-member overloads (reference CuCaQP.h:38-48, CuCaQP.cpp:43-103): same checks, same messages, same bool returns.  A DM vector is
  // read through its structural non-zeros like the reference's casadiDMToEigenVector (CuCaQP.h:139-152): entries outside the sparsity are 0.
  bool setHessianMatrix(const casadi::DM &hessian) {
    pc_ = toInt(hessian.sparsity().colind(), hessian.size2() + 1); pr_ = toInt(hessian.sparsity().row(), hessian.nnz());
    return setHessianMatrix(CscView{(int)hessian.size1(), (int)hessian.size2(), pc_.data(), pr_.data(), hessian.ptr()});
  }
  bool setLinearConstraintsMatrix(const casadi::DM &A) {
    ac_ = toInt(A.sparsity().colind(), A.size2() + 1); ar_ = toInt(A.sparsity().row(), A.nnz());
    return setLinearConstraintsMatrix(CscView{(int)A.size1(), (int)A.size2(), ac_.data(), ar_.data(), A.ptr()});
  }
  bool setGradient(const casadi::DM &q) { const std::vector<double> v = denseColumn(q); return setGradient(v.data(), (int)v.size()); }
  bool setLowerBound(const casadi::DM &l) { const std::vector<double> v = denseColumn(l); return setLowerBound(v.data(), (int)v.size()); }
  bool setUpperBound(const casadi::DM &u) { const std::vector<double> v = denseColumn(u); return setUpperBound(v.data(), (int)v.size()); }
  // ... and the DM forms of the update members the reference declares private (CuCaQP.h:93-101)
  bool updateHessianMatrix(const casadi::DM &hessian) {
    pc_ = toInt(hessian.sparsity().colind(), hessian.size2() + 1); pr_ = toInt(hessian.sparsity().row(), hessian.nnz());
    return updateHessianMatrix(CscView{(int)hessian.size1(), (int)hessian.size2(), pc_.data(), pr_.data(), hessian.ptr()});
  }
  bool updateLinearConstraintsMatrix(const casadi::DM &A) {
    ac_ = toInt(A.sparsity().colind(), A.size2() + 1); ar_ = toInt(A.sparsity().row(), A.nnz());
    return updateLinearConstraintsMatrix(CscView{(int)A.size1(), (int)A.size2(), ac_.data(), ar_.data(), A.ptr()});
  }
  bool updateGradient(const casadi::DM &q) { const std::vector<double> v = denseColumn(q); return updateGradient(v.data(), (int)v.size()); }
  bool updateLowerBound(const casadi::DM &l) { const std::vector<double> v = denseColumn(l); return updateLowerBound(v.data(), (int)v.size()); }
  bool updateUpperBound(const casadi::DM &u) { const std::vector<double> v = denseColumn(u); return updateUpperBound(v.data(), (int)v.size()); }
  static std::vector<double> denseColumn(const casadi::DM &v) {      // (one instance: the CasADi seam is the batch = 1 drop-in)
    std::vector<double> d((size_t)(v.size1() * v.size2()), 0.0);
    const casadi_int *cp = v.sparsity().colind(), *ri = v.sparsity().row();
    for (casadi_int j = 0; j < v.size2(); j++) for (casadi_int k = cp[j]; k < cp[j + 1]; k++) d[(size_t)(j * v.size1() + ri[k])] = v.ptr()[k];
    return d;
  }
#endif

 private:
  bool assignPattern(const CscView &M, std::vector<int> &cp, std::vector<int> &ri) {
    std::vector<int> ncp(M.colptr, M.colptr + M.cols + 1), nri(M.rowidx, M.rowidx + M.colptr[M.cols]);
    bool changed = ncp != cp || nri != ri;
    cp.swap(ncp); ri.swap(nri);
    return changed;
  }
  // a matrix update keeps the plan: another sparsity pattern is stored but refused here and needs initSolver() (osqp_update_data_mat
  // has the same rule)
  static bool samePattern(const CscView &M, const std::vector<int> &cp, const std::vector<int> &ri) {
    return (int)cp.size() == M.cols + 1 && std::equal(cp.begin(), cp.end(), M.colptr) && (int)ri.size() == M.colptr[M.cols] && std::equal(ri.begin(), ri.end(), M.rowidx);
  }
  bool updateMatrix(bool same, const char *what) {
    if (!same) { isInitialized_ = false; std::cerr << "Error: " << what << " sparsity pattern changed. Call initSolver() again." << std::endl; return false; }
    vectorsOnly_ = false; matricesDirty_ = true;
    return true;
  }
  void clearSolver() { if (handle_) { mpcqp_destroy(handle_); handle_ = nullptr; } isInitialized_ = false; }
  bool updateVector(bool stored) {
    if (!isInitialized_) { std::cerr << "Error: Solver not initialized. Call initSolver() first." << std::endl; return false; }   // CuCaQP.cpp:118-121
    if (stored) vectorsOnly_ = true;
    return stored;
  }

  int batch_ = 1, numOfVariables_ = 0, numOfConstraints_ = 0;
  bool isInitialized_ = false, verbose_ = false, patternChanged_ = true, dirty_ = false;
  bool kept_ = false, vectorsOnly_ = false, matricesDirty_ = false, solvedOnce_ = false, presolve_ = false;
  int nfixed_ = 0;
  mpcqp_settings settings_;
  mpcqp_handle *handle_ = nullptr;
  std::vector<int> Pp_, Pi_, Ap_, Ai_, status_, iters_;
  std::vector<double> Pv_, Av_, q_, l_, u_, solution_;
#ifdef MPCQP_HAVE_CASADI
  std::vector<int> pc_, pr_, ac_, ar_;
#endif
};
