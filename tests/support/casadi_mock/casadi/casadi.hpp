// casadi_mock/casadi/casadi.hpp -- TEST INFRASTRUCTURE ONLY.  A minimal stand-in for the handful of CasADi types the CuCaQP seam
// touches (reference include/optimal_control_problem/sqp_solver/CuCaQP.h:105-152 and src/sqp_solver/SQPOptimizationSolver.cpp:155-167):
// a DM that is a CSC matrix of doubles with sparsity().colind() / row(), size1() / size2() / nnz(), ptr(), nonzeros(), and DMVector.
// CasADi is not installed in this image; this header exists so that the CasADi overloads of cpp/CuCaQP.hpp (compiled only when
// <casadi/casadi.hpp> is found) are compiled and driven through the literal reference call sequence.  It pins nothing about CasADi.
#pragma once
#include <vector>

typedef long long casadi_int;

namespace casadi {

class Sparsity {
 public:
  Sparsity() {}
  Sparsity(casadi_int nrow, casadi_int ncol, std::vector<casadi_int> colind, std::vector<casadi_int> row)
      : nrow_(nrow), ncol_(ncol), colind_(std::move(colind)), row_(std::move(row)) {}
  const casadi_int *colind() const { return colind_.data(); }
  const casadi_int *row() const { return row_.data(); }
  casadi_int size1() const { return nrow_; }
  casadi_int size2() const { return ncol_; }
  casadi_int nnz() const { return (casadi_int)row_.size(); }

 private:
  casadi_int nrow_ = 0, ncol_ = 0;
  std::vector<casadi_int> colind_{0}, row_;
};

class DM {
 public:
  DM() {}
  DM(const Sparsity &sp, std::vector<double> nz) : sp_(sp), nz_(std::move(nz)) {}
  explicit DM(const std::vector<double> &dense_column) : nz_(dense_column) {       // a dense column vector
    std::vector<casadi_int> row(dense_column.size());
    for (size_t i = 0; i < row.size(); i++) row[i] = (casadi_int)i;
    sp_ = Sparsity((casadi_int)dense_column.size(), 1, {0, (casadi_int)dense_column.size()}, row);
  }
  const Sparsity &sparsity() const { return sp_; }
  casadi_int size1() const { return sp_.size1(); }
  casadi_int size2() const { return sp_.size2(); }
  casadi_int nnz() const { return sp_.nnz(); }
  const double *ptr() const { return nz_.data(); }
  const std::vector<double> &nonzeros() const { return nz_; }

 private:
  Sparsity sp_;
  std::vector<double> nz_;
};

typedef std::vector<DM> DMVector;

}  // namespace casadi
