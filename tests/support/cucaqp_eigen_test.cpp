// The Eigen overloads of the CuCaQP facade (cpp/CuCaQP.hpp, compiled where <Eigen/Sparse> is found; reference
// include/optimal_control_problem/sqp_solver/CuCaQP.h:37, 49, 51, 53, 55 and getSolution() :76), each called on the GPU against
// tests/support/eigen_mock (Eigen itself is not installed here): the first-iteration QP of test/test.cpp case 1 member by member, in float (the
// reference's OSQPFloat) and in double, then a wrong dimension through every overload (false + message, like reference CuCaQP.cpp:43-103).
// Exit code 0 = pass, 3 = no GPU, 1 = wrong answer, 2 = the Eigen overloads were not compiled.
#include <cmath>
#include <cstdio>

#include "CuCaQP.hpp"

template <class T>
static int run() {
  typedef Eigen::Matrix<T, Eigen::Dynamic, 1> Vec;
  CuCaQP qp;
  if (!qp.setDimension(2, 3)) return 1;
  qp.setVerbosity(false); qp.setWarmStart(true); qp.setAbsoluteTolerance(1e-3); qp.setRelativeTolerance(1e-3); qp.setMaxIteration(10000);
  Eigen::SparseMatrix<T> P(2, 2), A(3, 2);
  P.setCsc({0, 1, 2}, {0, 1}, {T(2), T(2)});
  A.setCsc({0, 2, 4}, {0, 2, 1, 2}, {T(1), T(1), T(1), T(1)});
  const Vec q{T(0), T(0)}, l{T(-50), T(-100), T(1)}, u{T(50), T(100), T(1)};
  if (!qp.setHessianMatrix(P) || !qp.setGradient(q) || !qp.setLinearConstraintsMatrix(A) || !qp.setLowerBound(l) || !qp.setUpperBound(u)) return 1;
  if (!qp.initSolver()) return 3;
  if (!qp.solve()) return 1;
  const Eigen::Matrix<double, Eigen::Dynamic, 1> x = qp.getSolution();
  if (x.size() != 2) return 1;
  std::printf("Eigen overloads, scalar of %zu bytes: dx = %.6f %.6f\n", sizeof(T), x[0], x[1]);
  if (!(std::fabs(x[0] - 0.5) < 5e-3 && std::fabs(x[1] - 0.5) < 5e-3)) return 1;
  Eigen::SparseMatrix<T> P3(3, 3), A2(2, 2);
  P3.setCsc({0, 1, 2, 3}, {0, 1, 2}, {T(2), T(2), T(2)});
  A2.setCsc({0, 1, 2}, {0, 1}, {T(1), T(1)});
  const Vec v3{T(0), T(0), T(0)}, v2{T(0), T(0)};
  CuCaQP w;
  if (!w.setDimension(2, 3)) return 1;
  if (w.setHessianMatrix(P3) || w.setGradient(v3) || w.setLinearConstraintsMatrix(A2) || w.setLowerBound(v2) || w.setUpperBound(v2)) return 1;   // every one refuses a wrong size
  return 0;
}

int main() {
#ifndef MPCQP_HAVE_EIGEN
  return 2;
#else
  int rc = run<float>();
  if (rc) return rc;
  return run<double>();
#endif
}
