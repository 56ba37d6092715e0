// C++ smoke of the CuCaQP facade: the reference's test/test.cpp case 1 QP (x1^2 + x2^2, x1 + x2 = 1,
// test/test.cpp:13-36) through setDimension -> setSystem -> initSolver -> solve -> getSolution.
// Exit code 0 = pass, 3 = no GPU (facade reported the error as the reference would), 1 = wrong answer.
#include <cmath>
#include <cstdio>

#include "CuCaQP.hpp"

int main() {
  CuCaQP qp;
  if (qp.setDimension(0, 1)) return 1;                      // must refuse like reference CuCaQP.cpp:24-27
  if (qp.solve()) return 1;                                 // must refuse like reference CuCaQP.cpp:200-203
  if (!qp.setDimension(2, 3)) return 1;
  qp.setVerbosity(false); qp.setWarmStart(true); qp.setAbsoluteTolerance(1e-3); qp.setRelativeTolerance(1e-3); qp.setMaxIteration(10000);
  const int Pp[] = {0, 1, 2}, Pi[] = {0, 1}; const double Pv[] = {2.0, 2.0};
  const int Ap[] = {0, 2, 4}, Ai[] = {0, 2, 1, 2}; const double Av[] = {1, 1, 1, 1};
  const double q[] = {0, 0}, l[] = {-50, -100, 1}, u[] = {50, 100, 1};
  qp.setSystem({2, 2, Pp, Pi, Pv}, q, {3, 2, Ap, Ai, Av}, l, u);
  if (!qp.initSolver()) return 3;
  if (!qp.solve()) return 1;
  const auto &x = qp.getSolution();
  std::printf("x = %.6f %.6f status %d iters %d\n", x[0], x[1], qp.getStatus()[0], qp.getIterations()[0]);
  return (std::fabs(x[0] - 0.5) < 5e-3 && std::fabs(x[1] - 0.5) < 5e-3 && qp.getStatus()[0] == MPCQP_SOLVED) ? 0 : 1;
}
