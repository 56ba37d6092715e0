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
  const double lx[] = {-50, -100, 2}, ux[] = {50, 100, 1};   // crossed bounds: refused like osqp_setup refuses them (CuCaQP.cpp:183-197)
  qp.setSystem({2, 2, Pp, Pi, Pv}, q, {3, 2, Ap, Ai, Av}, lx, ux);
  if (qp.initSolver() || qp.solve()) return 1;
  qp.setSystem({2, 2, Pp, Pi, Pv}, q, {3, 2, Ap, Ai, Av}, l, u);
  if (!qp.initSolver()) return 3;
  if (!qp.solve()) return 1;
  const auto &x = qp.getSolution();
  std::printf("x = %.6f %.6f status %d iters %d\n", x[0], x[1], qp.getStatus()[0], qp.getIterations()[0]);
  if (!(std::fabs(x[0] - 0.5) < 5e-3 && std::fabs(x[1] - 0.5) < 5e-3 && qp.getStatus()[0] == MPCQP_SOLVED)) return 1;
  // the update* members (reference CuCaQP.cpp:106-161): new gradient and bounds on the kept workspace -> min (x1-1)^2 + x2^2 on
  // x1 + x2 = 2 is (1.5, 0.5); then a Hessian update (full setup again): 4 x1^2 + 2 x2^2 - 2 x1 -> (5/6, 7/6)
  const double q2[] = {-2, 0}, l2[] = {-50, -100, 2}, u2[] = {50, 100, 2};
  if (!qp.updateGradient(q2, 2) || !qp.updateLowerBound(l2, 3) || !qp.updateUpperBound(u2, 3) || !qp.solve()) return 1;
  std::printf("x = %.6f %.6f status %d iters %d\n", x[0], x[1], qp.getStatus()[0], qp.getIterations()[0]);
  if (!(std::fabs(x[0] - 1.5) < 5e-3 && std::fabs(x[1] - 0.5) < 5e-3)) return 1;
  if (qp.updateGradient(q2, 3)) return 1;                   // size mismatch must be refused (CuCaQP.cpp:122-126)
  const double Pv2[] = {8.0, 4.0};
  if (!qp.updateHessianMatrix({2, 2, Pp, Pi, Pv2}) || !qp.solve()) return 1;
  std::printf("x = %.6f %.6f status %d iters %d\n", x[0], x[1], qp.getStatus()[0], qp.getIterations()[0]);
  if (!(std::fabs(x[0] - 5.0 / 6.0) < 5e-3 && std::fabs(x[1] - 7.0 / 6.0) < 5e-3)) return 1;
  // setPresolveFixedRows (mpcqp_create_presolved): a variable pinned by a singleton row with l = u -- the shape of the reference's dp = 0 rows,
  // SQPOptimizationSolver.cpp:117 -- is found from the bounds and substituted: x1 = 0.3 exactly, x2 = 0.7 from x1 + x2 = 1
  CuCaQP pq;
  if (!pq.setDimension(2, 3)) return 1;
  pq.setAbsoluteTolerance(1e-3); pq.setRelativeTolerance(1e-3); pq.setMaxIteration(10000); pq.setPresolveFixedRows(true);
  const double lp[] = {0.3, -100, 1}, up[] = {0.3, 100, 1};
  pq.setSystem({2, 2, Pp, Pi, Pv}, q, {3, 2, Ap, Ai, Av}, lp, up);
  if (!pq.initSolver() || !pq.solve()) return 1;
  const auto &xp = pq.getSolution();
  std::printf("presolved rows %d: x = %.6f %.6f status %d\n", pq.presolvedRows(), xp[0], xp[1], pq.getStatus()[0]);
  return (pq.presolvedRows() == 1 && xp[0] == 0.3 && std::fabs(xp[1] - 0.7) < 5e-3 && pq.getStatus()[0] == MPCQP_SOLVED) ? 0 : 1;
}
