// The CasADi overloads of the CuCaQP facade (cpp/CuCaQP.hpp, compiled when <casadi/casadi.hpp> is found) through the literal call
// sequence of the reference's SQP loop (reference src/sqp_solver/SQPOptimizationSolver.cpp:80-85, 155-167):
//   setDimension -> setVerbosity / setWarmStart / setAbsoluteTolerance / setRelativeTolerance / setMaxIteration
//   -> per iteration: setSystem(DMVector{P, q, A, l, u}) -> initSolver() -> solve() -> getSolutionAsDM()
// against tests/support/casadi_mock (CasADi itself is not installed here).  The QP is the first-iteration QP of test/test.cpp case 1.
// Exit code 0 = pass, 3 = no GPU, 1 = wrong answer, 2 = the CasADi overloads were not compiled.
#include <cmath>
#include <cstdio>

#include "CuCaQP.hpp"

int main() {
#ifndef MPCQP_HAVE_CASADI
  return 2;
#else
  using casadi::DM; using casadi::DMVector; using casadi::Sparsity;
  CuCaQP qpSolver_;
  if (!qpSolver_.setDimension(2, 3)) return 1;
  qpSolver_.setVerbosity(false); qpSolver_.setWarmStart(true);
  qpSolver_.setAbsoluteTolerance(1e-3); qpSolver_.setRelativeTolerance(1e-3); qpSolver_.setMaxIteration(10000);
  const DM P(Sparsity(2, 2, {0, 1, 2}, {0, 1}), {2.0, 2.0});
  const DM A(Sparsity(3, 2, {0, 2, 4}, {0, 2, 1, 2}), {1, 1, 1, 1});
  double x0 = 0.0, x1 = 0.0;
  for (int it = 0; it < 2; it++) {                      // two "SQP iterations": the second shifts the bounds by the first step
    const DM q(std::vector<double>{2 * x0, 2 * x1});
    const DM l(std::vector<double>{-50 - x0, -100 - x1, 1 - x0 - x1}), u(std::vector<double>{50 - x0, 100 - x1, 1 - x0 - x1});
    DMVector localSystem{P, q, A, l, u};
    qpSolver_.setSystem(localSystem);
    if (!qpSolver_.initSolver()) return 3;
    if (!qpSolver_.solve()) return 1;
    DM solution = qpSolver_.getSolutionAsDM();
    if (solution.size1() != 2) return 1;
    x0 += solution.ptr()[0]; x1 += solution.ptr()[1];
    std::printf("iteration %d: dx = %.6f %.6f -> x = %.6f %.6f\n", it, solution.ptr()[0], solution.ptr()[1], x0, x1);
  }
  if (!(std::fabs(x0 - 0.5) < 5e-3 && std::fabs(x1 - 0.5) < 5e-3)) return 1;
  // the per-member overloads (reference CuCaQP.h:38-48), one by one, on a fresh object: the same QP member by member, then a wrong
  // dimension through each of them (false + message, like reference CuCaQP.cpp:43-103), then the DM forms of the update members
  CuCaQP m;
  if (!m.setDimension(2, 3)) return 1;
  m.setAbsoluteTolerance(1e-3); m.setRelativeTolerance(1e-3); m.setMaxIteration(10000);
  const DM q0(std::vector<double>{0.0, 0.0}), l0(std::vector<double>{-50, -100, 1}), u0(std::vector<double>{50, 100, 1});
  if (!m.setHessianMatrix(P) || !m.setGradient(q0) || !m.setLinearConstraintsMatrix(A) || !m.setLowerBound(l0) || !m.setUpperBound(u0)) return 1;
  if (!m.initSolver()) return 3;
  if (!m.solve()) return 1;
  DM s1 = m.getSolutionAsDM();
  std::printf("member by member: dx = %.6f %.6f\n", s1.ptr()[0], s1.ptr()[1]);
  if (!(std::fabs(s1.ptr()[0] - 0.5) < 5e-3 && std::fabs(s1.ptr()[1] - 0.5) < 5e-3)) return 1;
  // a sparse DM vector: entries outside its sparsity are zero (casadiDMToEigenVector reads element-wise)
  const DM qs(Sparsity(2, 1, {0, 1}, {1}), {-1.0});                  // q = (0, -1)
  if (!m.updateGradient(qs) || !m.updateLowerBound(l0) || !m.updateUpperBound(u0)) return 1;
  if (!m.solve()) return 1;
  DM s2 = m.getSolutionAsDM();                                       // min x0^2 + x1^2 - x1  s.t. x0 + x1 = 1  ->  (0.25, 0.75)
  std::printf("after updateGradient(DM): dx = %.6f %.6f\n", s2.ptr()[0], s2.ptr()[1]);
  if (!(std::fabs(s2.ptr()[0] - 0.25) < 5e-3 && std::fabs(s2.ptr()[1] - 0.75) < 5e-3)) return 1;
  if (!m.updateHessianMatrix(P) || !m.updateLinearConstraintsMatrix(A)) return 1;        // same pattern: accepted, full set-up at the next solve
  if (!m.solve()) return 1;
  const DM P3(Sparsity(3, 3, {0, 1, 2, 3}, {0, 1, 2}), {2.0, 2.0, 2.0}), v3(std::vector<double>{0.0, 0.0, 0.0}), v2(std::vector<double>{0.0, 0.0});
  const DM A2(Sparsity(2, 2, {0, 1, 2}, {0, 1}), {1.0, 1.0});
  CuCaQP w;
  if (!w.setDimension(2, 3)) return 1;
  if (w.setHessianMatrix(P3) || w.setGradient(v3) || w.setLinearConstraintsMatrix(A2) || w.setLowerBound(v2) || w.setUpperBound(v2)) return 1;   // every one refuses a wrong size
  return 0;
#endif
}
