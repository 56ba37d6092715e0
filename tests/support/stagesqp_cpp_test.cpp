// C++ smoke of StageSQP.hpp: 32 double-integrator MPC instances, 3 full SQP steps (the problem is a QP, one step solves it).
// Exit code 0 = pass, 3 = no GPU (refused loudly), 1 = wrong answer.
#include <cmath>
#include <cstdio>
#include <limits>

#include "StageSQP.hpp"

int main() {
  mpcqp_stage_desc d;
  if (mpcqp_stage_default(MPCQP_MODEL_DOUBLE_INTEGRATOR, 20, &d) != MPCQP_OK) return 1;
  const int B = 32;
  try {
    StageSQP sqp(d, B, 3, 1.0);
    const int f = sqp.nx() + sqp.nu(), N = 20;
    const double inf = std::numeric_limits<double>::infinity();
    StageSQP::Arg a;
    a.p.assign((size_t)B * sqp.np(), 0.0); a.lbg.assign((size_t)B * sqp.ng(), 0.0); a.ubg.assign((size_t)B * sqp.ng(), 0.0);
    a.lbx.resize((size_t)B * sqp.nvar()); a.ubx.resize((size_t)B * sqp.nvar());
    for (int b = 0; b < B; b++)
      for (int k = 0; k < N; k++) {
        double *lo = &a.lbx[((size_t)b * N + k) * f], *hi = &a.ubx[((size_t)b * N + k) * f];
        lo[0] = -inf; hi[0] = inf; lo[1] = -2.0; hi[1] = 2.0; lo[2] = -1.0; hi[2] = 1.0;      // |v| <= 2, |u| <= 1
        if (k == 0) { lo[0] = hi[0] = -1.0 + 2.0 * b / (B - 1); lo[1] = hi[1] = 0.5; lo[2] = hi[2] = 0.0; }   // first frame pinned
      }
    StageSQP::Result r = sqp.getOptimalSolution(a);
    double worst = 0.0, fmaxv = 0.0;
    for (int b = 0; b < B; b++) {
      worst = std::fmax(worst, sqp.constraintViolation()[b]);
      fmaxv = std::fmax(fmaxv, r.f[b]);
      if (std::fabs(r.x[(size_t)b * sqp.nvar()] - (-1.0 + 2.0 * b / (B - 1))) > 5e-3) return 1;       // pinned position kept
      for (int k = 1; k < N; k++) if (std::fabs(r.x[((size_t)b * N + k) * f + 2]) > 1.0 + 1e-2) return 1;  // input bound honoured
    }
    std::printf("StageSQP: max dynamics violation %.2e, max objective %.4f\n", worst, fmaxv);
    return (worst < 5e-3 && std::isfinite(fmaxv)) ? 0 : 1;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "StageSQP: %s\n", e.what());
    return std::string(e.what()).find("gfx950") != std::string::npos || std::string(e.what()).find("no device") != std::string::npos ? 3 : 1;
  }
}
