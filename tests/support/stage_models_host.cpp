// CPU harness for tests: runs the product's scalar-generic model functors (csrc/stage_models.hpp) on the host --
// value and forward-mode Jacobian of the discrete stage map, and the CSC structure builder -- so the formulas the HIP
// kernels instantiate are checked against models.py without a GPU.
#include <vector>
#include "../../optimal_control_problem_amd/csrc/stage_models.hpp"

template <class M>
static void eval_model(const double *par, double h, const double *s, const double *u, double *out, double *jac) {
  constexpr int nx = M::nx, nu = M::nu, f = nx + nu;
  for (int c = 0; c < f; c++) {
    Dual sd[nx], ud[nu], od[nx];
    for (int i = 0; i < nx; i++) sd[i] = {s[i], i == c ? 1.0 : 0.0};
    for (int i = 0; i < nu; i++) ud[i] = {u[i], nx + i == c ? 1.0 : 0.0};
    M::template F<Dual>(par, h, sd, ud, od);
    for (int r = 0; r < nx; r++) { jac[r * f + c] = od[r].d; out[r] = od[r].v; }
  }
}

extern "C" {
// out [nx], jac [nx * (nx+nu)] row-major
int sm_host_eval(int model, const double *par, double h, const double *s, const double *u, double *out, double *jac) {
  switch (model) {
    case SM_DOUBLE_INTEGRATOR: eval_model<SmDoubleIntegrator>(par, h, s, u, out, jac); return 0;
    case SM_QUADROTOR: eval_model<SmQuadrotor>(par, h, s, u, out, jac); return 0;
    case SM_CARTPOLE: eval_model<SmCartPole>(par, h, s, u, out, jac); return 0;
  }
  return 1;
}
int sm_host_dims(int model, int *nx, int *nu) { sm_model_dims(model, nx, nu); return *nx ? 0 : 1; }
// two-call protocol: sizes first (pointers null), then fill
int sm_host_pattern(int model, int N, int *nnzP, int *nnzA, int *Pp, int *Pi, int *Ap, int *Ai) {
  int nx, nu; sm_model_dims(model, &nx, &nu);
  if (!nx) return 1;
  std::vector<int> a, b, c, d;
  sm_build_pattern(nx, nu, N, 0, 0, a, b, c, d);
  *nnzP = (int)b.size(); *nnzA = (int)d.size();
  if (Pp) { std::copy(a.begin(), a.end(), Pp); std::copy(b.begin(), b.end(), Pi); std::copy(c.begin(), c.end(), Ap); std::copy(d.begin(), d.end(), Ai); }
  return 0;
}
}
