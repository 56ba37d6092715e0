// StageSQP.hpp -- C++ form of the device-resident SQP loop over the C ABI (include/mpcqp.h): the reference's
// SQPOptimizationSolver::getOptimalSolution (reference src/sqp_solver/SQPOptimizationSolver.cpp:127-216) for a batch of
// instances of a stage OCP, with every step on the GPU:
//     mpcqp_stage_eval (getLocalSystem, :100-120) -> mpcqp_update(device) + mpcqp_solve (setSystem/initSolver/solve, :155-157)
//     -> mpcqp_stage_step (result.x += alpha * solution[pSize:], :171-177) -> mpcqp_stage_merit (objective, :180-181).
// Same quirks as the reference: a fixed number of iterations (step_num), no line search, the iterate persists across calls
// and starts at zero, arg.x0 is ignored.  Opt-in extension: setTolerance(t) adds a convergence stop that the reference has
// only under `verbose` (:183-197) -- the loop ends after the first iteration in which every instance moved by less than t
// (max-norm of alpha * dx, mpcqp_stage_step's step_max).  Header-only; needs the HIP runtime for the device buffers (hipMalloc / hipMemcpy).
#pragma once
#include <hip/hip_runtime_api.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "mpcqp.h"

class StageSQP {
 public:
  struct Arg { std::vector<double> lbx, ubx, lbg, ubg, p; };      // instance-major, as the DMDict of the reference carries them
  struct Result { std::vector<double> x, f; };

  // desc: model / horizon / dt / weights (mpcqp_stage_default fills the zoo's values); library_path: a generated dynamics
  // library (codegen.py) or nullptr for the built-in model named by desc.model
  StageSQP(const mpcqp_stage_desc &desc, int batch, int stepNum, double alpha, const char *library_path = nullptr)
      : batch_(batch), stepNum_(stepNum), alpha_(alpha) {
    check(library_path ? mpcqp_stage_create_user(&desc, library_path, &ocp_) : mpcqp_stage_create(&desc, &ocp_), "mpcqp_stage_create");
    check(mpcqp_stage_dims(ocp_, dims_), "mpcqp_stage_dims");
    std::vector<int> Pp(n() + 1), Pi(nnzP()), Ap(n() + 1), Ai(nnzA());
    check(mpcqp_stage_pattern(ocp_, Pp.data(), Pi.data(), Ap.data(), Ai.data()), "mpcqp_stage_pattern");
    mpcqp_settings st; mpcqp_default_settings(&st);               // eps 1e-3 / 1e-3, max_iter 10000 (reference :81-85)
    st.device = desc.device;
    check(mpcqp_create(n(), m(), batch_, Pp.data(), Pi.data(), Ap.data(), Ai.data(), &st, &qp_), "mpcqp_create");
    const size_t B = batch_;
    x_ = dalloc(B * nvar()); hip(hipMemset(x_, 0, B * nvar() * sizeof(double)));   // result_.x persists, zero-initialised (:88-91)
    dP_ = dalloc(B * nnzP()); dq_ = dalloc(B * n()); dA_ = dalloc(B * nnzA()); dl_ = dalloc(B * m()); du_ = dalloc(B * m());
    dw_ = dalloc(B * n()); f_ = dalloc(B); g_ = dalloc(B);
    p_ = dalloc(B * np()); lbx_ = dalloc(B * nvar()); ubx_ = dalloc(B * nvar()); lbg_ = dalloc(B * ng()); ubg_ = dalloc(B * ng());
  }
  ~StageSQP() {
    for (double *p : bufs_) (void)hipFree(p);
    if (qp_) mpcqp_destroy(qp_);
    if (ocp_) mpcqp_stage_destroy(ocp_);
  }
  StageSQP(const StageSQP &) = delete;
  StageSQP &operator=(const StageSQP &) = delete;

  int nx() const { return dims_[0]; } int nu() const { return dims_[1]; } int np() const { return dims_[2]; }
  int n() const { return dims_[3]; } int m() const { return dims_[4]; } int nnzP() const { return dims_[5]; }
  int nnzA() const { return dims_[6]; } int nvar() const { return dims_[7]; } int ng() const { return m() - n(); }
  bool generalCost() const { return mpcqp_stage_has_cost(ocp_) == 1; }   // the generated library carries its own stage cost

  // per-frame diagonal weights (terminal costs): Qk [horizon * nx], Rk [horizon * nu]
  void setWeights(const std::vector<double> &Qk, const std::vector<double> &Rk) {
    need(Qk.size(), (size_t)(nvar() / (nx() + nu())) * nx(), "Qk"); need(Rk.size(), (size_t)(nvar() / (nx() + nu())) * nu(), "Rk");
    check(mpcqp_stage_set_weights(ocp_, Qk.data(), Rk.data()), "mpcqp_stage_set_weights");
  }

  // per-frame bounds of the path constraint [horizon * nh] for the violation measure (terminal constraints: loose but on the last frame)
  void setPathBounds(const std::vector<double> &lo, const std::vector<double> &hi) {
    check(mpcqp_stage_set_path_bounds(ocp_, lo.data(), hi.data()), "mpcqp_stage_set_path_bounds");
  }

  void setInitialGuess(const std::vector<double> &x) {            // extension: the reference always starts from zero
    need(x.size(), (size_t)batch_ * nvar(), "x");
    hip(hipMemcpy(x_, x.data(), x.size() * sizeof(double), hipMemcpyHostToDevice));
  }

  void setTolerance(double tol) { tol_ = tol; }                   // 0 (default) = the reference's fixed iteration count
  int iterationsDone() const { return iterationsDone_; }

  Result getOptimalSolution(const Arg &arg) {
    const size_t B = batch_;
    need(arg.lbx.size(), B * nvar(), "lbx"); need(arg.ubx.size(), B * nvar(), "ubx");
    need(arg.lbg.size(), B * ng(), "lbg"); need(arg.ubg.size(), B * ng(), "ubg"); need(arg.p.size(), B * np(), "p");
    up(lbx_, arg.lbx); up(ubx_, arg.ubx); up(lbg_, arg.lbg); up(ubg_, arg.ubg); up(p_, arg.p);
    for (int i = 0; i < stepNum_; i++) {
      check(mpcqp_stage_eval(ocp_, batch_, p_, x_, lbx_, ubx_, lbg_, ubg_, dP_, dq_, dA_, dl_, du_, nullptr), "mpcqp_stage_eval");
      check(mpcqp_update(qp_, dP_, nnzP(), dq_, n(), dA_, nnzA(), dl_, m(), du_, m(), MPCQP_MEM_DEVICE), "mpcqp_update");
      check(mpcqp_solve(qp_, nullptr), "mpcqp_solve");
      check(mpcqp_get(qp_, dw_, nullptr, nullptr, nullptr, nullptr, nullptr, MPCQP_MEM_DEVICE), "mpcqp_get");
      check(mpcqp_stage_step(ocp_, batch_, alpha_, dw_, x_, tol_ > 0.0 ? g_ : nullptr, nullptr, nullptr), "mpcqp_stage_step");
      iterationsDone_ = i + 1;
      if (tol_ > 0.0) {                                           // one vector of step norms back to the host per iteration
        stepMax_.resize(B);
        hip(hipMemcpy(stepMax_.data(), g_, B * sizeof(double), hipMemcpyDeviceToHost));
        // (instances whose QP failed report a NaN step: they neither stop the loop nor keep it going; a batch without a finite step goes on)
        double worst = 0.0; bool any = false;
        for (double v : stepMax_) if (v == v) { any = true; if (v > worst) worst = v; }
        if (any && worst < tol_) break;
      }
    }
    check(mpcqp_stage_merit(ocp_, batch_, p_, x_, f_, g_, nullptr), "mpcqp_stage_merit");
    Result r; r.x.resize(B * nvar()); r.f.resize(B); violation_.resize(B);
    hip(hipMemcpy(r.x.data(), x_, r.x.size() * sizeof(double), hipMemcpyDeviceToHost));
    hip(hipMemcpy(r.f.data(), f_, B * sizeof(double), hipMemcpyDeviceToHost));
    hip(hipMemcpy(violation_.data(), g_, B * sizeof(double), hipMemcpyDeviceToHost));
    return r;
  }
  const std::vector<double> &constraintViolation() const { return violation_; }    // max-norm per instance after the last call

 private:
  static void hip(hipError_t e) { if (e != hipSuccess) throw std::runtime_error(std::string("HIP: ") + hipGetErrorString(e)); }
  static void check(int rc, const char *what) { if (rc != MPCQP_OK) throw std::runtime_error(std::string(what) + ": " + mpcqp_strerror(rc)); }
  static void need(size_t got, size_t want, const char *name) {
    if (got != want) throw std::invalid_argument(std::string(name) + " 的维度不对: expected " + std::to_string(want) + ", got " + std::to_string(got));
  }
  double *dalloc(size_t count) { double *p = nullptr; hip(hipMalloc(&p, (count ? count : 1) * sizeof(double))); bufs_.push_back(p); return p; }
  void up(double *dst, const std::vector<double> &src) { if (!src.empty()) hip(hipMemcpy(dst, src.data(), src.size() * sizeof(double), hipMemcpyHostToDevice)); }

  int batch_, stepNum_; double alpha_; double tol_ = 0.0; int iterationsDone_ = 0;
  std::vector<double> stepMax_;
  int dims_[8] = {0};
  mpcqp_stage *ocp_ = nullptr; mpcqp_handle *qp_ = nullptr;
  double *x_ = nullptr, *dP_ = nullptr, *dq_ = nullptr, *dA_ = nullptr, *dl_ = nullptr, *du_ = nullptr, *dw_ = nullptr, *f_ = nullptr, *g_ = nullptr;
  double *p_ = nullptr, *lbx_ = nullptr, *ubx_ = nullptr, *lbg_ = nullptr, *ubg_ = nullptr;
  std::vector<double *> bufs_;
  std::vector<double> violation_;
};
