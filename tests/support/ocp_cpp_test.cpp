// C++ test of OptimalControlProblem.hpp: "config" (no GPU: YAML subset, OCPConfig, validation, error behaviour) and
// "run" (GPU: a subclass written like the reference's examples, double-integrator MPC with a terminal weight, batch of 6).
// Exit code 0 = pass, 3 = no GPU (refused loudly), 1 = failure.
#include <cstdio>
#include <cstring>

#include "OptimalControlProblem.hpp"

// the README example of the reference (readme.md:43-62) completed with the keys the code reads
static const char *kYaml = R"(
optimal_control_problem:
  discretization_settings:
    dt: 0.05          # seconds
    horizon: 20
  solver_settings:
    verbose: false
    gen_code: false
    load_lib: false
    max_iter: 1000
    warm_start: true
    solve_method: CUDA_SQP
    SQP_settings:
      alpha: 1.0
      step_num: 2
  OCP_variables:
    - name: "state"
      size: 2
      lower_bound: [-.inf, -2.0]
      upper_bound: [.inf, 2.0]
    - name: input
      size: 1
      lower_bound: ["-1.0"]
      upper_bound: [1.0]
)";

class DoubleIntegratorOCP : public OptimalControlProblem {
 public:
  using OptimalControlProblem::OptimalControlProblem;
  void deployConstraintsAndAddCost() override {
    const OCPConfig &cfg = *OCPConfigPtr_;
    const int N = cfg.getHorizon();
    Reference ref = setReference(2);
    const StageModel plant = StageModel::builtIn(MPCQP_MODEL_DOUBLE_INTEGRATOR);
    for (int k = 0; k < N; k++) {
      addVectorCost(k < N - 1 ? std::vector<double>{10.0, 1.0} : std::vector<double>{200.0, 20.0}, cfg.getVariable(k, "state") - ref);   // terminal weight
      addVectorCost({0.1}, cfg.getVariable(k, "input"));
    }
    for (int k = 0; k < N - 1; k++)
      addEquationConstraint("dynamics", cfg.getVariable(k + 1, "state"), Dynamics{plant, cfg.getVariable(k, "state"), cfg.getVariable(k, "input")});
  }
};

#define EXPECT(cond) do { if (!(cond)) { std::fprintf(stderr, "FAILED: %s (line %d)\n", #cond, __LINE__); return 1; } } while (0)
template <class E, class F> static bool throws(F &&f) { try { f(); } catch (const E &) { return true; } catch (...) { return false; } return false; }

static int test_config() {
  YamlNode root = YamlNode::Load(kYaml);
  const YamlNode &node = root["optimal_control_problem"];
  EXPECT(node.IsMap() && node["OCP_variables"].IsSequence() && node["OCP_variables"].size() == 2);
  EXPECT(node["discretization_settings"]["dt"].as<double>() == 0.05 && node["solver_settings"]["SQP_settings"]["step_num"].as<int>() == 2);
  EXPECT(node["OCP_variables"][0]["name"].as<std::string>() == "state" && !node["nope"]);
  YamlNode flow = YamlNode::Load("a: {b: [1, 2.5, -.inf], c: 'x y'}\nd:\n- 1\n- {e: 2}\n");
  EXPECT(flow["a"]["b"].size() == 3 && std::isinf(flow["a"]["b"][2].as<double>()) && flow["a"]["c"].as<std::string>() == "x y");
  EXPECT(flow["d"].IsSequence() && flow["d"][1]["e"].as<int>() == 2);
  OCPConfig cfg(node);
  EXPECT(cfg.getHorizon() == 20 && cfg.getDt() == 0.05 && cfg.getFrameSize() == 3 && cfg.getVariables() == 60);
  EXPECT(cfg.getLowerBounds().size() == 20 && std::isinf(cfg.getLowerBounds()[7][0]) && cfg.getLowerBounds()[7][1] == -2.0 && cfg.getUpperBounds()[19][2] == 1.0);
  ocp_expr::Var v = cfg.getVariable(3, "input");
  EXPECT(v.start == 3 * 3 + 2 && v.stop == 3 * 3 + 3 && v.offset == 2);
  EXPECT(throws<std::out_of_range>([&] { cfg.getVariable(20, "state"); }));
  EXPECT(throws<std::invalid_argument>([&] { cfg.getVariable(0, "nope"); }));
  // error behaviour of the constructor (reference OptimalControlProblem.cpp:16-18,43-45)
  std::string bad = kYaml; bad.replace(bad.find("      alpha: 1.0\n"), std::strlen("      alpha: 1.0\n"), "");
  EXPECT(throws<std::runtime_error>([&] { DoubleIntegratorOCP o(YamlNode::Load(bad)["optimal_control_problem"]); }));
  std::string unk = kYaml; unk.replace(unk.find("CUDA_SQP"), 8, "NOPE");
  EXPECT(throws<std::invalid_argument>([&] { DoubleIntegratorOCP o(YamlNode::Load(unk)["optimal_control_problem"]); }));
  std::string nolb = kYaml; nolb.replace(nolb.find("      lower_bound: [-.inf, -2.0]\n"), std::strlen("      lower_bound: [-.inf, -2.0]\n"), "");
  EXPECT(throws<std::invalid_argument>([&] { OCPConfig c(YamlNode::Load(nolb)["optimal_control_problem"]); }));
  std::string ip = kYaml; ip.replace(ip.find("CUDA_SQP"), 8, "IPOPT");
  DoubleIntegratorOCP o(YamlNode::Load(ip)["optimal_control_problem"]);
  o.deployConstraintsAndAddCost();
  EXPECT(o.getSolverType() == OptimalControlProblem::SolverType::IPOPT && o.getConstraints() == 19 && o.getCostFunction() == 40);
  EXPECT(throws<std::runtime_error>([&] { o.genSolver(); }));             // third-party NLP arms are out of scope
  std::printf("config ok\n");
  return 0;
}

static int test_run() {
  const int B = 6;
  DoubleIntegratorOCP ocp(YamlNode::Load(kYaml)["optimal_control_problem"], B);
  ocp.deployConstraintsAndAddCost();
  try { ocp.genSolver(); }
  catch (const std::exception &e) {
    std::fprintf(stderr, "genSolver: %s\n", e.what());
    return std::string(e.what()).find("no usable gfx950") != std::string::npos ? 3 : 1;
  }
  std::vector<double> frame((size_t)B * 3), ref((size_t)B * 2, 0.0);
  for (int b = 0; b < B; b++) { frame[b * 3] = -1.0 + 0.4 * b; frame[b * 3 + 1] = 0.3; frame[b * 3 + 2] = 0.0; }
  EXPECT(throws<std::invalid_argument>([&] { ocp.computeOptimalTrajectory(std::vector<double>(B * 2), ref); }));   // "State dimension mismatch"
  const std::vector<double> &traj = ocp.computeOptimalTrajectory(frame, ref);
  EXPECT((int)traj.size() == B * 60);
  double worst = 0.0;
  for (int b = 0; b < B; b++) {
    EXPECT(std::fabs(traj[(size_t)b * 60] - frame[b * 3]) < 5e-3 && std::fabs(traj[(size_t)b * 60 + 1] - 0.3) < 5e-3);       // first frame pinned
    for (int k = 1; k < 20; k++) EXPECT(std::fabs(traj[(size_t)b * 60 + k * 3 + 2]) <= 1.0 + 1e-2 && std::fabs(traj[(size_t)b * 60 + k * 3 + 1]) <= 2.0 + 1e-2);
    worst = std::fmax(worst, ocp.constraintViolation()[b]);
    EXPECT(std::fabs(traj[(size_t)b * 60 + 19 * 3]) < std::fabs(frame[b * 3]) + 1e-9);   // the heavy terminal weight pulls the position towards the reference
  }
  EXPECT(worst < 5e-3);
  std::printf("run ok: max dynamics violation %.2e, terminal |position| of instance 0: %.4f\n", worst, std::fabs(traj[19 * 3]));
  return 0;
}

// general stage cost from a generated library (argv[2], made by codegen.trace(F, ..., lcost=, lterm=)): same YAML, the library's
// dynamics are the double integrator's; prints the trajectories for the caller to compare with the Python facade
class GeneralCostOCP : public OptimalControlProblem {
 public:
  GeneralCostOCP(const YamlNode &n, int batch, const std::string &lib) : OptimalControlProblem(n, batch), lib_(lib) {}
  void deployConstraintsAndAddCost() override {
    const OCPConfig &cfg = *OCPConfigPtr_;
    const int N = cfg.getHorizon();
    Reference ref = setReference(2);
    const StageModel plant = StageModel::fromLibrary(lib_);
    for (int k = 0; k < N; k++) {
      if (vectorCosts_) { addVectorCost({10.0, 1.0}, cfg.getVariable(k, "state") - ref); addVectorCost({0.1}, cfg.getVariable(k, "input")); }
      else addScalarCost(StageCost{plant, cfg.getVariable(k, "state"), cfg.getVariable(k, "input"), ref});
    }
    for (int k = 0; k < N - 1; k++)
      addEquationConstraint("dynamics", cfg.getVariable(k + 1, "state"), Dynamics{plant, cfg.getVariable(k, "state"), cfg.getVariable(k, "input")});
  }
  std::string lib_;
  bool vectorCosts_ = false;
};

static int test_cost(const char *lib) {
  const int B = 4;
  GeneralCostOCP ocp(YamlNode::Load(kYaml)["optimal_control_problem"], B, lib);
  ocp.deployConstraintsAndAddCost();
  EXPECT(ocp.getCostFunction() == 20);
  ocp.genSolver();
  std::vector<double> frame((size_t)B * 3), ref((size_t)B * 2, 0.0);
  for (int b = 0; b < B; b++) { frame[b * 3] = -0.8 + 0.5 * b; frame[b * 3 + 1] = 0.2; frame[b * 3 + 2] = 0.0; ref[b * 2] = 0.1 * b; }
  const std::vector<double> &traj = ocp.computeOptimalTrajectory(frame, ref);
  EXPECT((int)traj.size() == B * 60);
  for (int b = 0; b < B; b++) { std::printf("traj"); for (int i = 0; i < 60; i++) std::printf(" %.17g", traj[(size_t)b * 60 + i]); std::printf("\n"); }
  // the library carries its own cost: diagonal addVectorCost terms on top of it are refused
  GeneralCostOCP mixed(YamlNode::Load(kYaml)["optimal_control_problem"], B, lib);
  mixed.vectorCosts_ = true;
  mixed.deployConstraintsAndAddCost();
  EXPECT(throws<std::runtime_error>([&] { mixed.genSolver(); }));
  std::printf("cost ok\n");
  return 0;
}

// link constraint between consecutive frames (a rate limit on the input) from a generated library (argv[2], made by
// codegen.trace(F, ..., kfun=, nk=, k_lo=, k_hi=)): prints the trajectories for the caller to compare with the Python facade
class RateLimitOCP : public OptimalControlProblem {
 public:
  RateLimitOCP(const YamlNode &n, int batch, const std::string &lib) : OptimalControlProblem(n, batch), lib_(lib) {}
  void deployConstraintsAndAddCost() override {
    const OCPConfig &cfg = *OCPConfigPtr_;
    const int N = cfg.getHorizon();
    Reference ref = setReference(2);
    const StageModel plant = StageModel::fromLibrary(lib_);
    for (int k = 0; k < N; k++) { addVectorCost({10.0, 1.0}, cfg.getVariable(k, "state") - ref); addVectorCost({0.1}, cfg.getVariable(k, "input")); }
    for (int k = 0; k < N - 1; k++) {
      addEquationConstraint("dynamics", cfg.getVariable(k + 1, "state"), Dynamics{plant, cfg.getVariable(k, "state"), cfg.getVariable(k, "input")});
      if (links_on_) addInequalityConstraint("rate", {-0.15}, Link{plant, cfg.getVariable(k, "state"), cfg.getVariable(k, "input"),
                                                                   cfg.getVariable(k + 1, "state"), cfg.getVariable(k + 1, "input"), 1}, {0.15});
    }
  }
  std::string lib_;
  bool links_on_ = true;
};

static int test_link(const char *lib) {
  const int B = 6;
  RateLimitOCP ocp(YamlNode::Load(kYaml)["optimal_control_problem"], B, lib);
  ocp.deployConstraintsAndAddCost();
  EXPECT(ocp.getConstraints() == 19 + 19);
  ocp.genSolver();
  std::vector<double> frame((size_t)B * 3), ref((size_t)B * 2, 0.0);
  for (int b = 0; b < B; b++) { frame[b * 3] = -1.5 + 0.6 * b; frame[b * 3 + 1] = 0.4 - 0.15 * b; frame[b * 3 + 2] = 0.0; }
  const std::vector<double> &traj = ocp.computeOptimalTrajectory(frame, ref);
  EXPECT((int)traj.size() == B * 60);
  for (int b = 0; b < B; b++) {
    for (int k = 0; k + 1 < 20; k++) EXPECT(std::fabs(traj[(size_t)b * 60 + (k + 1) * 3 + 2] - traj[(size_t)b * 60 + k * 3 + 2]) <= 0.15 + 5e-3);   // the limit holds
    std::printf("traj"); for (int i = 0; i < 60; i++) std::printf(" %.17g", traj[(size_t)b * 60 + i]); std::printf("\n");
  }
  // the library carries a link constraint: leaving the Link terms out is refused (the row counts differ)
  RateLimitOCP without(YamlNode::Load(kYaml)["optimal_control_problem"], B, lib);
  without.links_on_ = false;
  without.deployConstraintsAndAddCost();
  EXPECT(throws<std::runtime_error>([&] { without.genSolver(); }));
  std::printf("link ok\n");
  return 0;
}

int main(int argc, char **argv) {
  try {
    if (argc > 1 && std::strcmp(argv[1], "config") == 0) return test_config();
    if (argc > 2 && std::strcmp(argv[1], "cost") == 0) return test_cost(argv[2]);
    if (argc > 2 && std::strcmp(argv[1], "link") == 0) return test_link(argv[2]);
    return test_run();
  } catch (const std::exception &e) { std::fprintf(stderr, "uncaught: %s\n", e.what()); return 1; }
}
