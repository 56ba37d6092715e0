// OptimalControlProblem.hpp -- the reference's OCP-level host API in its own language, over the GPU engine.
//
// Mirrors include/optimal_control_problem/OCP_config/OCPConfig.h:37-85 (+ src/OCP_config/OCPConfig.cpp:29-249) and
// include/optimal_control_problem/OptimalControlProblem.h:13-107 (+ src/OptimalControlProblem.cpp) for the CUDA_SQP solve
// method: same YAML keys, same class and member names, same call order
//     class MyOCP : public OptimalControlProblem { void deployConstraintsAndAddCost() override { ... } };
//     MyOCP ocp(node); ocp.deployConstraintsAndAddCost(); ocp.genSolver(); ocp.computeOptimalTrajectory(frame, reference);
// Three things differ, all forced by what is (not) installed: (1) yaml-cpp is absent, so `YamlNode` below parses the YAML
// subset the reference's config files use (block and flow maps / sequences, scalars, .inf, comments); (2) CasADi is absent,
// so the SX expressions the builders take are replaced by a small expression layer that covers stage-structured OCPs --
// a variable slice of a frame (OCPConfig::getVariable), the reference parameter, their difference, the discrete dynamics
// F(state_k, input_k) of a compiled model and a per-frame path constraint; (3) one object may carry a batch of independent
// instances (frame and reference are instance-major), batch = 1 being the drop-in case.
// genSolver() checks that the problem has the stage structure the device evaluator handles (tracking cost with diagonal,
// possibly per-step weights; dynamics defects between consecutive frames; one path constraint per frame) and builds the
// device-resident SQP loop (StageSQP.hpp).  The IPOPT / SQP(qpOASES) / MIXED arms are third-party NLP solvers: out of scope.
#pragma once
#include <cmath>
#include <fstream>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "StageSQP.hpp"

// ------------------------------------------------------------------------------------------------ YAML subset
class YamlNode {
 public:
  enum Kind { Null, Scalar, Map, Seq };
  Kind kind = Null;
  std::string scalar;
  std::vector<std::pair<std::string, YamlNode>> map;
  std::vector<YamlNode> seq;

  bool IsDefined() const { return kind != Null; }
  bool IsSequence() const { return kind == Seq; }
  bool IsMap() const { return kind == Map; }
  explicit operator bool() const { return IsDefined(); }
  size_t size() const { return kind == Seq ? seq.size() : kind == Map ? map.size() : 0; }
  const YamlNode &operator[](const std::string &key) const {
    static const YamlNode none;
    if (kind == Map) for (auto &kv : map) if (kv.first == key) return kv.second;
    return none;
  }
  const YamlNode &operator[](const char *key) const { return (*this)[std::string(key)]; }
  const YamlNode &operator[](size_t i) const { static const YamlNode none; return kind == Seq && i < seq.size() ? seq[i] : none; }
  const YamlNode &operator[](int i) const { return (*this)[(size_t)i]; }

  template <class T> T as() const;

  static YamlNode Load(const std::string &text) { Parser p(text); return p.parse(); }
  static YamlNode LoadFile(const std::string &path) {
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open YAML file: " + path);
    std::stringstream ss; ss << in.rdbuf();
    return Load(ss.str());
  }

 private:
  struct Line { int indent; std::string text; };
  class Parser {
   public:
    explicit Parser(const std::string &text) {
      std::stringstream ss(text); std::string raw;
      while (std::getline(ss, raw)) {
        std::string s = stripComment(raw);
        size_t a = s.find_first_not_of(" \t");
        if (a == std::string::npos) continue;
        size_t b = s.find_last_not_of(" \t\r");
        lines_.push_back({(int)a, s.substr(a, b - a + 1)});
      }
    }
    YamlNode parse() { return lines_.empty() ? YamlNode() : block(lines_[0].indent); }

   private:
    std::vector<Line> lines_; size_t pos_ = 0;
    static std::string stripComment(const std::string &s) {
      bool sq = false, dq = false;
      for (size_t i = 0; i < s.size(); i++) {
        if (s[i] == '\'' && !dq) sq = !sq; else if (s[i] == '"' && !sq) dq = !dq;
        else if (s[i] == '#' && !sq && !dq && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) return s.substr(0, i);
      }
      return s;
    }
    static std::string trim(const std::string &s) {
      size_t a = s.find_first_not_of(" \t"), b = s.find_last_not_of(" \t");
      return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
    }
    static std::string unquote(const std::string &s) {
      if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) return s.substr(1, s.size() - 2);
      return s;
    }
    // flow value: [a, b], {k: v, ...} or a scalar; `i` advances over it
    static YamlNode flow(const std::string &s, size_t &i) {
      while (i < s.size() && s[i] == ' ') i++;
      YamlNode n;
      if (i < s.size() && s[i] == '[') {
        n.kind = Seq; i++;
        for (;;) {
          while (i < s.size() && s[i] == ' ') i++;
          if (i >= s.size()) throw std::runtime_error("YAML: unterminated [");
          if (s[i] == ']') { i++; break; }
          n.seq.push_back(flow(s, i));
          while (i < s.size() && s[i] == ' ') i++;
          if (i < s.size() && s[i] == ',') i++;
        }
      } else if (i < s.size() && s[i] == '{') {
        n.kind = Map; i++;
        for (;;) {
          while (i < s.size() && s[i] == ' ') i++;
          if (i >= s.size()) throw std::runtime_error("YAML: unterminated {");
          if (s[i] == '}') { i++; break; }
          size_t c = s.find(':', i);
          if (c == std::string::npos) throw std::runtime_error("YAML: expected key: value inside {}");
          std::string key = unquote(trim(s.substr(i, c - i))); i = c + 1;
          n.map.emplace_back(key, flow(s, i));
          while (i < s.size() && s[i] == ' ') i++;
          if (i < s.size() && s[i] == ',') i++;
        }
      } else {
        size_t a = i; bool sq = false, dq = false;
        while (i < s.size()) {
          if (s[i] == '\'' && !dq) sq = !sq; else if (s[i] == '"' && !sq) dq = !dq;
          else if (!sq && !dq && (s[i] == ',' || s[i] == ']' || s[i] == '}')) break;
          i++;
        }
        std::string v = trim(s.substr(a, i - a));
        if (!v.empty() && v != "~" && v != "null") { n.kind = Scalar; n.scalar = unquote(v); }
      }
      return n;
    }
    static YamlNode value(const std::string &text) { size_t i = 0; return flow(text, i); }
    // position of the ':' that ends a block-map key (followed by space or end of line, outside quotes / brackets)
    static size_t keyColon(const std::string &t) {
      bool sq = false, dq = false; int depth = 0;
      for (size_t i = 0; i < t.size(); i++) {
        char ch = t[i];
        if (ch == '\'' && !dq) sq = !sq; else if (ch == '"' && !sq) dq = !dq;
        else if (!sq && !dq) {
          if (ch == '[' || ch == '{') depth++; else if (ch == ']' || ch == '}') depth--;
          else if (ch == ':' && depth == 0 && (i + 1 == t.size() || t[i + 1] == ' ')) return i;
        }
      }
      return std::string::npos;
    }
    YamlNode block(int indent) {
      YamlNode n;
      if (pos_ >= lines_.size()) return n;
      if (lines_[pos_].text.rfind("- ", 0) == 0 || lines_[pos_].text == "-") {
        n.kind = Seq;
        while (pos_ < lines_.size() && lines_[pos_].indent == indent && (lines_[pos_].text.rfind("- ", 0) == 0 || lines_[pos_].text == "-")) {
          std::string rest = lines_[pos_].text.size() > 1 ? trim(lines_[pos_].text.substr(2)) : std::string();
          const int inner = indent + 2;
          if (rest.empty()) { pos_++; n.seq.push_back(pos_ < lines_.size() && lines_[pos_].indent > indent ? block(lines_[pos_].indent) : YamlNode()); }
          else if (rest[0] != '[' && rest[0] != '{' && keyColon(rest) != std::string::npos) {
            lines_[pos_] = {inner, rest};              // "- key: value" opens a map whose further keys are indented by 2
            n.seq.push_back(block(inner));
          } else { pos_++; n.seq.push_back(value(rest)); }
        }
        return n;
      }
      n.kind = Map;
      while (pos_ < lines_.size() && lines_[pos_].indent == indent) {
        const std::string &t = lines_[pos_].text;
        size_t c = keyColon(t);
        if (c == std::string::npos) throw std::runtime_error("YAML: expected 'key: value', got: " + t);
        std::string key = unquote(trim(t.substr(0, c))), rest = trim(t.substr(c + 1));
        pos_++;
        if (!rest.empty()) n.map.emplace_back(key, value(rest));
        else if (pos_ < lines_.size() && (lines_[pos_].indent > indent || (lines_[pos_].indent == indent && lines_[pos_].text.rfind("- ", 0) == 0)))
          n.map.emplace_back(key, block(lines_[pos_].indent));
        else n.map.emplace_back(key, YamlNode());
      }
      return n;
    }
  };
};

template <> inline std::string YamlNode::as<std::string>() const {
  if (kind != Scalar) throw std::runtime_error("YAML: bad conversion (not a scalar)");
  return scalar;
}
template <> inline double YamlNode::as<double>() const {
  const std::string s = as<std::string>();
  if (s == ".inf" || s == ".Inf" || s == ".INF" || s == "+.inf") return std::numeric_limits<double>::infinity();
  if (s == "-.inf" || s == "-.Inf" || s == "-.INF") return -std::numeric_limits<double>::infinity();
  size_t used = 0; double v = 0;
  try { v = std::stod(s, &used); } catch (...) { used = 0; }
  if (used != s.size() || s.empty()) throw std::runtime_error("YAML: bad conversion to double: " + s);
  return v;
}
template <> inline int YamlNode::as<int>() const {
  const std::string s = as<std::string>(); size_t used = 0; int v = 0;
  try { v = std::stoi(s, &used); } catch (...) { used = 0; }
  if (used != s.size() || s.empty()) throw std::runtime_error("YAML: bad conversion to int: " + s);
  return v;
}
template <> inline bool YamlNode::as<bool>() const {
  const std::string s = as<std::string>();
  if (s == "true" || s == "True" || s == "TRUE" || s == "yes" || s == "on") return true;
  if (s == "false" || s == "False" || s == "FALSE" || s == "no" || s == "off") return false;
  throw std::runtime_error("YAML: bad conversion to bool: " + s);
}

// ------------------------------------------------------------------------------------------------ expression layer
namespace ocp_expr {
struct Var { int step = 0; std::string name; int start = 0, stop = 0, offset = 0; int size() const { return stop - start; } };
struct Reference { int n = 0; int size() const { return n; } };
struct Diff { Var a; Reference b; int size() const { return a.size(); } };               // variable - reference
inline Diff operator-(const Var &a, const Reference &b) { return Diff{a, b}; }
// compiled dynamics / path constraint: a built-in zoo model (MPCQP_MODEL_*) or a generated library (codegen.py / a hand-written
// functor in the form of csrc/stage_models.hpp built with hipcc) -- what the reference gets from CasADi code generation
struct StageModel { int builtin = -1; std::string library; double par[8] = {0};
  static StageModel builtIn(int id) { StageModel m; m.builtin = id; return m; }
  static StageModel fromLibrary(const std::string &path) { StageModel m; m.library = path; return m; } };
struct Dynamics { StageModel model; Var state, input; int size() const { return state.size(); } };   // F(state_k, input_k)
struct Path { StageModel model; Var state, input; int rows = 0; int size() const { return rows; } };  // h(state_k, input_k)
// k(state_k, input_k, state_{k+1}, input_{k+1}) of two consecutive frames (a rate limit u_{k+1} - u_k, a slew limit on a state), carried by the
// generated library of the dynamics (codegen.trace(F, ..., kfun=, nk=, k_lo=, k_hi=)): the stand-in for an SX expression over several
// frames (src/OptimalControlProblem.cpp:448-489); ocp.py's Link
struct Link { StageModel model; Var state, input, stateNext, inputNext; int rows = 0; int size() const { return rows; } };
struct Cost { int kind = 0; Var var; bool minusReference = false; std::vector<double> weight; };       // sum_i w_i e_i^2
// general cost term l(state_k, input_k, reference) of one frame, carried by the generated library of the dynamics (codegen.trace with
// lcost / lterm): the stand-in for an arbitrary SX term of the reference (src/OptimalControlProblem.cpp:491-497)
struct StageCost { StageModel model; Var state, input; Reference reference; int size() const { return 1; } };
}  // namespace ocp_expr

// ------------------------------------------------------------------------------------------------ OCPConfig
class OCPConfig {
 public:
  explicit OCPConfig(const YamlNode &configNode) {                                  // reference OCPConfig.cpp:83-105
    dt_ = configNode["discretization_settings"]["dt"].as<double>();
    horizon_ = configNode["discretization_settings"]["horizon"].as<int>();
    verbose_ = configNode["solver_settings"]["verbose"].as<bool>();
    const YamlNode &frame = configNode["OCP_variables"];
    if (!frame) throw std::invalid_argument("node [OCP_variables] not found in YAML file");          // :113-116
    if (!frame.IsSequence()) throw std::invalid_argument("status_frame should be a sequence");         // :120-123
    std::vector<double> lo, hi;
    for (size_t v = 0; v < frame.size(); v++) {                                     // initializeFrame, :56-81
      const YamlNode &var = frame[v];
      if (!var["name"]) throw std::invalid_argument("Field name not found in frame");
      if (!var["size"]) throw std::invalid_argument("Field size not found in frame");
      const std::string name = var["name"].as<std::string>(); const int size = var["size"].as<int>();
      if (size <= 0) throw std::invalid_argument("Field size must be positive: " + name);
      fields_.emplace_back(name, size); offsets_[name] = totalSize_; totalSize_ += size;
      for (int side = 0; side < 2; side++) {
        const char *key = side ? "upper_bound" : "lower_bound";
        if (!var[key]) throw std::invalid_argument(std::string("Missing ") + key + " for variable: " + name);   // :138-141,180-183
        const YamlNode &seq = var[key];
        for (int i = 0; i < size; i++) (side ? hi : lo).push_back(seq.IsSequence() && (size_t)i < seq.size() ? seq[i].as<double>() : 0.0);   // size mismatch only warns, :147-151
      }
    }
    lowerBounds_.assign(horizon_, lo); upperBounds_.assign(horizon_, hi);            // coverLowerBounds: one frame x horizon
  }
  ocp_expr::Var getVariable(int stepID, const std::string &variableName) const {   // :29-46
    if (stepID < 0 || stepID >= horizon_) throw std::out_of_range("Frame ID out of range");
    auto it = offsets_.find(variableName);
    if (it == offsets_.end()) throw std::invalid_argument("Field name not found in frame");
    int size = 0; for (auto &f : fields_) if (f.first == variableName) size = f.second;
    const int start = stepID * totalSize_ + it->second;
    return ocp_expr::Var{stepID, variableName, start, start + size, it->second};
  }
  int getVariables() const { return horizon_ * totalSize_; }
  const std::vector<std::vector<double>> &getLowerBounds() const { return lowerBounds_; }
  const std::vector<std::vector<double>> &getUpperBounds() const { return upperBounds_; }
  int getHorizon() const { return horizon_; }
  double getDt() const { return dt_; }
  int getFrameSize() const { return totalSize_; }
  bool verbose() const { return verbose_; }
  void setInitialGuess(const std::vector<double> &g) { initialGuess_ = g; }
  const std::vector<double> &getInitialGuess() const { return initialGuess_; }

 private:
  int horizon_ = 10; double dt_ = 0.1; bool verbose_ = false; int totalSize_ = 0;
  std::vector<std::pair<std::string, int>> fields_; std::map<std::string, int> offsets_;
  std::vector<std::vector<double>> lowerBounds_, upperBounds_; std::vector<double> initialGuess_;
};

// ------------------------------------------------------------------------------------------------ OptimalControlProblem
class OptimalControlProblem {
 public:
  enum class SolverType { IPOPT, SQP, CUDA_SQP, MIXED };
  using Var = ocp_expr::Var; using Reference = ocp_expr::Reference; using Diff = ocp_expr::Diff;
  using Dynamics = ocp_expr::Dynamics; using Path = ocp_expr::Path; using StageModel = ocp_expr::StageModel; using StageCost = ocp_expr::StageCost; using Link = ocp_expr::Link;

  std::unique_ptr<OCPConfig> OCPConfigPtr_;
  Reference reference_;

  explicit OptimalControlProblem(const YamlNode &configNode, int batch = 1) : batch_(batch) {
    if (!validateConfig(configNode)) throw std::runtime_error("Invalid configuration file");        // reference OptimalControlProblem.cpp:16-18
    OCPConfigPtr_.reset(new OCPConfig(configNode));
    const YamlNode &s = configNode["solver_settings"];
    maxIter_ = s["max_iter"].as<int>(); warmStart_ = s["warm_start"].as<bool>(); verbose_ = s["verbose"].as<bool>();
    genCode_ = s["gen_code"].as<bool>(); loadLib_ = s["load_lib"].as<bool>();
    alpha_ = s["SQP_settings"]["alpha"].as<double>(); stepNum_ = s["SQP_settings"]["step_num"].as<int>();
    const std::string method = s["solve_method"].as<std::string>();                                   // :31-43
    if (method == "IPOPT") solverType_ = SolverType::IPOPT; else if (method == "MIXED") solverType_ = SolverType::MIXED;
    else if (method == "SQP") solverType_ = SolverType::SQP; else if (method == "CUDA_SQP") solverType_ = SolverType::CUDA_SQP;
    else throw std::invalid_argument("Unknown solver type: " + method);
  }
  virtual ~OptimalControlProblem() = default;
  virtual void deployConstraintsAndAddCost() = 0;                                   // reference OptimalControlProblem.h:101

  void setSolverType(SolverType t) { solverType_ = t; }                             // :499-505
  SolverType getSolverType() const { return solverType_; }
  Reference setReference(int size) { reference_ = Reference{size}; return reference_; }
  Reference getReference() const { return reference_; }
  const std::vector<double> &getOptimalTrajectory() const { return optimalTrajectory_; }

  // builders (:444-497,574-600)
  void addVectorCost(const std::vector<double> &param, const Diff &cost) {
    if ((int)param.size() != cost.size()) { std::cout << "损失的符号向量和参数向量维度不一致" << std::endl; return; }   // :576-579 (prints and returns)
    costs_.push_back(ocp_expr::Cost{0, cost.a, true, param});
  }
  void addVectorCost(const std::vector<double> &param, const Var &cost) {
    if ((int)param.size() != cost.size()) { std::cout << "损失的符号向量和参数向量维度不一致" << std::endl; return; }
    costs_.push_back(ocp_expr::Cost{0, cost, false, param});
  }
  void addScalarCost(const StageCost &cost) { stageCosts_.push_back(cost); }          // :491-497
  void addEquationConstraint(const std::string &constraintName, const Var &leftSX, const Dynamics &rightSX) {
    if (leftSX.size() != rightSX.size()) throw std::invalid_argument("SX used for constraints has different dimension!");   // :472-474
    dynamics_.push_back({leftSX, rightSX}); constraintNames_.insert(constraintNames_.end(), leftSX.size(), constraintName);
  }
  void addInequalityConstraint(const std::string &constraintName, const std::vector<double> &lowerBound, const Path &expression,
                               const std::vector<double> &upperBound) {
    if ((int)lowerBound.size() != expression.size() || (int)upperBound.size() != expression.size())
      throw std::invalid_argument("SX used for inequality constraints has different dimensions!");    // :452-454
    paths_.push_back({expression, lowerBound, upperBound}); constraintNames_.insert(constraintNames_.end(), expression.size(), constraintName);
  }
  void addInequalityConstraint(const std::string &constraintName, const std::vector<double> &lowerBound, const Link &expression,
                               const std::vector<double> &upperBound) {
    if ((int)lowerBound.size() != expression.size() || (int)upperBound.size() != expression.size())
      throw std::invalid_argument("SX used for inequality constraints has different dimensions!");    // :452-454
    links_.push_back({expression, lowerBound, upperBound}); constraintNames_.insert(constraintNames_.end(), expression.size(), constraintName);
  }
  size_t getConstraints() const { return dynamics_.size() + paths_.size() + links_.size(); }
  size_t getCostFunction() const { return costs_.size() + stageCosts_.size(); }

  // genSolver (:224-442), CUDA_SQP arm :391-401
  void genSolver() {
    const OCPConfig &cfg = *OCPConfigPtr_;
    if (cfg.getVariables() == 0) throw std::runtime_error("Status or input variables are empty");
    if (dynamics_.empty() && paths_.empty()) throw std::runtime_error("Constraints are empty");        // :231-233
    if (solverType_ != SolverType::CUDA_SQP)
      throw std::runtime_error("this solve_method relies on third-party NLP solvers (IPOPT / qpOASES) and is out of scope; use CUDA_SQP");
    const int N = cfg.getHorizon(), f = cfg.getFrameSize();
    if ((int)dynamics_.size() != N - 1 || !(paths_.empty() || (int)paths_.size() == N) || !(links_.empty() || (int)links_.size() == N - 1))
      throw std::runtime_error("this facade compiles dynamics defects x_{k+1} - F(x_k, u_k) between consecutive frames, one per-frame path constraint and one link constraint per pair of consecutive frames");
    const Var s0 = dynamics_[0].second.state, u0 = dynamics_[0].second.input;
    const int nx = s0.size(), nu = u0.size();
    if (s0.offset != 0 || u0.offset != nx || nx + nu != f) throw std::runtime_error("frame layout must be [state; input]");
    std::vector<char> seen(N, 0);
    for (auto &d : dynamics_) {
      const int k = d.second.state.step;
      if (k < 0 || k >= N - 1 || seen[k] || d.first.step != k + 1 || d.first.name != s0.name || d.second.input.step != k || d.second.state.name != s0.name)
        throw std::runtime_error("dynamics constraints must link frame k to frame k + 1");
      seen[k] = 1;
    }
    const StageModel &mdl = dynamics_[0].second.model;
    // weights: per-step sums of the addVectorCost terms (the reference sums SX terms, :491-497)
    std::vector<double> Qk((size_t)N * nx, 0.0), Rk((size_t)N * nu, 0.0);
    std::vector<char> hasQ(N, 0), hasR(N, 0);
    for (auto &c : costs_) {
      if (c.minusReference && c.var.name == s0.name) { for (int i = 0; i < nx; i++) Qk[(size_t)c.var.step * nx + i] += c.weight[i]; hasQ[c.var.step] = 1; }
      else if (!c.minusReference && c.var.name == u0.name) { for (int i = 0; i < nu; i++) Rk[(size_t)c.var.step * nu + i] += c.weight[i]; hasR[c.var.step] = 1; }
      else throw std::runtime_error("cost term not recognised: use (state - reference) and (input) terms");
    }
    const bool general = !stageCosts_.empty();
    if (general) {
      // the objective lives in the generated library: one StageCost per frame, from the library that also holds the dynamics
      if (!costs_.empty() || (int)stageCosts_.size() != N || mdl.builtin >= 0) throw std::runtime_error("general costs: exactly one StageCost term per frame, no other cost terms, dynamics from a generated library");
      std::vector<char> got(N, 0);
      for (auto &c : stageCosts_) {
        const int k = c.state.step;
        if (k < 0 || k >= N || got[k] || c.input.step != k || c.state.name != s0.name || c.input.name != u0.name || c.model.library != mdl.library)
          throw std::runtime_error("a StageCost takes the state and the input of its own frame and the library of the dynamics");
        got[k] = 1;
      }
    }
    for (int k = 0; k < N && !general; k++) if (!hasQ[k] || !hasR[k]) throw std::runtime_error("tracking and input costs must be added for every step");
    if (reference_.size() != nx) throw std::runtime_error("reference must have the state's dimension");
    mpcqp_stage_desc d;
    if (mpcqp_stage_default(mdl.builtin >= 0 ? mdl.builtin : 0, N, &d) != MPCQP_OK) throw std::runtime_error(mpcqp_strerror(MPCQP_ERR_ARG));
    d.dt = cfg.getDt();
    for (int i = 0; i < 16; i++) d.Q[i] = i < nx ? Qk[i] : 0.0;
    for (int i = 0; i < 8; i++) d.R[i] = i < nu ? Rk[i] : 0.0;
    if (mdl.builtin >= 0) { bool any = false; for (double v : mdl.par) any |= v != 0.0; if (any) for (int i = 0; i < 8; i++) d.par[i] = mdl.par[i]; }
    solver_.reset(new StageSQP(d, batch_, stepNum_, alpha_, mdl.builtin >= 0 ? nullptr : mdl.library.c_str()));
    if (solver_->nx() != nx || solver_->nu() != nu) throw std::runtime_error("the compiled model's state / input sizes differ from the YAML frame");
    nh_ = paths_.empty() ? 0 : paths_[0].expr.size();
    nk_ = links_.empty() ? 0 : links_[0].expr.size();
    if (nk_ > 0) {
      std::vector<char> got(N, 0);
      for (auto &l : links_) {
        const int k = l.expr.state.step;
        if (k < 0 || k >= N - 1 || got[k] || l.expr.input.step != k || l.expr.stateNext.step != k + 1 || l.expr.inputNext.step != k + 1 ||
            l.expr.size() != nk_ || l.expr.model.library != mdl.library || mdl.builtin >= 0)
          throw std::runtime_error("a Link takes the state and input of frames k and k + 1 and lives in the library of the dynamics");
        got[k] = 1;
      }
    }
    // rows of the constraint vector: dynamics defects, path rows, link rows (csrc/stage_kernels.hpp)
    if (solver_->ng() != (N - 1) * nx + N * nh_ + (N - 1) * nk_) throw std::runtime_error("the compiled model's path / link constraints differ from the ones added");
    if (nh_ > 0) {   // bounds may differ by frame (a terminal constraint is loose on every frame but the last): tell the violation measure
      std::vector<double> lo((size_t)N * nh_), hi((size_t)N * nh_);
      for (auto &p : paths_) for (int r = 0; r < nh_; r++) { lo[(size_t)p.expr.state.step * nh_ + r] = p.lo[r]; hi[(size_t)p.expr.state.step * nh_ + r] = p.hi[r]; }
      solver_->setPathBounds(lo, hi);
    }
    bool same = true;
    for (int k = 1; k < N && same; k++) { for (int i = 0; i < nx; i++) same &= Qk[(size_t)k * nx + i] == Qk[i]; for (int i = 0; i < nu; i++) same &= Rk[(size_t)k * nu + i] == Rk[i]; }
    if (general != solver_->generalCost())
      throw std::runtime_error(general ? "StageCost terms were added but the library was generated without a stage cost"
                                       : "the library carries its own stage cost: add StageCost terms instead of addVectorCost");
    if (!same && !general) solver_->setWeights(Qk, Rk);
    nx_ = nx; nu_ = nu;
  }

  // computeOptimalTrajectory (:78-222), CUDA_SQP arm; frame [batch * frameSize], reference [batch * nx], both instance-major
  const std::vector<double> &computeOptimalTrajectory(const std::vector<double> &frame, const std::vector<double> &reference) {
    const OCPConfig &cfg = *OCPConfigPtr_;
    const int fs = cfg.getFrameSize(), N = cfg.getHorizon(), nv = cfg.getVariables();
    if ((int)frame.size() != batch_ * fs)
      throw std::invalid_argument("State dimension mismatch: received " + std::to_string(frame.size() / (size_t)std::max(batch_, 1)) + ", expected " + std::to_string(fs));   // :79-84
    if ((int)reference.size() != batch_ * reference_.size())
      throw std::invalid_argument("Reference dimension mismatch: received " + std::to_string(reference.size() / (size_t)std::max(batch_, 1)) + ", expected " + std::to_string(reference_.size()));   // :85-90
    if (!solver_) throw std::runtime_error("Optimization failed: genSolver() has not been called");
    StageSQP::Arg arg;
    arg.lbx.resize((size_t)batch_ * nv); arg.ubx.resize((size_t)batch_ * nv);
    for (int b = 0; b < batch_; b++)
      for (int k = 0; k < N; k++)
        for (int i = 0; i < fs; i++) {
          const size_t at = ((size_t)b * N + k) * fs + i;
          arg.lbx[at] = k == 0 ? frame[(size_t)b * fs + i] : cfg.getLowerBounds()[k][i];               // :95-96 the whole first frame is pinned
          arg.ubx[at] = k == 0 ? frame[(size_t)b * fs + i] : cfg.getUpperBounds()[k][i];
        }
    const int ngd = (N - 1) * nx_, ngh = ngd + N * nh_, ng = ngh + (N - 1) * nk_;
    arg.lbg.assign((size_t)batch_ * ng, 0.0); arg.ubg.assign((size_t)batch_ * ng, 0.0);                 // dynamics rows: [0, 0]
    for (auto &l : links_)
      for (int b = 0; b < batch_; b++)
        for (int r = 0; r < nk_; r++) {
          arg.lbg[(size_t)b * ng + ngh + (size_t)l.expr.state.step * nk_ + r] = l.lo[r];
          arg.ubg[(size_t)b * ng + ngh + (size_t)l.expr.state.step * nk_ + r] = l.hi[r];
        }
    for (auto &p : paths_)
      for (int b = 0; b < batch_; b++)
        for (int r = 0; r < nh_; r++) {
          arg.lbg[(size_t)b * ng + ngd + (size_t)p.expr.state.step * nh_ + r] = p.lo[r];
          arg.ubg[(size_t)b * ng + ngd + (size_t)p.expr.state.step * nh_ + r] = p.hi[r];
        }
    arg.p = reference;
    try { optimalTrajectory_ = solver_->getOptimalSolution(arg).x; }                                      // :141
    catch (const std::exception &e) { throw std::runtime_error(std::string("Optimization failed: ") + e.what()); }   // :219-221
    firstTime_ = false;
    return optimalTrajectory_;
  }
  const std::vector<double> &constraintViolation() const { return solver_->constraintViolation(); }

 private:
  static bool validateConfig(const YamlNode &config) {                              // :54-62
    const YamlNode &s = config["solver_settings"];
    if (!s.IsMap()) return false;
    for (const char *k : {"max_iter", "warm_start", "SQP_settings", "verbose", "gen_code", "load_lib", "solve_method"}) if (!s[k]) return false;
    return s["SQP_settings"]["alpha"].IsDefined() && s["SQP_settings"]["step_num"].IsDefined();
  }
  struct PathRow { Path expr; std::vector<double> lo, hi; };
  struct LinkRow { Link expr; std::vector<double> lo, hi; };
  std::vector<LinkRow> links_;
  int batch_ = 1, maxIter_ = 1000, stepNum_ = 10, nx_ = 0, nu_ = 0, nh_ = 0, nk_ = 0;
  double alpha_ = 0.1; bool warmStart_ = true, verbose_ = true, genCode_ = false, loadLib_ = false, firstTime_ = true;
  SolverType solverType_ = SolverType::CUDA_SQP;
  std::vector<std::pair<Var, Dynamics>> dynamics_; std::vector<PathRow> paths_; std::vector<ocp_expr::Cost> costs_; std::vector<StageCost> stageCosts_;
  std::vector<std::string> constraintNames_;
  std::vector<double> optimalTrajectory_;
  std::unique_ptr<StageSQP> solver_;
};
