"""Host-API facade: OCPConfig + OptimalControlProblem with the reference's names, YAML keys and call order.

Mirrors reference include/optimal_control_problem/OCP_config/OCPConfig.h:37-85 + src/OCP_config/OCPConfig.cpp and
include/optimal_control_problem/OptimalControlProblem.h:13-107 + src/OptimalControlProblem.cpp for the CUDA_SQP solve
method (the IPOPT / qpOASES / MIXED arms are third-party NLP solvers and out of scope).

CasADi is not available here, so the symbolic SX expressions the reference's builders take are replaced by a tiny
expression layer that covers what a stage-structured OCP needs: variable slices of a frame (OCPConfig.getVariable),
the reference parameter vector, differences, and a discrete-dynamics call.  genSolver() recognises the resulting
structure (quadratic tracking cost with diagonal weights or a general traced stage cost, dynamics defects between
consecutive frames, per-frame path constraints) and builds the
batched local-system evaluator (models.StageOCP) that plays the role of the CasADi-generated localSystemFunction_
(reference src/sqp_solver/SQPOptimizationSolver.cpp:74-77).  One object may drive a batch of independent instances.
"""
import numpy as np

from . import models
from .sqp import DeviceSQPOptimizationSolver, SQPOptimizationSolver


# ---------------------------------------------------------------------------------------------------- expressions
class Expr:
    def __sub__(self, other):
        return Diff(self, other)


class Var(Expr):
    """slice [start, stop) of the decision vector X = horizon x frameSize (reference OCPConfig.cpp:29-46)"""

    def __init__(self, step, name, start, stop, offset):
        self.step, self.name, self.start, self.stop, self.offset = step, name, start, stop, offset

    @property
    def size(self):
        return self.stop - self.start


class Reference(Expr):
    def __init__(self, size):
        self.size = size


class Diff(Expr):
    def __init__(self, a, b):
        self.a, self.b = a, b
        self.size = a.size


class Dynamics(Expr):
    """F(state, input): discrete map given as a NumPy callable on [..., nx], [..., nu] (complex-step differentiable)"""

    def __init__(self, F, state, inp):
        self.F, self.state, self.inp = F, state, inp
        self.size = state.size


class Path(Expr):
    """h(state, input) of one frame: a per-stage path-constraint expression given as a NumPy callable on [..., nx], [..., nu]
    returning [..., size]; used with addInequalityConstraint(name, lower, Path(...), upper)"""

    def __init__(self, h, state, inp, size):
        self.h, self.state, self.inp, self.size = h, state, inp, int(size)


class Link(Expr):
    """k(state_k, input_k, state_{k+1}, input_{k+1}) of two consecutive frames: a constraint expression that couples them beyond the
    dynamics -- rate limits u_{k+1} - u_k, slew limits on a state -- given as a NumPy callable on [..., nx], [..., nu], [..., nx],
    [..., nu] returning [..., size]; used with addInequalityConstraint(name, lower, Link(...), upper) on every stage k = 0 .. N-2.
    Stands for the SX expressions over several frames the reference accepts (src/OptimalControlProblem.cpp:448-489)."""

    def __init__(self, k, state, inp, state_next, inp_next, size):
        self.k, self.state, self.inp, self.state_next, self.inp_next, self.size = k, state, inp, state_next, inp_next, int(size)


class StageCost(Expr):
    """l(state, input, reference) of one frame: a general scalar cost term given as a NumPy callable on [..., nx], [..., nu],
    [..., nx] returning [...]; used with addScalarCost(StageCost(...)) on every frame (the last frame may use another function:
    a terminal cost).  Stands for the arbitrary SX terms the reference sums (src/OptimalControlProblem.cpp:491-497)."""

    def __init__(self, l, state, inp, reference):
        self.l, self.state, self.inp, self.reference = l, state, inp, reference
        self.size = 1


class General(Expr):
    """fn(X, p) over the WHOLE decision vector X (horizon x frameSize, reference OCPConfig.cpp:29-46) and the parameter vector p: any
    expression the stage pattern does not cover -- terms coupling frames that are not neighbours, different functions per frame, terms
    in the parameters.  A NumPy callable on 1-D arrays returning `size` values (one for addScalarCost); it is traced, so it must be
    straight-line code, like a CasADi SX expression (reference src/OptimalControlProblem.cpp:444-497 takes any SX)."""

    def __init__(self, fn, size=1):
        self.fn, self.size = fn, int(size)


def evaluate_expression(e, X, p):
    """value of a facade expression at decision vector X and parameters p (NumPy arrays or tracers): what the reference's SX graph of the
    same expression evaluates to"""
    if isinstance(e, tuple) and e[0] == "weighted_square":          # addVectorCost: sum_i param_i * cost_i^2 (reference :574-600)
        v = evaluate_expression(e[2], X, p)
        return sum(float(w) * (v[i] * v[i]) for i, w in enumerate(e[1]))
    if isinstance(e, Var):
        return X[e.start:e.stop]
    if isinstance(e, Reference):
        return p
    if isinstance(e, Diff):
        return evaluate_expression(e.a, X, p) - evaluate_expression(e.b, X, p)
    if isinstance(e, Dynamics):
        return e.F(evaluate_expression(e.state, X, p), evaluate_expression(e.inp, X, p))
    if isinstance(e, Path):
        return e.h(evaluate_expression(e.state, X, p), evaluate_expression(e.inp, X, p))
    if isinstance(e, Link):
        return e.k(evaluate_expression(e.state, X, p), evaluate_expression(e.inp, X, p), evaluate_expression(e.state_next, X, p), evaluate_expression(e.inp_next, X, p))
    if isinstance(e, StageCost):
        return e.l(evaluate_expression(e.state, X, p), evaluate_expression(e.inp, X, p), evaluate_expression(e.reference, X, p))
    if isinstance(e, General):
        return e.fn(X, p)
    raise TypeError("not a facade expression: %r" % (e,))


# ---------------------------------------------------------------------------------------------------- OCPConfig
def _bound_value(v):
    """'.inf' / '-.inf' strings as the reference handles them (OCPConfig.cpp:152-159); PyYAML already yields floats"""
    if isinstance(v, str):
        s = v.strip()
        if s in (".inf", ".Inf", ".INF"):
            return float("inf")
        if s in ("-.inf", "-.Inf", "-.INF"):
            return float("-inf")
        return float(s)
    return float(v)


class OCPConfig:
    def __init__(self, configNode):
        self.dt_ = float(configNode["discretization_settings"]["dt"])              # OCPConfig.cpp:90
        self.horizon_ = int(configNode["discretization_settings"]["horizon"])      # :92
        self.verbose_ = bool(configNode["solver_settings"]["verbose"])             # :94
        if "OCP_variables" not in configNode or configNode["OCP_variables"] is None:
            raise ValueError("node [OCP_variables] not found in YAML file")         # :113-116
        frame = configNode["OCP_variables"]
        if not isinstance(frame, (list, tuple)):
            raise ValueError("status_frame should be a sequence")                  # :120-123
        self.fields, self.fieldOffsets, total = [], {}, 0
        lower, upper = [], []
        for var in frame:                                                           # initializeFrame, :56-81
            if "name" not in var:
                raise ValueError("Field name not found in frame")
            if "size" not in var:
                raise ValueError("Field size not found in frame")
            name, size = str(var["name"]), int(var["size"])
            if size <= 0:
                raise ValueError("Field size must be positive: " + name)
            self.fields.append((name, size)); self.fieldOffsets[name] = total; total += size
            for key, dst in (("lower_bound", lower), ("upper_bound", upper)):
                if key not in var:
                    raise ValueError("Missing %s for variable: %s" % (key, name))   # :138-141,180-183
                vals = np.zeros(size)
                seq = var[key]
                if isinstance(seq, (list, tuple)):
                    for i in range(min(len(seq), size)):                            # size mismatch only warns, :147-151
                        vals[i] = _bound_value(seq[i])
                dst.append(vals)
        self.totalSize = total
        one_lo, one_hi = np.concatenate(lower), np.concatenate(upper)
        self.lowerBounds_ = [one_lo.copy() for _ in range(self.horizon_)]           # coverLowerBounds: one frame x horizon
        self.upperBounds_ = [one_hi.copy() for _ in range(self.horizon_)]
        self.initialGuess_ = None

    def getVariable(self, stepID, variableName):
        if stepID < 0 or stepID >= self.horizon_:
            raise IndexError("Frame ID out of range")                               # :30-32
        if variableName not in self.fieldOffsets:
            raise ValueError("Field name not found in frame")                       # :33-36
        size = dict(self.fields)[variableName]
        start = stepID * self.totalSize + self.fieldOffsets[variableName]
        return Var(stepID, variableName, start, start + size, self.fieldOffsets[variableName])

    def getVariables(self):
        return self.horizon_ * self.totalSize

    def getLowerBounds(self):
        return self.lowerBounds_

    def getUpperBounds(self):
        return self.upperBounds_

    def getHorizon(self):
        return self.horizon_

    def getDt(self):
        return self.dt_

    def getFrameSize(self):
        return self.totalSize

    def setInitialGuess(self, initialGuess):
        self.initialGuess_ = np.asarray(initialGuess, float)

    def getInitialGuess(self):
        return self.initialGuess_


def _as_scalar(v):
    """a cost term must be one value (a vector-valued expression of size 1 counts)"""
    if hasattr(v, "items") and not isinstance(v, dict):
        if len(v.items) != 1:
            raise ValueError("a scalar cost term evaluated to %d values" % len(v.items))
        return v.items[0]
    if isinstance(v, np.ndarray):
        if v.size != 1:
            raise ValueError("a scalar cost term evaluated to %d values" % v.size)
        return v.reshape(-1)[0]
    return v


# ---------------------------------------------------------------------------------------------------- OptimalControlProblem
class _FacadeStageOCP(models.StageOCP):
    name = "facade_ocp"

    def __init__(self, nx, nu, N, dt, Q, R, F, lo, hi, h=None, nh=0, h_lo=None, h_hi=None, lcost=None, lterm=None, k=None, nk=0, k_lo=None, k_hi=None):
        self.nx, self.nu, self._F, self._lo, self._hi = nx, nu, F, lo, hi
        self._h, self.nh, self.h_lo, self.h_hi = h, int(nh), h_lo, h_hi
        self._k, self.nk, self.k_lo, self.k_hi = k, int(nk), k_lo, k_hi
        self.lcost, self.lterm = lcost, lterm
        super().__init__(N, dt, Q, R)

    def F(self, s, u):
        return self._F(s, u)

    def hfun(self, s, u):
        return self._h(s, u)

    def kfun(self, s, u, sn, un):
        return self._k(s, u, sn, un)

    def frame_bounds(self):
        return self._lo, self._hi


class OptimalControlProblem:
    """Abstract base: subclasses implement deployConstraintsAndAddCost() (reference OptimalControlProblem.h:101)."""

    SOLVER_TYPES = ("IPOPT", "SQP", "CUDA_SQP", "MIXED")

    def __init__(self, configNode, batch=1, qp_solver=None, device_resident=None):
        """device_resident: run the whole SQP tick on the GPU -- the dynamics are traced, emitted as code and compiled for
        gfx950 at genSolver() (codegen.py).  Default: the YAML's solver_settings.gen_code, the flag with which the reference
        generates and compiles its C code at the same point (reference src/OptimalControlProblem.cpp:263-287)."""
        if not self.validateConfig(configNode):
            raise RuntimeError("Invalid configuration file")                        # OptimalControlProblem.cpp:16-18
        self.OCPConfigPtr_ = OCPConfig(configNode)
        s = configNode["solver_settings"]
        self.solverSettings = dict(maxIter=int(s["max_iter"]), warmStart=bool(s["warm_start"]),
                                   alpha=float(s["SQP_settings"]["alpha"]), stepNum=int(s["SQP_settings"]["step_num"]),
                                   verbose=bool(s["verbose"]), genCode=bool(s["gen_code"]), loadLib=bool(s["load_lib"]))
        method = str(s["solve_method"])
        if method not in self.SOLVER_TYPES:
            raise ValueError("Unknown solver type: " + method)                       # :43-45
        self.solverType = method
        self.batch = int(batch)
        self._qp_solver = qp_solver
        self.deviceResident = self.solverSettings["genCode"] if device_resident is None else bool(device_resident)
        self.constraints_, self.constraintNames_ = [], []
        self.constraintLowerBounds_, self.constraintUpperBounds_ = [], []
        self.costs_ = []
        self.reference_ = None
        self.firstTime_ = True
        self.optimalTrajectory_ = None
        self.OSQPSolverPtr_ = None
        self.model_ = None

    @staticmethod
    def validateConfig(config):                                                      # :54-62
        try:
            s = config["solver_settings"]
            return all(k in s for k in ("max_iter", "warm_start", "SQP_settings", "verbose", "gen_code", "load_lib", "solve_method")) \
                and all(k in s["SQP_settings"] for k in ("alpha", "step_num"))
        except (KeyError, TypeError):
            return False

    # -- builders (:444-497,574-600)
    def setReference(self, size):
        self.reference_ = Reference(int(size))
        return self.reference_

    def getReference(self):
        return self.reference_

    def addScalarCost(self, cost):
        self.costs_.append(cost)

    def addVectorCost(self, param, cost):
        param = np.asarray(param, float).ravel()
        if param.shape[0] != cost.size:
            print("损失的符号向量和参数向量维度不一致")                                    # :576-579 (prints and returns)
            return
        self.addScalarCost(("weighted_square", param, cost))

    def addInequalityConstraint(self, constraintName, lowerBound, expression, upperBound):
        lowerBound = np.asarray(lowerBound, float).ravel(); upperBound = np.asarray(upperBound, float).ravel()
        if lowerBound.shape[0] != expression.size or upperBound.shape[0] != expression.size:
            raise ValueError("SX used for inequality constraints has different dimensions!")   # :452-454
        self.constraints_.append(expression); self.constraintNames_ += [constraintName] * expression.size
        self.constraintLowerBounds_.append(lowerBound); self.constraintUpperBounds_.append(upperBound)

    def addEquationConstraint(self, constraintName, leftSX, rightSX=None):
        if rightSX is not None and leftSX.size != rightSX.size:
            raise ValueError("SX used for constraints has different dimension!")     # :472-474
        expr = leftSX if rightSX is None else leftSX - rightSX
        self.constraints_.append(expr); self.constraintNames_ += [constraintName] * expr.size
        self.constraintLowerBounds_.append(np.zeros(expr.size)); self.constraintUpperBounds_.append(np.zeros(expr.size))

    def getCostFunction(self):
        """the list of cost terms (the reference sums SX terms into totalCost_, OptimalControlProblem.cpp:491-497)"""
        self.totalCost_ = list(self.costs_)
        return self.totalCost_

    def setSolverType(self, solverType):                                             # :499-505
        name = solverType.name if hasattr(solverType, "name") else str(solverType)
        if name not in self.SOLVER_TYPES:
            raise ValueError("Unknown solver type: " + name)
        self.solverType = name

    def getSolverType(self):
        return self.solverType

    def genCode(self):
        """generate + compile the local-system evaluation for the GPU (the reference's genCode writes C for the NLP solver and
        shells out to gcc, OptimalControlProblem.cpp:263-287,602-640); returns the path of the shared library"""
        from . import codegen
        if getattr(self, "generalPath_", False):
            raise NotImplementedError("gen_code compiles the stage pattern for the device; this problem takes the general path (%s)" % self.generalPathReason_)
        model = self.model_ if self.model_ is not None else self._compile_stage_model()
        h_lo, h_hi = model.path_bounds() if model.nh else (None, None)
        tape = codegen.trace(model.F, model.nx, model.nu, model.hfun if model.nh else None, model.nh, h_lo[0] if model.nh else None, h_hi[0] if model.nh else None,
                             lcost=model.lcost if model.general_cost else None, lterm=model.lterm if model.general_cost else None)
        return codegen.build_device_library(tape)

    def getConstraints(self):
        return self.constraints_

    def getConstraintLowerBounds(self):
        return self.constraintLowerBounds_

    def getConstraintUpperBounds(self):
        return self.constraintUpperBounds_

    def deployConstraintsAndAddCost(self):
        raise NotImplementedError("pure virtual in the reference (OptimalControlProblem.h:101)")

    # -- genSolver (:224-442), CUDA_SQP arm :391-401
    def genSolver(self):
        cfg = self.OCPConfigPtr_
        if cfg.getVariables() == 0:
            raise RuntimeError("Status or input variables are empty")
        if not self.constraints_:
            raise RuntimeError("Constraints are empty")                              # :231-233
        if self.solverType != "CUDA_SQP":
            raise NotImplementedError("solve_method %s relies on third-party NLP solvers (IPOPT / qpOASES) and is out of scope; "
                                      "use CUDA_SQP" % self.solverType)
        # the stage pattern compiles to the batched evaluator (host or device); anything else -- terms coupling frames that are not
        # neighbours, different functions per frame, arbitrary expressions over the whole decision vector -- takes the general path: f and g
        # traced and differentiated over the whole vector on the host (general_nlp.GeneralNLP), the QPs solved on the GPU all the same.
        # The reference handles every problem that way (src/OptimalControlProblem.cpp:235-240, SQPOptimizationSolver.cpp:47-77).
        self.generalPath_ = False
        try:
            self.model_ = self._compile_stage_model()
        except NotImplementedError as why:
            self.model_ = self._compile_general_model()
            self.generalPath_, self.generalPathReason_ = True, str(why)
            if self.solverSettings["verbose"]:
                print("OptimalControlProblem: not a stage pattern (%s); general host evaluation, QPs on the GPU" % why)
        options = {"max_iter": self.solverSettings["stepNum"], "alpha": self.solverSettings["alpha"],
                   "verbose": self.solverSettings["verbose"]}
        if self.deviceResident and self._qp_solver is None and not self.generalPath_:
            self.OSQPSolverPtr_ = DeviceSQPOptimizationSolver(self.model_, options, batch=self.batch)
        else:
            self.OSQPSolverPtr_ = SQPOptimizationSolver(self.model_, options, batch=self.batch, qp_solver=self._qp_solver)

    def _compile_general_model(self):
        """f = sum of the cost terms, g = the constraints in the order they were added, over w = [reference; X] (the reference's
        {x, f, g, p}, src/OptimalControlProblem.cpp:235-240)"""
        from .general_nlp import GeneralNLP
        cfg = self.OCPConfigPtr_
        nvar = cfg.getVariables(); npar = self.reference_.size if self.reference_ is not None else 0
        if not self.costs_:
            raise RuntimeError("Cost function is empty")
        cost = lambda w: sum(_as_scalar(evaluate_expression(c, w[npar:], w[:npar])) for c in self.costs_)
        cons = lambda w: [evaluate_expression(c, w[npar:], w[:npar]) for c in self.constraints_]
        model = GeneralNLP(nvar, npar, cost, cons)
        if model.ng != sum(len(b) for b in self.constraintLowerBounds_):
            raise ValueError("SX used for constraints has different dimension!")
        self._row_order = list(range(len(self.constraints_)))
        return model

    def _compile_stage_model(self):
        cfg = self.OCPConfigPtr_
        N, f = cfg.getHorizon(), cfg.getFrameSize()
        dyn_idx = [i for i, c in enumerate(self.constraints_) if isinstance(c, Diff) and isinstance(c.b, Dynamics)]
        path_idx = [i for i, c in enumerate(self.constraints_) if isinstance(c, Path)]
        link_idx = [i for i, c in enumerate(self.constraints_) if isinstance(c, Link)]
        dyn = [self.constraints_[i] for i in dyn_idx]; path = [self.constraints_[i] for i in path_idx]; link = [self.constraints_[i] for i in link_idx]
        if len(dyn) + len(path) + len(link) != len(self.constraints_) or len(dyn) != N - 1 or len(path) not in (0, N) or len(link) not in (0, N - 1):
            raise NotImplementedError("this facade compiles dynamics defects x_{k+1} - F(x_k, u_k) between consecutive frames, one per-frame path "
                                      "constraint Path(h, state_k, input_k) on every frame and one link constraint Link(k, frame_k, frame_{k+1}) on every stage")
        s0, u0, F = dyn[0].b.state, dyn[0].b.inp, dyn[0].b.F
        nx, nu = s0.size, u0.size
        if s0.offset != 0 or u0.offset != nx or nx + nu != f:
            raise NotImplementedError("frame layout must be [state; input]")
        for k, c in enumerate(dyn):
            if not (isinstance(c.a, Var) and c.a.step == k + 1 and c.a.name == s0.name and c.b.state.step == k and c.b.inp.step == k and c.b.F == F):
                raise NotImplementedError("dynamics constraints must link frame k to frame k + 1 in order")
        Qk = np.zeros((N, nx)); Rk = np.zeros((N, nu))        # per-step weights: terminal costs and ramps are ordinary here
        seenQ, seenR = set(), set()
        lcost = lterm = None
        general = [c for c in self.costs_ if isinstance(c, StageCost)]
        if general:
            # general stage cost: one StageCost per frame, the same function on every frame but (optionally) the last
            if len(general) != len(self.costs_) or sorted(c.state.step for c in general) != list(range(N)):
                raise NotImplementedError("general costs: exactly one StageCost term per frame and no other cost terms")
            general.sort(key=lambda c: c.state.step)
            for k, c in enumerate(general):
                if not (c.state.name == s0.name and c.inp.name == u0.name and c.inp.step == k and c.reference is self.reference_):
                    raise NotImplementedError("a StageCost takes the state, the input and the reference of its own frame")
            lcost = general[0].l
            if any(c.l != lcost for c in general[:-1]):
                raise NotImplementedError("the stage cost must be the same function on every frame except the last")
            lterm = general[-1].l if general[-1].l != lcost else None
            seenQ = seenR = set(range(N))
        for term in ([] if general else self.costs_):
            kind, w, e = term if isinstance(term, tuple) else (None, None, None)
            if kind != "weighted_square":
                raise NotImplementedError("only addVectorCost terms and StageCost terms are compiled")
            if isinstance(e, Diff) and isinstance(e.a, Var) and isinstance(e.b, Reference) and e.a.name == s0.name:
                Qk[e.a.step] += w; seenQ.add(e.a.step)        # repeated terms on one step add up, like the SX sum (:491-497)
            elif isinstance(e, Var) and e.name == u0.name:
                Rk[e.step] += w; seenR.add(e.step)
            else:
                raise NotImplementedError("cost term not recognised")
        same = (Qk == Qk[0]).all() and (Rk == Rk[0]).all()
        Q, R = (Qk[0], Rk[0]) if same else (Qk, Rk)
        if seenQ != set(range(N)) or seenR != set(range(N)):
            raise NotImplementedError("tracking and input costs must be added for every step")
        if self.reference_ is None or self.reference_.size != nx:
            raise NotImplementedError("reference must have the state's dimension")
        h = None; nh = 0; h_lo = h_hi = None
        if path:
            path.sort(key=lambda c: c.state.step); path_idx.sort(key=lambda i: self.constraints_[i].state.step)
            h, nh = path[0].h, path[0].size
            for k, (c, i) in enumerate(zip(path, path_idx)):
                if not (c.h == h and c.size == nh and c.state.step == k and c.inp.step == k and c.state.name == s0.name and c.inp.name == u0.name):
                    raise NotImplementedError("the path constraint must be the same function on every frame (its bounds may differ by frame)")
            h_lo = np.stack([self.constraintLowerBounds_[i] for i in path_idx]); h_hi = np.stack([self.constraintUpperBounds_[i] for i in path_idx])
            if (h_lo == h_lo[0]).all() and (h_hi == h_hi[0]).all():
                h_lo, h_hi = h_lo[0], h_hi[0]
        kf = None; nk = 0; k_lo = k_hi = None
        if link:
            link_idx.sort(key=lambda i: self.constraints_[i].state.step); link = [self.constraints_[i] for i in link_idx]
            kf, nk = link[0].k, link[0].size
            for k, c in enumerate(link):
                if not (c.k == kf and c.size == nk and c.state.step == k and c.inp.step == k and c.state_next.step == k + 1 and c.inp_next.step == k + 1
                        and c.state.name == s0.name and c.inp.name == u0.name and c.state_next.name == s0.name and c.inp_next.name == u0.name):
                    raise NotImplementedError("the link constraint must be the same function of (frame k, frame k + 1) on every stage")
            k_lo = self.constraintLowerBounds_[link_idx[0]]; k_hi = self.constraintUpperBounds_[link_idx[0]]
            if any((self.constraintLowerBounds_[i] != k_lo).any() or (self.constraintUpperBounds_[i] != k_hi).any() for i in link_idx):
                raise NotImplementedError("the link constraint's bounds must be the same on every stage")
        # rows of the compiled model: dynamics rows in frame order, then the path rows in frame order, then the link rows in stage order
        self._row_order = [i for _, i in sorted((self.constraints_[i].a.step, i) for i in dyn_idx)] + path_idx + link_idx
        return _FacadeStageOCP(nx, nu, N, cfg.getDt(), Q, R, F, cfg.getLowerBounds()[0], cfg.getUpperBounds()[0], h, nh, h_lo, h_hi, lcost, lterm, kf, nk, k_lo, k_hi)

    # -- computeOptimalTrajectory (:78-222), CUDA_SQP arm
    def computeOptimalTrajectory(self, frame, reference):
        cfg = self.OCPConfigPtr_
        frame = np.asarray(frame, float).reshape(self.batch, -1)
        reference = np.asarray(reference, float).reshape(self.batch, -1)
        if frame.shape[1] != cfg.getFrameSize():
            raise ValueError("State dimension mismatch: received %d, expected %d" % (frame.shape[1], cfg.getFrameSize()))      # :79-84
        rsize = self.reference_.size if self.reference_ is not None else 0           # (no setReference(): an empty parameter vector, like an empty SX)
        if reference.shape[1] != rsize:
            raise ValueError("Reference dimension mismatch: received %d, expected %d" % (reference.shape[1], rsize))  # :85-90
        lbx = np.tile(np.concatenate(cfg.getLowerBounds()), (self.batch, 1))
        ubx = np.tile(np.concatenate(cfg.getUpperBounds()), (self.batch, 1))
        fs = cfg.getFrameSize()
        lbx[:, :fs] = frame; ubx[:, :fs] = frame                                       # :95-96 the whole first frame is pinned
        order = getattr(self, "_row_order", range(len(self.constraintLowerBounds_)))
        lbg = np.tile(np.concatenate([self.constraintLowerBounds_[i] for i in order]), (self.batch, 1))
        ubg = np.tile(np.concatenate([self.constraintUpperBounds_[i] for i in order]), (self.batch, 1))
        x0 = np.zeros((self.batch, cfg.getVariables())) if self.firstTime_ or self.optimalTrajectory_ is None else self.optimalTrajectory_
        arg = dict(lbx=lbx, ubx=ubx, lbg=lbg, ubg=ubg, x0=x0, p=reference)
        if not self.solverInputCheck(arg):
            raise RuntimeError("Solver input validation failed")                        # :116-118
        try:
            res = self.OSQPSolverPtr_.getOptimalSolution(arg)                           # :141
        except Exception as e:                                                         # :219-221
            raise RuntimeError("Optimization failed: " + str(e))
        self.optimalTrajectory_ = res["x"]
        self.firstTime_ = False
        return self.optimalTrajectory_

    def solverInputCheck(self, arg):                                                  # :511-552
        ng = sum(len(b) for b in self.constraintLowerBounds_)
        nv = self.OCPConfigPtr_.getVariables()
        ok = arg["lbg"].shape[1] == ng and arg["ubg"].shape[1] == ng and arg["lbx"].shape[1] == nv and \
            arg["ubx"].shape[1] == nv and arg["x0"].shape[1] == nv and arg["p"].shape[1] == (self.reference_.size if self.reference_ is not None else 0)
        return bool(ok)

    def getOptimalTrajectory(self):
        return self.optimalTrajectory_
