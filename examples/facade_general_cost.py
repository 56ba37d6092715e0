"""The reference-shaped host API (OCPConfig YAML + OptimalControlProblem builders, reference readme.md:104-142) with what the
device evaluator adds: user dynamics, a general stage cost with a terminal cost, and a terminal constraint -- all traced from
NumPy callables and compiled for gfx950 (solver_settings.gen_code: true); the SQP tick never leaves the GPU.  Needs an MI355X."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import yaml  # noqa: E402

from optimal_control_problem_amd import models  # noqa: E402
from optimal_control_problem_amd.ocp import Dynamics, OptimalControlProblem, Path, StageCost  # noqa: E402

CONFIG = """
optimal_control_problem:
  discretization_settings: {dt: 0.05, horizon: 20}
  solver_settings: {verbose: false, gen_code: true, load_lib: false, max_iter: 1000, warm_start: true, solve_method: CUDA_SQP,
                    SQP_settings: {alpha: 0.8, step_num: 5}}
  OCP_variables:
    - {name: state, size: 4, lower_bound: [-2.4, -.inf, -.inf, -.inf], upper_bound: [2.4, .inf, .inf, .inf]}
    - {name: input, size: 1, lower_bound: [-20.0], upper_bound: [20.0]}
"""
plant = models.CartPole(20, 0.05)          # only its discrete map F(s, u) is used: any NumPy callable works


def stage_cost(s, u, r):
    e = s - r
    return e[..., 0] ** 2 + 10.0 * (1.0 - np.cos(e[..., 1])) + 0.1 * e[..., 2] ** 2 + 0.1 * e[..., 3] ** 2 + 0.01 * u[..., 0] ** 2


def terminal_cost(s, u, r):
    e = s - r
    return 20.0 * e[..., 0] ** 2 + 100.0 * e[..., 1] ** 2 + 2.0 * e[..., 2] ** 2 + 2.0 * e[..., 3] ** 2 + 0.01 * u[..., 0] ** 2


def cart_state(s, u):
    return np.stack([s[..., 0], s[..., 2]], axis=-1)


class BalanceOCP(OptimalControlProblem):
    def deployConstraintsAndAddCost(self):
        cfg = self.OCPConfigPtr_; ref = self.setReference(4); N = cfg.getHorizon()
        dynamics = plant.F
        for k in range(N):
            st, inp = cfg.getVariable(k, "state"), cfg.getVariable(k, "input")
            self.addScalarCost(StageCost(stage_cost if k < N - 1 else terminal_cost, st, inp, ref))
            last = k == N - 1                  # terminal constraint: the cart ends near the origin, almost at rest
            self.addInequalityConstraint("terminal", [-0.2, -0.5] if last else [-np.inf] * 2, Path(cart_state, st, inp, 2), [0.2, 0.5] if last else [np.inf] * 2)
            if k < N - 1:
                self.addEquationConstraint("dynamics", cfg.getVariable(k + 1, "state"), Dynamics(dynamics, st, inp))


batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ocp = BalanceOCP(yaml.safe_load(CONFIG)["optimal_control_problem"], batch=batch)
ocp.deployConstraintsAndAddCost()
ocp.genSolver()
rng = np.random.default_rng(0)
frame = np.zeros((batch, 5)); frame[:, 0] = rng.uniform(-0.5, 0.5, batch); frame[:, 1] = rng.normal(0, 0.1, batch)
traj = ocp.computeOptimalTrajectory(frame, np.zeros((batch, 4))).reshape(batch, 20, 5)
print("batch %d, horizon 20: dynamics violation %.2e, terminal |x| <= %.3f, |v| <= %.3f, first input of instance 0: %.3f"
      % (batch, np.abs(ocp.model_.constraints(traj.reshape(batch, -1))).max(), np.abs(traj[:, -1, 0]).max(), np.abs(traj[:, -1, 2]).max(), traj[0, 1, 4]))
