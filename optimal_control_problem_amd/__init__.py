"""optimal_control_problem_amd -- MI355X-native batched QP engine for the SQP-MPC hot path of
LockedFlysher/optimal_control_problem (src/sqp_solver), behind the C ABI of include/mpcqp.h.

Importing the package never touches the GPU; the first compute call loads libmpcqp.so and fails loudly if
the library or a gfx950 device is missing (there is no CPU fallback)."""
from . import models  # noqa: F401

__all__ = ["models", "BatchQP", "StageQP", "CuCaQP", "SQPOptimizationSolver", "DeviceSQPOptimizationSolver", "StageEvaluator",
           "OptimalControlProblem", "OCPConfig"]


def __getattr__(name):
    if name == "BatchQP":
        from .batch_qp import BatchQP
        return BatchQP
    if name == "StageQP":
        from .stage_qp import StageQP
        return StageQP
    if name == "CuCaQP":
        from .cucaqp import CuCaQP
        return CuCaQP
    if name in ("SQPOptimizationSolver", "DeviceSQPOptimizationSolver"):
        from . import sqp
        return getattr(sqp, name)
    if name == "StageEvaluator":
        from .stage_eval import StageEvaluator
        return StageEvaluator
    if name in ("OptimalControlProblem", "OCPConfig"):
        from . import ocp
        return getattr(ocp, name)
    raise AttributeError(name)
