"""ocp_module -- the Python module the reference meant to ship: its pybind11 file defines PYBIND11_MODULE(ocp_module, m)
with a SolverType enum and an OptimalControlProblem class whose methods are the snake_case names below (reference
src/pybind/python_bindings.cpp:409-446; the whole file is commented out and stale against the C++ class, e.g.
get_optimal_input_first_frame has no C++ counterpart any more).  Since the host side of this engine is Python already, the
"binding" is a naming layer over optimal_control_problem_amd.ocp: code written against the
intended module runs on the GPU engine.  The trampoline of the reference (PyOptimalControlProblem, :393-407) corresponds to
subclassing and overriding deploy_constraints_and_add_cost here."""
import enum

from . import ocp as _ocp
from .ocp import Dynamics, OCPConfig, Path, StageCost  # noqa: F401  (the expression layer standing in for casadi::SX)


class SolverType(enum.Enum):
    IPOPT = 0
    SQP = 1
    CUDA_SQP = 2
    MIXED = 3


IPOPT, SQP, CUDA_SQP, MIXED = SolverType.IPOPT, SolverType.SQP, SolverType.CUDA_SQP, SolverType.MIXED     # .export_values()


class OptimalControlProblem(_ocp.OptimalControlProblem):
    def deploy_constraints_and_add_cost(self):
        raise NotImplementedError("pure virtual (PYBIND11_OVERRIDE_PURE in the reference's trampoline)")

    def deployConstraintsAndAddCost(self):
        return self.deploy_constraints_and_add_cost()

    def set_solver_type(self, t):
        return self.setSolverType(t)

    def get_solver_type(self):
        return SolverType[self.getSolverType()]

    def get_reference(self):
        return self.getReference()

    def set_reference(self, size):
        return self.setReference(size)

    def get_optimal_trajectory(self):
        return self.getOptimalTrajectory()

    def gen_solver(self):
        return self.genSolver()

    def gen_code(self):
        return self.genCode()

    def compute_optimal_trajectory(self, frame, reference):
        return self.computeOptimalTrajectory(frame, reference)

    def add_scalar_cost(self, cost):
        return self.addScalarCost(cost)

    def add_vector_cost(self, param, cost):
        return self.addVectorCost(param, cost)

    def add_inequality_constraint(self, name, lower, expression, upper):
        return self.addInequalityConstraint(name, lower, expression, upper)

    def add_equation_constraint(self, name, left, right=None):
        return self.addEquationConstraint(name, left, right)

    def get_cost_function(self):
        return self.getCostFunction()

    def get_constraints(self):
        return self.getConstraints()

    def get_constraint_lower_bounds(self):
        return self.getConstraintLowerBounds()

    def get_constraint_upper_bounds(self):
        return self.getConstraintUpperBounds()

    def solver_input_check(self, arg):
        return self.solverInputCheck(arg)

    @property
    def total_cost_(self):
        return self.getCostFunction()
