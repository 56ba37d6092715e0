#!/usr/bin/env python3
"""Closed-loop receding-horizon MPC on the device (BASELINE config 2 shape: double-integrator LQ-MPC, one QP per tick):
every tick pins the first frame to the measured state + the input being applied (reference
src/OptimalControlProblem.cpp:93-96), evaluates the local system on the GPU, solves the batch of QPs and advances the
plants with the first free input.  Compares cold starts (what the reference does: CuCaQP::setSystem clears the solver)
with the kept-workspace behaviour (ADMM warm start + carried rho).
usage: python tools/mpc_loop_bench.py [batch] [ticks]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from optimal_control_problem_amd import models
from optimal_control_problem_amd.sqp import DeviceSQPOptimizationSolver

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 60
mdl, ls, meta = models.make_workload("double_integrator", B)
nx, nu, f, N, h = mdl.nx, mdl.nu, mdl.f, mdl.N, mdl.dt
out = {"workload": "double_integrator N=%d batch=%d, %d closed-loop ticks (1 QP per tick)" % (N, B, T)}
for label, opt in (("_process_warmup", {}), ("cold", {}), ("warm", {"warm_start_admm": True}), ("warm+rho", {"warm_start_admm": True, "carry_rho": True}),
                   ("kept workspace", {"constant_matrices": True}), ("kept workspace+warm", {"constant_matrices": True, "warm_start_admm": True})):
    dev = DeviceSQPOptimizationSolver(mdl, dict({"max_iter": 1, "alpha": 1.0, "skip_failed_steps": True}, **opt), batch=B)
    arg = {k: torch.as_tensor(meta[k], dtype=torch.float64, device="cuda") for k in ("lbx", "ubx", "lbg", "ubg", "p")}
    state = torch.as_tensor(meta["frame0"][:, :nx], dtype=torch.float64, device="cuda").clone()
    u_now = torch.zeros((B, nu), dtype=torch.float64, device="cuda")
    failed = 0; iters = []; cost = torch.zeros(B, dtype=torch.float64, device="cuda")
    dev.getOptimalSolution(arg, to_host=False); dev.setInitialGuess(np.zeros(mdl.nvar))     # warm-up launch, then reset
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(T):
        arg["lbx"][:, :nx] = state; arg["ubx"][:, :nx] = state
        arg["lbx"][:, nx:f] = u_now; arg["ubx"][:, nx:f] = u_now
        res = dev.getOptimalSolution(arg, to_host=False)
        X = res["x"].view(B, N, f)
        # plant = the model's exact discrete map under the input being applied; then switch to the first free input
        pos = state[:, 0] + h * state[:, 1] + 0.5 * h * h * u_now[:, 0]
        vel = state[:, 1] + h * u_now[:, 0]
        cost += 10.0 * state[:, 0] ** 2 + state[:, 1] ** 2 + 0.1 * u_now[:, 0] ** 2
        state = torch.stack([pos, vel], dim=1)
        ok = (dev.status == 1).unsqueeze(1)      # hard state bounds + eps 1e-3 solutions: an occasional tick is infeasible -> hold the input
        u_now = torch.where(ok, X[:, 1, nx:], u_now)
        failed += int((~ok).sum())
        iters.append(dev.iters.clone())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    it = torch.stack(iters).double()
    out[label] = {"s": dt, "ticks_per_s": B * T / dt, "mean_admm_iters": float(it.mean()), "mean_admm_iters_first_last_tick": [float(it[0].mean()), float(it[-1].mean())],
                  "closed_loop_cost_mean": float(cost.mean()), "final_state_norm_max": float(state.abs().max()),
                  "infeasible_ticks": failed}
    dev.close()
out.pop("_process_warmup")      # the first loop in a process pays torch / HIP lazy initialisation
print(json.dumps(out))
