#!/usr/bin/env python3
"""A batch of linear time-varying MPC problems handed to the engine in stage blocks (the structured stage form of include/mpcqp.h):
    min  sum_k 1/2 [s_k; u_k]' H_k [s_k; u_k] + g_k' [s_k; u_k]   s.t.  s_{k+1} = A_k s_k + B_k u_k + c_k,  s_0 given,  |u| <= 1
No CSC arrays anywhere: the library derives the pattern from {N, nx, nu} and gathers the values on the device.   (needs an MI355X)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimal_control_problem_amd import StageQP          # noqa: E402

N, nx, nu, B = 30, 4, 2, 2048
f = nx + nu; n = N * f; m = n + (N - 1) * nx
rng = np.random.default_rng(0)
H = np.zeros((B, N, f, f)); H[:, :, np.arange(f), np.arange(f)] = np.r_[np.full(nx, 2.0), np.full(nu, 0.2)]           # tracking weights
AB = np.zeros((B, N - 1, nx, f))
AB[:, :, :, :nx] = np.eye(nx) + 0.05 * rng.normal(size=(B, N - 1, nx, nx))                                        # every instance, every stage its own A_k
AB[:, :, :, nx:] = 0.1 * rng.normal(size=(B, N - 1, nx, nu))
q = np.zeros((B, n))
l = np.full((B, m), -np.inf); u = np.full((B, m), np.inf)
s0 = rng.uniform(-1, 1, size=(B, nx))
l[:, :nx] = s0; u[:, :nx] = s0                                              # first state pinned (what computeOptimalTrajectory does with lbx = ubx)
for k in range(N):
    l[:, k * f + nx:(k + 1) * f] = -1.0; u[:, k * f + nx:(k + 1) * f] = 1.0  # input box
l[:, n:] = 0.0; u[:, n:] = 0.0                                              # s_{k+1} - A_k s_k - B_k u_k = 0

qp = StageQP(N, nx, nu, B)                                                  # np_ = 0: no parameter block
qp.update_blocks(H, None, None, AB, q, l, u)
qp.solve(); qp.sync()                                                     # (first launch: code load)
qp.update_blocks(H, None, None, AB, q, l, u)
qp.solve()
res = qp.get()
x = res["x"].reshape(B, N, f)
gap = x[:, 1:, :nx] - np.einsum("bkrc,bkc->bkr", AB, x[:, :-1])
print("n = %d, m = %d, kernel variant %d; %d of %d solved, mean %.1f ADMM iterations, %.2f ms for the batch; max dynamics residual %.1e, |s_N| mean %.3f (from %.3f)"
      % (qp.n, qp.m, qp.plan_info()["variant"], int((res["status"] == 1).sum()), B, res["iters"].mean(), qp.last_kernel_ms(), np.abs(gap).max(),
         np.abs(x[:, -1, :nx]).mean(), np.abs(s0).mean()))
qp.close()
