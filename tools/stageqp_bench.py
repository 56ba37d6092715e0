#!/usr/bin/env python3
"""Structured stage form on the headline size: what the gather kernel (csrc/stageqp.hip) costs next to the solve, and its rate against the
HBM roofline.  usage (GPU box): python tools/stageqp_bench.py [workload] [batch] [horizon]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                     # noqa: E402
from optimal_control_problem_amd import models                  # noqa: E402
from optimal_control_problem_amd.batch_qp import BatchQP         # noqa: E402
from optimal_control_problem_amd.stage_qp import StageQP, blocks_from_dense   # noqa: E402
from tests.support import stage_blocks as sb                    # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "quadrotor"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
N = int(sys.argv[3]) if len(sys.argv) > 3 else 20
mdl, ls1, _ = models.make_workload(name, 64, N=N)
nx, nu = mdl.nx, mdl.nu
cm, dm = sb.masks_of(ls1, N, nx, nu, nx)
mdl, ls, _ = models.make_workload(name, B, N=N)
dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(np.array(np.broadcast_to(a, (B,) + tuple(np.shape(a)[1:])), dtype=np.float64, order="C")).to(dev)
# blocks of the whole batch (host side, once): from the dense matrices of 64-instance slices
H = np.zeros((B, N, nx + nu, nx + nu)); Hp = np.zeros((B, N, nx, nx + nu)); Hpp = np.zeros((B, nx, nx)); AB = np.zeros((B, N - 1, nx, nx + nu))
for b0 in range(0, B, 64):
    sl = models.LocalSystem(ls.n, ls.m, ls.Pp, ls.Pi, ls.Ap, ls.Ai, np.broadcast_to(ls.P, (B, len(ls.Pi)))[b0:b0 + 64], np.broadcast_to(ls.q, (B, ls.n))[b0:b0 + 64],
                            np.broadcast_to(ls.A, (B, len(ls.Ai)))[b0:b0 + 64], np.broadcast_to(ls.l, (B, ls.m))[b0:b0 + 64], np.broadcast_to(ls.u, (B, ls.m))[b0:b0 + 64])
    Pd, Ad = sb.dense_batch(sl)
    H[b0:b0 + 64], Hp[b0:b0 + 64], Hpp[b0:b0 + 64], AB[b0:b0 + 64] = blocks_from_dense(Pd, Ad, N, nx, nu, nx)
dH, dHp, dHpp, dAB, dq, dl, du = (t(a) for a in (H, Hp, Hpp, AB, ls.q, ls.l, ls.u))
dP, dA = t(ls.P), t(ls.A)
sq = StageQP(N, nx, nu, B, np_=nx, cost_mask=cm, dyn_mask=dm)
qp = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
st = torch.cuda.current_stream(dev).cuda_stream
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]


def run(f, reps=10):
    f(); torch.cuda.synchronize(dev)
    best = 1e9
    for _ in range(reps):
        ev[0].record(); f(); ev[1].record(); torch.cuda.synchronize(dev)
        best = min(best, ev[0].elapsed_time(ev[1]))
    return best


t_csc = run(lambda: (qp.update(dP, dq, dA, dl, du), qp.solve(stream=st)))
t_blk = run(lambda: (sq.update_blocks(dH, dHp, dHpp, dAB, dq, dl, du, stream=st), sq.solve(stream=st)))
t_pack = run(lambda: sq.update_blocks(dH, dHp, dHpp, dAB, dq, dl, du, stream=st), reps=30)
a, b = qp.get(), sq.get()
same = all(np.array_equal(a[k], b[k], equal_nan=True) for k in ("x", "y", "status", "iters"))
bytes_out = 8 * (len(ls.Pi) + len(ls.Ai)); bytes_in = bytes_out - 8 * (ls.n + (N - 1) * nx)      # every value but the constants (identity rows, the +1 of s_{k+1}) is one block entry read
print("%s N=%d x %d: CSC update + solve %.3f ms, block update + solve %.3f ms (results identical: %s); gather kernel alone %.3f ms = %.0f GB/s of %d read + %d written "
      "algorithmic bytes per QP (HBM roofline 8000 GB/s: %.1f %%)" % (name, N, B, t_csc, t_blk, same, t_pack, B * (bytes_in + bytes_out) / t_pack / 1e6, bytes_in, bytes_out,
                                                                 100 * B * (bytes_in + bytes_out) / t_pack / 1e6 / 8000))
