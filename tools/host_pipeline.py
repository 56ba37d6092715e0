#!/usr/bin/env python3
"""mpcqp_solve_host on the headline batch (pinned host buffers in, x / y / status / iters out), a few repetitions per chunk count: wall time per
step.  Under `rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python3 tools/host_pipeline.py 8` the trace shows how the
copies and the slice kernels overlap.   usage: python tools/host_pipeline.py [chunks ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from optimal_control_problem_amd import models
from optimal_control_problem_amd.batch_qp import BatchQP

chunk_list = [int(a) for a in sys.argv[1:]] or [4, 6, 8, 12, 16]
B = 8192
mdl, ls, _ = models.make_workload("quadrotor", B)
pin = [torch.from_numpy(np.ascontiguousarray(np.broadcast_to(a, (B,) + a.shape[1:]))).pin_memory().numpy() for a in (ls.P, ls.q, ls.A, ls.l, ls.u)]
qp = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
qp.set_dispatch_hint(False)
for ch in chunk_list:
    res = qp.solve_host(*pin, chunks=ch)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); res = qp.solve_host(*pin, chunks=ch, out=res); ts.append(time.perf_counter() - t0)
    print("chunks %2d: %.2f ms per step (best of 5: %.2f) = %.0f QP/s; solved %d" % (ch, 1e3 * float(np.median(ts)), 1e3 * min(ts), B / min(ts), int((res["status"] == 1).sum())), flush=True)
