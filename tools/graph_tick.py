#!/usr/bin/env python3
"""Small-batch latency of one device-resident SQP tick (eval -> update -> solve -> get -> step -> merit), launched eagerly
from Python vs captured once in a HIP graph (torch.cuda.CUDAGraph) and replayed.  The C ABI entry points used here issue
only stream-ordered work (kernel launches, D2D copies, event records), so the whole tick is capturable.
usage: python tools/graph_tick.py [workload] [horizon] [batches...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from optimal_control_problem_amd import models
from optimal_control_problem_amd.sqp import DeviceSQPOptimizationSolver

name = sys.argv[1] if len(sys.argv) > 1 else "double_integrator"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
batches = [int(v) for v in sys.argv[3:]] or [1, 16, 64, 256, 1024]
rows = []
for B in batches:
    mdl, ls, meta = models.make_workload(name, B, N=N)
    arg = {k: torch.as_tensor(meta[k], dtype=torch.float64, device="cuda") for k in ("lbx", "ubx", "lbg", "ubg", "p")}
    dev = DeviceSQPOptimizationSolver(mdl, {"max_iter": 1, "alpha": 1.0}, batch=B)
    x0 = torch.as_tensor(meta["x_iterate"], dtype=torch.float64, device="cuda")

    def run_eager(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            dev.x.copy_(x0)
            dev.getOptimalSolution(arg, to_host=False)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    run_eager(3)
    t_eager = run_eager(50)
    ref = dev.x.clone(); ref_it = dev.iters.clone()
    # capture one tick on a side stream (torch requires a non-default stream), replay it
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        dev.x.copy_(x0); dev.getOptimalSolution(arg, to_host=False)       # warm-up on the capture stream
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            dev.x.copy_(x0)
            dev.getOptimalSolution(arg, to_host=False)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    same = bool(torch.equal(dev.x, ref) and torch.equal(dev.iters, ref_it))
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / 200
    rows.append({"batch": B, "eager_us_per_tick": t_eager * 1e6, "graph_us_per_tick": t_graph * 1e6, "graph_equals_eager": same,
                 "mean_admm_iters": float(ref_it.double().mean())})
    dev.close()
print(json.dumps({"workload": "%s N=%d, one SQP tick (1 QP)" % (name, N), "rows": rows}))
