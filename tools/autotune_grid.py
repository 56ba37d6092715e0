#!/usr/bin/env python3
"""Kernel family by measurement (mpcqp_create_tuned) against the families forced one by one, on the size grid of the variant-grid profiles.
For every size: the kernel time of each family on the workload's real data, the family the tuned create picked (from its synthetic run) and
how far its time is from the best column.   usage (GPU box): python tools/autotune_grid.py [batch] > profiles/rNN_autotune_grid.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from optimal_control_problem_amd import _lib, models
from optimal_control_problem_amd.batch_qp import BatchQP

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
GRID = [("double_integrator", N) for N in (10, 20, 30, 40, 60, 100)] + [("quadrotor", N) for N in (5, 6, 10, 15, 20, 25, 30, 50)] + [("cartpole", N) for N in (20, 30, 50, 70, 100)]
FAMILIES = ["res1", "res2", "res4", "gres2", "gres4", "oc4", "oc8"]
NAME = {1: "res1", 2: "res2", 4: "res4", 8: "res8", 102: "gres2", 104: "gres4", 204: "oc4", 208: "oc8", 0: "stream"}
dev = torch.device("cuda", 0)


def time_handle(qp, d):
    qp.set_dispatch_hint(False)
    ms = []
    for _ in range(3):
        qp.update(*d); qp.solve(); torch.cuda.synchronize(); ms.append(qp.last_kernel_ms())
    return min(ms[1:])


print("%-22s " % "size" + " ".join("%8s" % f for f in FAMILIES) + "   rule(ms)  tuned -> family (ms)   vs best", flush=True)
worst = 0.0
for name, N in GRID:
    mdl, ls, _ = models.make_workload(name, B, N=N)
    d = [torch.from_numpy(a).to(dev) for a in (ls.P, ls.q, ls.A, ls.l, ls.u)]
    col = {}
    for fam in FAMILIES:
        os.environ["MPCQP_VARIANT"] = fam
        try:
            qp = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
            col[fam] = time_handle(qp, d); qp.close()
        except _lib.MpcqpError:
            col[fam] = None
    del os.environ["MPCQP_VARIANT"]
    qp = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai); rule_ms = time_handle(qp, d); rule_v = qp.plan_info()["variant"]; qp.close()
    qp = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai, tuned=True); tuned_ms = time_handle(qp, d); tuned_v = qp.plan_info()["variant"]; qp.close()
    best = min([v for v in col.values() if v is not None] + [rule_ms])
    gap = tuned_ms / best - 1.0
    worst = max(worst, gap)
    print("%-22s " % ("%s N=%d" % (name, N)) + " ".join("%8s" % ("-" if col[f] is None else "%.3f" % col[f]) for f in FAMILIES) +
          "   %s %.3f   %s %.3f   %+.1f %%" % (NAME.get(rule_v, rule_v), rule_ms, NAME.get(tuned_v, tuned_v), tuned_ms, 100 * gap), flush=True)
    del d
print("worst gap of the tuned choice to the best column: %+.1f %%" % (100 * worst))
