#!/usr/bin/env python3
"""Eight-wave on-chip instances (long chains): parity against the oracle on small batches, then timings at full batch against the
global-block kernels the same sizes ran on before.   usage (GPU box): python tools/oc8_quick.py [parity|time|all] [env=val ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
from optimal_control_problem_amd import models            # noqa: E402
from optimal_control_problem_amd.batch_qp import BatchQP  # noqa: E402
from tests.support import problems                        # noqa: E402


def parity():
    os.environ["MPCQP_VARIANT"] = "oc8"
    bad = 0
    for name, B, N in [("quadrotor", 12, 50), ("cartpole", 12, 100), ("quadrotor", 9, 30), ("cartpole", 7, 70), ("double_integrator", 12, 100), ("quadrotor", 6, 25)]:
        mdl, ls, _ = models.make_workload(name, B, N=N)
        qp = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
        info = qp.plan_info()
        qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); got = qp.get(); qp.close()
        ref = problems.oracle_solve(ls)
        fin = np.isfinite(ref["x"]) & np.isfinite(got["x"])
        err = np.abs(got["x"][fin] - ref["x"][fin]).max() / (1 + np.abs(ref["x"][fin]).max()) if fin.any() else float("nan")
        ok = (got["status"] == ref["status"]).all() and (got["iters"] == ref["iters"]).all() and err < 1e-6
        bad += not ok
        print("%-18s N=%3d variant %d lds %6d: status %s iters %s / %s  rel|dx| %.2e  %s" % (name, N, info["variant"], info["lds_bytes"], got["status"][:4], got["iters"][:6], ref["iters"][:6], err,
                                                                                         "ok" if ok else "MISMATCH"), flush=True)
    del os.environ["MPCQP_VARIANT"]
    return bad


def timing():
    import torch
    dev = torch.device("cuda", 0)
    for name, N, B in [("quadrotor", 50, 8192), ("cartpole", 100, 16384), ("quadrotor", 30, 8192)]:
        mdl, ls, _ = models.make_workload(name, B, seed={"quadrotor": 2024, "cartpole": 7}[name], N=N)
        d = [torch.from_numpy(a).to(dev) for a in (ls.P, ls.q, ls.A, ls.l, ls.u)]
        res = {}
        for tag, env in [("oc8", {}), ("gres", {"MPCQP_NO_OC8": "1"})]:
            for k, v in env.items():
                os.environ[k] = v
            qp = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
            for k in env:
                del os.environ[k]
            qp.set_dispatch_hint(False)
            info = qp.plan_info()
            ms = []
            for it in range(4):
                qp.update(*d); qp.solve(); torch.cuda.synchronize(); ms.append(qp.last_kernel_ms())
            got = qp.get(); qp.close()
            res[tag] = got
            print("%-10s N=%3d x %5d %-4s variant %3d lds %6d: kernel %.2f ms (%s) = %.0f QP/s, mean iters %.1f, solved %.3f" % (
                name, N, B, tag, info["variant"], info["lds_bytes"], min(ms[1:]), " ".join("%.2f" % m for m in ms), B / (min(ms[1:]) * 1e-3), got["iters"].mean(), (got["status"] == 1).mean()), flush=True)
        a, b = res["oc8"], res["gres"]
        fin = np.isfinite(a["x"]) & np.isfinite(b["x"])
        print("   oc8 vs gres: iters equal %.4f, status equal %.4f, max|dx| %.2e" % ((a["iters"] == b["iters"]).mean(), (a["status"] == b["status"]).mean(), np.abs(a["x"][fin] - b["x"][fin]).max()), flush=True)


if __name__ == "__main__":
    rc = 0
    if mode in ("parity", "all"):
        rc = parity()
    if mode in ("time", "all") and rc == 0:
        timing()
    sys.exit(1 if rc else 0)
