"""Loader for tests/golden/qp_fixtures.npz (made by tests/golden/make_golden.py)."""
import os

import numpy as np

from optimal_control_problem_amd import models

_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "qp_fixtures.npz")


def load():
    z = np.load(_PATH)
    out = {}
    for name in z["names"]:
        name = str(name)
        n, m, B, np_ = [int(v) for v in z[name + "/dims"]]
        g = lambda k: z[name + "/" + k]
        ls = models.LocalSystem(n, m, g("Pp"), g("Pi"), g("Ap"), g("Ai"), g("P"), g("q"), g("A"), g("l"), g("u"), np_)
        out[name] = dict(ls=ls, x_star=g("x_star"), y_star=g("y_star"),
                         oracle={k: g("oracle_" + k) for k in ("x", "y", "iters", "status")},
                         analytic=z[name + "/analytic"] if name + "/analytic" in z.files else None)
    return out


NAMES = [str(n) for n in np.load(_PATH)["names"]] if os.path.exists(_PATH) else []
