"""Numerical robustness gate (GPU): a 60-pattern slice of the randomised sweep of tools/fuzz_gpu.py, every kernel family that takes the
pattern, against the oracle.

What is asserted, per family:
  * statuses and NaN patterns never differ;
  * an instance is at the TIGHT bar when iteration counts are equal and |x_gpu - x_oracle| <= 1e-6 relative -- the bar of every MPC workload
    in tests/test_gpu_parity.py -- and otherwise at TOLERANCE LEVEL: same status, iteration counts apart (a 25-iteration termination check or
    an adaptive-rho decision flipped on an ill-conditioned problem that needs hundreds of ADMM iterations), x apart by less than the
    eps = 1e-3 the reference configures (src/sqp_solver/SQPOptimizationSolver.cpp:83-84) allows.  At most 3 % of a family's solves may be
    tolerance-level (measured on the round-2 build: 1.7 - 3.2 % on 1000 patterns, profiles/r02_fuzz_gpu.txt; this slice: see DESIGN.md section 2);
  * for every tolerance-level instance the GPU's point is as good an answer as the oracle's by the reference's own measure: its TRUE
    residuals -- computed here in NumPy from the caller's unscaled data, r_p = |A x - clip(A x, l, u)|_inf and r_d = |P x + q + A' y|_inf,
    each divided by OSQP's termination threshold eps_abs + eps_rel * max(norms) -- are at most max(1, 2 x the oracle's): the point passes
    OSQP's termination test, or, where neither side terminated, is no worse than twice the oracle's.
"""
import os

import numpy as np
import pytest

from tests.support.problems import random_stage_ocp, sparse_batch

pytestmark = pytest.mark.gpu

FAMILIES = [None, "res1", "res4", "gres4", "stream"]
NPAT = 60


def _dense(ls, b):
    Pd, Ad = ls.dense(b)
    return np.triu(Pd) + np.triu(Pd, 1).T, Ad


def _normalised_residual(ls, b, x, y, eps=1e-3):
    """max(r_p / eps_p, r_d / eps_d) of (x, y) on instance b, with OSQP's unscaled termination thresholds"""
    P, A = _dense(ls, b)
    ax = A @ x; z = np.clip(ax, ls.l[b], ls.u[b])
    rp = np.abs(ax - z).max() if ls.m else 0.0
    px = P @ x; aty = A.T @ y
    rd = np.abs(px + ls.q[b] + aty).max()
    ep = eps + eps * max(np.abs(ax).max() if ls.m else 0.0, np.abs(z).max() if ls.m else 0.0)
    ed = eps + eps * max(np.abs(px).max(), np.abs(aty).max(), np.abs(ls.q[b]).max())
    return max(rp / ep, rd / ed)


def test_random_patterns_every_family(built, monkeypatch):
    from optimal_control_problem_amd import _lib
    from optimal_control_problem_amd.batch_qp import BatchQP
    from oracle import oracle as orc
    stats = {f: dict(solves=0, soft=0) for f in FAMILIES}
    worst = 0.0
    for c in range(NPAT):
        rng = np.random.default_rng(1000 + c)
        n = int(rng.integers(2, 140)); m = int(rng.integers(1, 200)); B = int(rng.integers(1, 9))
        dens = float(rng.choice([0.05, 0.15, 0.4, 1.0]))
        ls = sparse_batch(n, m, B, c, dens)
        ref = orc.Pattern(ls.n, ls.m, ls.Pp, ls.Pi, ls.Ap, ls.Ai).solve(ls.P, ls.q, ls.A, ls.l, ls.u, orc.default_settings())
        fin = np.isfinite(ref["x"])
        for fam in FAMILIES:
            if fam:
                monkeypatch.setenv("MPCQP_VARIANT", fam)
            else:
                monkeypatch.delenv("MPCQP_VARIANT", raising=False)
            try:
                qp = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
            except _lib.MpcqpError as e:
                if e.code == _lib.ERR_LIMIT:
                    continue
                raise
            qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); got = qp.get(); qp.close()
            tag = "pattern %d (n=%d m=%d B=%d dens=%.2f) family %s" % (c, n, m, B, dens, fam)
            assert (got["status"] == ref["status"]).all(), (tag, got["status"], ref["status"])
            assert np.array_equal(np.isfinite(got["x"]), fin), tag
            for b in range(B):
                stats[fam]["solves"] += 1
                if not fin[b].all():
                    # certificate instances: NaN on both sides; the certificate may pass its eps_inf test one check earlier or later
                    if got["iters"][b] != ref["iters"][b]:
                        assert abs(int(got["iters"][b]) - int(ref["iters"][b])) <= 25, tag
                        stats[fam]["soft"] += 1
                    continue
                scale = 1 + np.abs(ref["x"][b]).max()
                err = np.abs(got["x"][b] - ref["x"][b]).max() / scale
                if got["iters"][b] == ref["iters"][b] and err <= 1e-6:
                    continue
                stats[fam]["soft"] += 1
                assert err <= 2e-2, (tag, b, err, got["iters"][b], ref["iters"][b])
                rg = _normalised_residual(ls, b, got["x"][b], got["y"][b]); ro = _normalised_residual(ls, b, ref["x"][b], ref["y"][b])
                worst = max(worst, rg / max(ro, 1e-300))
                assert rg <= max(1.0, 2.0 * ro) * (1 + 1e-9), (tag, b, rg, ro)
    report = {str(f): "%d of %d" % (s["soft"], s["solves"]) for f, s in stats.items()}
    print("tolerance-level instances per family:", report, "worst residual ratio gpu / oracle among them: %.2f" % worst)
    for fam, s in stats.items():
        assert s["solves"] > 0, fam
        assert s["soft"] <= 0.03 * s["solves"], (fam, report)


OC_FAMILIES = [("oc4", 204), ("oc8", 208)]
OC_NPAT = 20


def test_random_stage_patterns_onchip_families(built, monkeypatch):
    """The same gate for the families that are the DEFAULT on the north-star workload and on BASELINE configs 3, 4, 5 -- the on-chip kernels, which
    only take block-tridiagonal + arrow patterns and so never see the random patterns above: a 20-pattern slice of tools/fuzz_oc.py per family
    (random nx / nu / N / weights / dynamics / iterate), a cold solve and a kept-workspace solve (new q, shifted bounds) each, under the
    reference's settings (src/sqp_solver/SQPOptimizationSolver.cpp:81-85)."""
    from optimal_control_problem_amd import _lib
    from optimal_control_problem_amd.batch_qp import BatchQP
    from oracle import oracle as orc
    stats = {f: dict(solves=0, soft=0, cases=0, dissected=0) for f, _ in OC_FAMILIES}
    worst = 0.0
    for fam, code in OC_FAMILIES:
        monkeypatch.setenv("MPCQP_VARIANT", fam)
        c = 0
        while stats[fam]["cases"] < OC_NPAT and c < 4 * OC_NPAT:
            ls, dims, rng = random_stage_ocp(c, fam); c += 1
            B = ls.batch
            try:
                qp = BatchQP(ls.n, ls.m, B, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
            except _lib.MpcqpError as e:
                if e.code == _lib.ERR_LIMIT:      # outside the instance's limits (chain length, LDS): not a case
                    continue
                raise
            assert qp.plan_info()["variant"] == code
            stats[fam]["cases"] += 1
            stats[fam]["dissected"] += qp.oc_info()["chain_pairs"] > 1       # (separators of the stage chain in the hub block: several twisted pairs of chains)
            qp.keep_workspace(True)
            st = orc.State(orc.Pattern(ls.n, ls.m, ls.Pp, ls.Pi, ls.Ap, ls.Ai), B, orc.default_settings())
            q2 = ls.q * 1.2 + 0.05 * rng.normal(size=ls.q.shape); sh = 0.02 * rng.normal(size=ls.l.shape)
            for leg in ("cold", "kept"):
                if leg == "cold":
                    qp.update(ls.P, ls.q, ls.A, ls.l, ls.u); qp.solve(); got = qp.get()
                    ref = st.solve(ls.P, ls.q, ls.A, ls.l, ls.u); q, l, u = ls.q, ls.l, ls.u
                else:
                    qp.update_vectors(q2, ls.l + sh, ls.u + sh); qp.solve(); got = qp.get()
                    ref = st.solve_vectors(q2, ls.l + sh, ls.u + sh); q, l, u = q2, ls.l + sh, ls.u + sh
                tag = "%s case %d (%s) %s" % (fam, c - 1, dims, leg)
                fin = np.isfinite(ref["x"])
                assert (got["status"] == ref["status"]).all(), (tag, got["status"], ref["status"])
                assert np.array_equal(np.isfinite(got["x"]), fin), tag
                for b in range(B):
                    stats[fam]["solves"] += 1
                    if not fin[b].all():
                        if got["iters"][b] != ref["iters"][b]:
                            assert abs(int(got["iters"][b]) - int(ref["iters"][b])) <= 25, tag
                            stats[fam]["soft"] += 1
                        continue
                    err = np.abs(got["x"][b] - ref["x"][b]).max() / (1 + np.abs(ref["x"][b]).max())
                    if got["iters"][b] == ref["iters"][b] and err <= 1e-6:
                        continue
                    stats[fam]["soft"] += 1
                    assert err <= 2e-2, (tag, b, err, got["iters"][b], ref["iters"][b])
                    lsb = type(ls)(ls.n, ls.m, ls.Pp, ls.Pi, ls.Ap, ls.Ai, ls.P, q, ls.A, l, u)
                    rg = _normalised_residual(lsb, b, got["x"][b], got["y"][b]); ro = _normalised_residual(lsb, b, ref["x"][b], ref["y"][b])
                    worst = max(worst, rg / max(ro, 1e-300))
                    assert rg <= max(1.0, 2.0 * ro) * (1 + 1e-9), (tag, b, rg, ro)
            qp.close()
    report = {f: "%d of %d solves, %d patterns (%d in the dissected order)" % (s["soft"], s["solves"], s["cases"], s["dissected"]) for f, s in stats.items()}
    print("on-chip families, tolerance-level instances:", report, "worst residual ratio gpu / oracle among them: %.2f" % worst)
    for fam, s in stats.items():
        assert s["cases"] == OC_NPAT, (fam, report)
        assert 0 < s["dissected"] < s["cases"], (fam, report)        # both orders are in the gate
        assert s["soft"] <= 0.03 * s["solves"], (fam, report)
