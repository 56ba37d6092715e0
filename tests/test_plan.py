"""CPU: host-side plan (ordering, ELL layouts, block pattern, assembly recipe, factor op list, block streams)
executed by the scalar interpreter in tests/support/plan_interp.cpp and compared with dense linear algebra."""
import ctypes as C
import os

import numpy as np
import pytest

from optimal_control_problem_amd import models
from tests.support import problems

SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "support", "libplan_interp.so")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _run(ls, b=0, force=-1, seed=0):
    L = C.CDLL(SO)
    rng = np.random.default_rng(seed)
    n, m = ls.n, ls.m
    info = np.zeros(8, np.int64); pos = np.zeros(n, np.int32)
    assert L.plan_describe(n, m, _p(ls.Pp), _p(ls.Pi), _p(ls.Ap), _p(ls.Ai), force, _p(info), _p(pos)) == 0
    assert sorted(pos.tolist()) == list(range(n))
    Pd, Ad = ls.dense(b)
    Pd = np.triu(Pd) + np.triu(Pd, 1).T
    rho = rng.choice([0.1, 100.0, 1e-6], size=m); sigma = 1e-6
    M = Pd + sigma * np.eye(n) + Ad.T @ (rho[:, None] * Ad)
    rhs = rng.normal(size=n); xin = rng.normal(size=n); win = rng.normal(size=m)
    sol = np.zeros(n); Ax = np.zeros(m); Atw = np.zeros(n); Px = np.zeros(n)
    Pv = np.ascontiguousarray(ls.P[b]); Av = np.ascontiguousarray(ls.A[b])
    rc = L.plan_execute(n, m, _p(ls.Pp), _p(ls.Pi), _p(ls.Ap), _p(ls.Ai), force, _p(Pv), _p(Av), _p(rho), C.c_double(sigma),
                        _p(rhs), _p(sol), _p(xin), _p(win), _p(Ax), _p(Atw), _p(Px))
    assert rc == 0
    ref = np.linalg.solve(M, rhs)
    rel = lambda a, r: np.abs(a - r).max() / max(np.abs(r).max(), 1e-300)
    assert rel(sol, ref) < 1e-9
    assert rel(Ax, Ad @ xin) < 1e-13 and rel(Atw, Ad.T @ win) < 1e-13 and rel(Px, Pd @ xin) < 1e-13
    return dict(npad=info[0], mpad=info[1], nb=info[2], nblk=info[3], nT=info[4], nfac=info[5], ordering=info[6], lds=info[7])


@pytest.mark.parametrize("idx", range(7))
def test_plan_toy(built, idx):
    mdl, arg, _ = models.reference_test_cases()[idx]
    info = _run(problems.toy_local_system(mdl, arg))
    assert info["nblk"] == 1


@pytest.mark.parametrize("name,N,blocks", [("double_integrator", 20, 9), ("quadrotor", 20, 60), ("cartpole", 30, None), ("quadrotor", 50, 150)])
def test_plan_stage_models(built, name, N, blocks):
    mdl, ls, _ = models.make_workload(name, 2, N=N)
    info = _run(ls, 1)
    assert info["ordering"] == 1                      # the parameter block p is recognised as a hub and moved last
    if blocks:
        assert info["nblk"] == blocks                 # block-tridiagonal + arrow, no extra fill
    nat = _run(ls, 0, force=0)
    assert nat["nblk"] > info["nblk"]                 # natural order (p first) fills in


@pytest.mark.parametrize("seed", range(4))
def test_plan_random(built, seed):
    _run(problems.random_qp(9 + 11 * seed, 13 + 17 * seed, seed, density=0.2), seed=seed)


def test_plan_no_constraints_rows_only_boxes(built):
    mdl, arg, _ = models.reference_test_cases()[1]    # no g rows: A is the identity
    info = _run(problems.toy_local_system(mdl, arg))
    assert info["nT"] == 0


# ---- resident variant: block LDL' factor plan + phase schedule over nw waves (race-checked by the interpreter)
def _run_res(ls, nw, b=0, seed=0):
    L = C.CDLL(SO)
    rng = np.random.default_rng(seed)
    n, m = ls.n, ls.m
    Pd, Ad = ls.dense(b)
    Pd = np.triu(Pd) + np.triu(Pd, 1).T
    rho = rng.choice([0.1, 100.0, 1e-6], size=m); sigma = 1e-6
    M = Pd + sigma * np.eye(n) + Ad.T @ (rho[:, None] * Ad)
    rhs = rng.normal(size=n); sol = np.zeros(n); info = np.zeros(4, np.int64)
    Pv = np.ascontiguousarray(ls.P[b]); Av = np.ascontiguousarray(ls.A[b])
    rc = L.plan_execute_res(n, m, _p(ls.Pp), _p(ls.Pi), _p(ls.Ap), _p(ls.Ai), nw, _p(Pv), _p(Av), _p(rho), C.c_double(sigma),
                            _p(rhs), _p(sol), _p(info))
    assert rc == 0, rc
    ref = np.linalg.solve(M, rhs)
    assert np.abs(sol - ref).max() / np.abs(ref).max() < 1e-9
    return dict(ntemp=info[0] % 100, ordering=info[0] // 100, nphase=info[1], lds=info[2], barriers=info[3] % 1000, segments=info[3] // 1000)


@pytest.mark.parametrize("nw", [1, 2, 4, 8])
@pytest.mark.parametrize("name,N", [("double_integrator", 20), ("quadrotor", 20), ("cartpole", 30)])
def test_res_plan_stage_models(built, name, N, nw):
    mdl, ls, _ = models.make_workload(name, 1, N=N)
    info = _run_res(ls, nw)
    assert info["ntemp"] == 2                                   # next-stage block + parameter (arrow) block per column
    if nw > 1:
        assert info["barriers"] == 4                            # arrow phase, diagonal phase, and the two chain ends
    if name == "quadrotor":
        assert info["nphase"] == 41 and info["lds"] < 160 * 1024
        # 100 block ops compress into a handful of arithmetic-progression segments per wave
        assert info["segments"] <= 6 * nw + 4


@pytest.mark.parametrize("seed", range(3))
def test_res_plan_random(built, seed):
    ls = problems.random_qp(20 + 13 * seed, 30 + 11 * seed, seed, density=0.15)
    for nw in (1, 2, 4):
        _run_res(ls, nw, seed=seed)


def test_res_plan_rejects_indefinite(built):
    mdl, arg, _ = models.reference_test_cases()[7]
    ls = problems.toy_local_system(mdl, arg)
    L = C.CDLL(SO)
    rho = np.full(ls.m, 0.1); rhs = np.ones(ls.n); sol = np.zeros(ls.n); info = np.zeros(4, np.int64)
    Pv = np.ascontiguousarray(ls.P[0]); Av = np.ascontiguousarray(ls.A[0])
    assert L.plan_execute_res(ls.n, ls.m, _p(ls.Pp), _p(ls.Pi), _p(ls.Ap), _p(ls.Ai), 1, _p(Pv), _p(Av), _p(rho), C.c_double(1e-6),
                              _p(rhs), _p(sol), _p(info)) == 2


@pytest.mark.parametrize("nw", [2, 4, 8])
def test_res_plan_twisted_ordering(built, nw):
    """two-sided elimination of the stage chain (ordering 2, passed as nw + 200): same block count and LDS footprint,
    roughly half the phases, still hazard-free and exact"""
    mdl, ls, _ = models.make_workload("quadrotor", 1, N=20)
    base = _run_res(ls, nw)
    tw = _run_res(ls, nw + 200)
    sp = _run_res(ls, nw + 200 + 10000)                          # the same with the arrow run split over the waves (global-block plans)
    if nw in (2, 4):                                               # 20 arrow ops >= 4 * nw: split; with 8 waves the run stays whole
        assert sp["nphase"] == tw["nphase"] + 1 and sp["barriers"] <= tw["barriers"] + 2
    else:
        assert sp["nphase"] == tw["nphase"]
    assert base["ordering"] == 1 and tw["ordering"] == 2
    assert tw["nphase"] <= base["nphase"] // 2 + 3
    assert abs(tw["lds"] - base["lds"]) < 6144 and tw["lds"] <= 160 * 1024    # two more temp tiles (both columns of a level)
    # stage frames that do not tile 16-blocks (cart-pole f = 5) keep the plain hubs-last order
    mdl, ls, _ = models.make_workload("cartpole", 1, N=30)
    assert _run_res(ls, nw + 200)["ordering"] == 1


@pytest.mark.parametrize("name,N", [("double_integrator", 20), ("quadrotor", 20), ("quadrotor", 50), ("cartpole", 100)])
def test_slab_layout_and_schedule_bounds(built, name, N):
    """host-side guards for what the kernels will address: slab regions ordered, aligned and large enough; every schedule record
    and expanded segment inside the block array and the solve vector (incl. the partial-sum slots of split runs)"""
    L = C.CDLL(SO)
    L.plan_check_layout.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_int]
    L.plan_check_segments.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_void_p]
    mdl, ls, _ = models.make_workload(name, 1, N=N)
    for force in (-1, 0, 1, 2):
        assert L.plan_check_layout(ls.n, ls.m, _p(ls.Pp), _p(ls.Pi), _p(ls.Ap), _p(ls.Ai), force) == 0
        for nw in (1, 2, 4, 8):
            info = np.zeros(8, np.int64)
            for split in (0, 10000):
                assert L.plan_check_segments(ls.n, ls.m, _p(ls.Pp), _p(ls.Pi), _p(ls.Ap), _p(ls.Ai), nw + 100 * (force + 1) + split, _p(info)) == 0


def _run_oc(ls, NG=5, NH=3, b=0, seed=0, ldl=1):
    """the on-chip plan (plan.hpp build_oc_plan) through the lane-accurate emulation of kernel_onchip.hpp's solve in plan_interp.cpp:
    MFMA operand layouts, phi, swizzled LDS images read as rows and columns, chain tables, phantom slots, junction and hub phases"""
    L = C.CDLL(SO)
    rng = np.random.default_rng(seed)
    n, m = ls.n, ls.m
    Pd, Ad = ls.dense(b)
    Pd = np.triu(Pd) + np.triu(Pd, 1).T
    rho = rng.choice([0.1, 100.0, 1e-6], size=m); sigma = 1e-6
    M = Pd + sigma * np.eye(n) + Ad.T @ (rho[:, None] * Ad)
    rhs = rng.normal(size=n); sol = np.zeros(n); info = np.zeros(8, np.int64)
    Pv = np.ascontiguousarray(np.broadcast_to(ls.P, (ls.batch, len(ls.Pi)))[b]); Av = np.ascontiguousarray(np.broadcast_to(ls.A, (ls.batch, len(ls.Ai)))[b])
    rc = L.plan_execute_oc(n, m, _p(ls.Pp), _p(ls.Pi), _p(ls.Ap), _p(ls.Ai), NG, NH, ldl, _p(Pv), _p(Av), _p(rho), C.c_double(sigma), _p(rhs), _p(sol), _p(info))
    if rc == 0:
        ref = np.linalg.solve(M, rhs)
        assert np.abs(sol - ref).max() / np.abs(ref).max() < 1e-9
    return rc, dict(nbc=info[0], has_hub=info[1], junc=info[2], nlds=info[3], nhr=info[4], lds=info[5])


@pytest.mark.parametrize("name,N,expect", [("quadrotor", 20, dict(nbc=20, has_hub=1, junc=1, nlds=28, nhr=3)),      # the north-star size: twisted chains + hub
                                           ("quadrotor", 12, dict(nbc=12, has_hub=1, junc=1, nhr=3)),              # phantom slots (12 positions, 5 per wave)
                                           ("cartpole", 30, dict(has_hub=1, junc=0)),                             # one chain, the hub shares the last block
                                           ("double_integrator", 20, dict(has_hub=1, junc=0))])
@pytest.mark.parametrize("ldl", [0, 1])      # the factor from the level loop / from the in-register LDL' (oc_ldl), both emulated
def test_onchip_plan_emulated(built, name, N, expect, ldl):
    mdl, ls, _ = models.make_workload(name, 2, N=N)
    rc, info = _run_oc(ls, b=1, ldl=ldl)
    assert rc == 0, (rc, info)
    for k, v in expect.items():
        assert info[k] == v, (k, info)
    if name == "quadrotor" and N == 20:
        assert info["lds"] <= 80 * 1024                   # two workgroups per CU


def _run_oc8(ls, NG, NH, zyg, b=0, seed=0, ldl=1):
    """the eight-wave on-chip plan (ordering 3: padded twist; chains of any length) through the same emulation"""
    L = C.CDLL(SO)
    L.plan_execute_oc_nw.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p] * 3 + [C.c_double] + [C.c_void_p] * 3
    rng = np.random.default_rng(seed)
    n, m = ls.n, ls.m
    Pd, Ad = ls.dense(b)
    Pd = np.triu(Pd) + np.triu(Pd, 1).T
    rho = rng.choice([0.1, 100.0, 1e-6], size=m); sigma = 1e-6
    M = Pd + sigma * np.eye(n) + Ad.T @ (rho[:, None] * Ad)
    rhs = rng.normal(size=n); sol = np.zeros(n); info = np.zeros(8, np.int64)
    Pv = np.ascontiguousarray(np.broadcast_to(ls.P, (ls.batch, len(ls.Pi)))[b]); Av = np.ascontiguousarray(np.broadcast_to(ls.A, (ls.batch, len(ls.Ai)))[b])
    rc = L.plan_execute_oc_nw(n, m, _p(ls.Pp), _p(ls.Pi), _p(ls.Ap), _p(ls.Ai), 8, NG, NH, ldl, zyg, _p(Pv), _p(Av), _p(rho), C.c_double(sigma), _p(rhs), _p(sol), _p(info))
    if rc == 0:
        ref = np.linalg.solve(M, rhs)
        assert np.abs(sol - ref).max() / np.abs(ref).max() < 1e-9
    return rc, dict(nbc=info[0], has_hub=info[1], junc=info[2], nlds=info[3], nhr=info[4], lds=info[5], LE=info[6], LF=info[7])


@pytest.mark.parametrize("name,N,inst,expect", [("quadrotor", 50, (7, 7, 0), dict(nbc=50, has_hub=1, junc=1, nlds=50, LE=24, LF=26)),       # BASELINE config 3 as mpcqp_create takes it: 49 chain blocks + the hub's inverse in LDS, every hub block in registers
                                                ("quadrotor", 50, (7, 5, 1), dict(nbc=50, has_hub=1, junc=1, nlds=60)),                      # (the alternative split: ten hub blocks in LDS, z / y in the slab)
                                                ("cartpole", 100, (4, 4, 0), dict(nbc=32, has_hub=1, junc=1, nlds=32, LE=15, LF=17)),        # config 4: padded twist, two chains instead of one of 31
                                                ("quadrotor", 30, (4, 4, 0), dict(nbc=30, junc=1)), ("quadrotor", 40, (7, 7, 0), dict(nbc=40)),
                                                ("cartpole", 150, (7, 7, 0), dict(nbc=47, junc=1)), ("double_integrator", 100, (4, 4, 0), dict(nbc=19))])
@pytest.mark.parametrize("ldl", [0, 1])
def test_onchip_long_chain_plan_emulated(built, name, N, inst, expect, ldl):
    """the eight-wave instances' tables (kernel_onchip.hpp with NW = 8: per-wave partial sums 1 .. 8 and the zero block behind them, positions
    p = w + 8 s, idle waves in the factorisation) on the sizes the four-wave plan refuses"""
    mdl, ls, _ = models.make_workload(name, 2, N=N)
    rc, info = _run_oc8(ls, *inst, b=1, ldl=ldl)
    assert rc == 0, (rc, info)
    for k, v in expect.items():
        assert info[k] == v, (k, info)
    assert info["lds"] <= 160 * 1024                       # one workgroup per CU
    if N > 20 and name == "quadrotor":
        assert _run_oc(ls)[0] == 5                         # ... and the four-wave plan does not take it


def _run_oc8_dissected(ls, NG, NH, b=0, seed=0, ldl=1):
    """ordering 4 (plan.hpp build_plan: separators of the stage chain in the hub block, several twisted pairs) through the same emulation"""
    L = C.CDLL(SO)
    L.plan_execute_oc_dissected.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p] * 3 + [C.c_double] + [C.c_void_p] * 3
    rng = np.random.default_rng(seed)
    n, m = ls.n, ls.m
    Pd, Ad = ls.dense(b)
    Pd = np.triu(Pd) + np.triu(Pd, 1).T
    rho = rng.choice([0.1, 100.0, 1e-6], size=m); sigma = 1e-6
    M = Pd + sigma * np.eye(n) + Ad.T @ (rho[:, None] * Ad)
    rhs = rng.normal(size=n); sol = np.zeros(n); info = np.zeros(8, np.int64)
    Pv = np.ascontiguousarray(np.broadcast_to(ls.P, (ls.batch, len(ls.Pi)))[b]); Av = np.ascontiguousarray(np.broadcast_to(ls.A, (ls.batch, len(ls.Ai)))[b])
    rc = L.plan_execute_oc_dissected(n, m, _p(ls.Pp), _p(ls.Pi), _p(ls.Ap), _p(ls.Ai), NG, NH, ldl, _p(Pv), _p(Av), _p(rho), C.c_double(sigma), _p(rhs), _p(sol), _p(info))
    if rc == 0:
        ref = np.linalg.solve(M, rhs)
        assert np.abs(sol - ref).max() / np.abs(ref).max() < 1e-9
    return rc, dict(nbc=info[0], has_hub=info[1], npairs=info[2], nlds=info[3], nhr=info[4], lds=info[5], LE=info[6], LF=info[7])


@pytest.mark.parametrize("name,N,inst,expect", [("cartpole", 100, (4, 4), dict(nbc=32, has_hub=1, npairs=4, nlds=29, LE=3, LF=5)),        # BASELINE config 4: 3 separators of 4 states beside the 4 parameters
                                                ("cartpole", 150, (7, 7), dict(nbc=48, npairs=4, LE=5, LF=7)), ("cartpole", 60, (4, 4), dict(nbc=20, npairs=4)),
                                                ("double_integrator", 100, (4, 4), dict(nbc=20, npairs=4))])
@pytest.mark.parametrize("ldl", [0, 1])
def test_onchip_dissected_plan_emulated(built, name, N, inst, expect, ldl):
    """the dissected order: where the hub block has room, separators of the stage chain join the hub and the chain falls into segments, each a twisted pair
    of short chains that chain waves 2 i, 2 i + 1 walk in the solve and the factorisation takes one after the other -- factor (level-parallel or the emulated
    oc_ldl) and solve, lane-accurately, against dense linear algebra, with the hazard checks of the other emulations; no block more than the twisted order has"""
    mdl, ls, _ = models.make_workload(name, 2, N=N)
    rc, info = _run_oc8_dissected(ls, *inst, b=1, ldl=ldl)
    assert rc == 0, (rc, info)
    for k, v in expect.items():
        assert info[k] == v, (k, info)
    assert info["lds"] <= 160 * 1024
    rc0, info0 = _run_oc8(ls, *inst, 0, b=1, ldl=ldl)
    assert rc0 == 0 and info["nlds"] <= info0["nlds"] and info["LF"] < info0["LF"] / 2


@pytest.mark.parametrize("name,N,expect", [("cartpole", 30, dict(nbc=10, npairs=2, LE=2, LF=3)), ("cartpole", 50, dict(nbc=16, npairs=2, LE=3, LF=5)),
                                           ("double_integrator", 60, dict(nbc=12, npairs=2))])
@pytest.mark.parametrize("ldl", [0, 1])
def test_onchip_dissected_plan_four_waves_emulated(built, name, N, expect, ldl):
    """the same order with ONE separator for the four-wave instances: two twisted pairs, chain waves 0 .. 3 (their own kernel instances run the eight-wave
    instances' solve text); too short a chain, or no room in the hub, and the order does not apply"""
    L = C.CDLL(SO)
    L.plan_execute_oc_dissected4.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p] * 3 + [C.c_double] + [C.c_void_p] * 3
    mdl, ls, _ = models.make_workload(name, 2, N=N)
    rng = np.random.default_rng(3)
    n, m, b = ls.n, ls.m, 1
    Pd, Ad = ls.dense(b)
    Pd = np.triu(Pd) + np.triu(Pd, 1).T
    rho = rng.choice([0.1, 100.0, 1e-6], size=m); sigma = 1e-6
    M = Pd + sigma * np.eye(n) + Ad.T @ (rho[:, None] * Ad)
    rhs = rng.normal(size=n); sol = np.zeros(n); info = np.zeros(8, np.int64)
    Pv = np.ascontiguousarray(np.broadcast_to(ls.P, (ls.batch, len(ls.Pi)))[b]); Av = np.ascontiguousarray(np.broadcast_to(ls.A, (ls.batch, len(ls.Ai)))[b])
    rc = L.plan_execute_oc_dissected4(n, m, _p(ls.Pp), _p(ls.Pi), _p(ls.Ap), _p(ls.Ai), 5, 3, ldl, _p(Pv), _p(Av), _p(rho), C.c_double(sigma), _p(rhs), _p(sol), _p(info))
    assert rc == 0
    ref = np.linalg.solve(M, rhs)
    assert np.abs(sol - ref).max() / np.abs(ref).max() < 1e-9
    got = dict(nbc=info[0], npairs=info[2], LE=info[6], LF=info[7])
    for k, v in expect.items():
        assert got[k] == v, (k, got)
    assert info[5] <= 80 * 1024                                   # two workgroups per CU
    for nm, NN in (("cartpole", 20), ("double_integrator", 40), ("quadrotor", 20)):
        mdl, l2, _ = models.make_workload(nm, 1, N=NN)
        Pv2 = np.ascontiguousarray(np.broadcast_to(l2.P, (1, len(l2.Pi)))[0]); Av2 = np.ascontiguousarray(np.broadcast_to(l2.A, (1, len(l2.Ai)))[0])
        r2 = np.ones(l2.m); s2 = np.zeros(l2.n)
        assert L.plan_execute_oc_dissected4(l2.n, l2.m, _p(l2.Pp), _p(l2.Pi), _p(l2.Ap), _p(l2.Ai), 5, 3, ldl, _p(Pv2), _p(Av2), _p(r2), C.c_double(sigma), _p(s2.copy()), _p(s2), _p(info)) == 5


def test_dissected_order_needs_room_in_the_hub(built):
    """the quadrotor's 12 parameters leave 4 places in the hub block, its separators are 12 states: the order does not apply (mpcqp_create keeps the padded twist)"""
    mdl, ls, _ = models.make_workload("quadrotor", 1, N=50)
    assert _run_oc8_dissected(ls, 7, 7)[0] == 5


@pytest.mark.parametrize("name,N,order,ntile", [("quadrotor", 20, 2, 19), ("quadrotor", 50, 3, 49), ("quadrotor", 10, 2, 9), ("cartpole", 100, 3, 0), ("double_integrator", 20, 2, 0)])
def test_tile_plan_products(built, name, N, order, ntile):
    """dense tiles of A for the iteration's sweeps (plan.hpp build_tile_plan; opt-in MPCQP_TILES=1): every entry of A exactly once in a tile or
    in each remainder layout; A x and A' w from the tiles -- through the emulated 4-block MFMA in the kernel's lane layouts -- plus the
    remainders equal the CSC products; patterns without dense blocks get no tiles"""
    L = C.CDLL(SO)
    L.plan_tile_check.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 5
    mdl, ls, _ = models.make_workload(name, 2, N=N)
    rng = np.random.default_rng(0)
    Av = np.ascontiguousarray(np.broadcast_to(ls.A, (ls.batch, len(ls.Ai)))[1]); x = rng.normal(size=ls.n); w = rng.normal(size=ls.m)
    err = C.c_double(0); out = np.zeros(8, np.int64)
    rc = L.plan_tile_check(ls.n, ls.m, _p(ls.Pp), _p(ls.Pi), _p(ls.Ap), _p(ls.Ai), order, _p(Av), _p(x), _p(w), C.byref(err), _p(out))
    if ntile == 0:
        assert rc == 4 and out[0] == 0
        return
    assert rc == 0 and out[0] == ntile and err.value < 1e-13
    assert out[2] < out[4] / 4 and out[3] < out[5] / 4 and out[7] == 1          # the remainders are a fraction of the ELL layouts; one tile per column block


def test_onchip_long_chain_plan_limits(built):
    mdl, ls, _ = models.make_workload("quadrotor", 1, N=57)          # 57 chain blocks: more than seven positions per wave
    assert _run_oc8(ls, 7, 7, 0)[0] == 5
    mdl, ls, _ = models.make_workload("quadrotor", 1, N=40)          # 40 positions do not fit the <4, 4> instance
    assert _run_oc8(ls, 4, 4, 0)[0] == 5 and _run_oc8(ls, 7, 7, 0)[0] == 0
    mdl, ls, _ = models.make_workload("quadrotor", 1, N=56)          # the tables exist, but the LDS of one CU does not hold it: mpcqp_create keeps the global-block kernel
    rc, info = _run_oc8(ls, 7, 7, 0)
    assert rc == 0 and info["lds"] > 160 * 1024


def test_onchip_plan_without_hub_and_limits(built):
    """the reduced form's pattern (no parameter block): two chains meeting in their last element, no hub phases; sizes past the
    instance's limits are refused by the plan, not mis-executed"""
    mdl, ls, _ = models.make_workload("quadrotor", 2, N=20)
    red, *_ = problems.reduce_qp(ls, list(range(mdl.np)))
    rc, info = _run_oc(red, NG=5, NH=0, b=1)
    assert rc == 0 and (info["nbc"], info["has_hub"], info["junc"], info["nlds"]) == (20, 0, 1, 19)
    mdl, ls, _ = models.make_workload("quadrotor", 1, N=30)          # 30 chain blocks: more than five positions per wave
    assert _run_oc(ls)[0] == 5
    assert _run_oc(problems.random_qp(40, 30, 3))[0] == 0            # three blocks: a dense pattern still is chain + hub
    assert _run_oc(problems.random_qp(120, 60, 3))[0] == 5           # eight dense blocks are not block tridiagonal + arrow


@pytest.mark.parametrize("name,N,late", [("quadrotor", 20, (4, 5)),      # 5 1/4 chunks of A' on four waves: the last chain positions polled, the hub's rows free
                                         ("quadrotor", 12, (-1, -1)),    # four chunks: nothing to move
                                         ("cartpole", 30, (-1, -1))])
def test_onchip_plan_extras(built, name, N, late):
    """host-side pieces of the on-chip mode added with its set-up rewrite: which chunks of A' may be computed during the chain phase
    (plan.hpp oc_late_chunks), the batch-aware ELL padding, the assembly records and the zero tile they point unused terms at"""
    L = C.CDLL(SO)
    mdl, ls, _ = models.make_workload(name, 1, N=N)
    out = np.zeros(8, np.int32)
    assert L.plan_oc_extras(ls.n, ls.m, _p(ls.Pp), _p(ls.Pi), _p(ls.Ap), _p(ls.Ai), _p(out)) == 0
    assert (out[0], out[1]) == late
    assert out[2] == 1 and out[3] == 1 and out[4] == 1
