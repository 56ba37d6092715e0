#!/usr/bin/env python3
"""bench.py -- QP solves/sec of the batched OSQP-style ADMM hot path on N MI355X (one process per GPU).

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N > 1 launched through
torch.distributed.run, one rank per GPU.  A "step" is one pass of the hot path over one batch that is
already resident in HBM: mpcqp_update (borrow device pointers) + mpcqp_solve (scaling, factorisation, ADMM
to eps_abs = eps_rel = 1e-3) + mpcqp_get (device-to-device copy of x, y, status, iters).

Workload (config.workload): the configuration BASELINE.json's metric is quoted on -- 12-state quadrotor,
horizon N = 20, batch 8192 per GPU, each instance linearised about its own perturbed hover trajectory
(reference formulation n = 332, m = 560).  Weak scaling: every rank solves its
own 8192 instances (seed 2024 + rank); no collective on the data path, one gather of the solutions at the end.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters)
FP64_VEC_PEAK_TFLOPS = 78.6
LDS_PEAK_GBS = 150000.0    # every CU streaming ds_read_b64 / b128: 256 B/clk/CU x 256 CUs at ~2.4 GHz (same guide, section LDS)
MFMA_F64_4X4_DEP_CYCLES = 52   # a v_mfma_f64_4x4x4_4b_f64 that waits for the previous one's accumulator (tools/probes/mfma_4x4_probe.hip)
SHADER_CLOCK_GHZ = 2.4
CUS = 256


def onchip_rooflines(pinfo, ocinfo, n, m, batch, iters_mean, kernel_ms):
    """SURVEY.md section 8(d) beyond HBM, for the on-chip kernels: the LDS-bandwidth fraction and the share of a QP's residence that is the
    dependent MFMA chain of the two triangular sweeps.  ALGORITHMIC LDS bytes of one ADMM iteration: every factor block
    that lives in LDS read once per sweep direction (2 x 2 KiB), one 8-byte gathered operand per ELL slot entry of A and A', and the iterate
    vectors (x, q, r twice in the solve: 8 passes over npad; z, y, w: 6 over mpad).  mfma_issue_frac: per iteration each sweep walks the longer chain
    position by position, 4 dependent 4x4x4 MFMAs each, at the dependent issue interval, over the cycles a QP is resident (kernel time x the QPs a CU
    holds at once)."""
    if pinfo["variant"] < 200:
        return None, None
    nw = pinfo["variant"] - 200
    lds_iter = 8 * (2 * ocinfo["lds_blocks"] * 256 + (ocinfo["slots_A"] + ocinfo["slots_At"]) * 64 + 8 * pinfo["npad"] + 6 * pinfo["mpad"])
    ach = lds_iter * iters_mean * batch / (kernel_ms * 1e-3) / 1e9
    lds = {"bound": "lds", "achieved": ach, "peak": LDS_PEAK_GBS, "unit": "GB/s", "frac": ach / LDS_PEAK_GBS, "algorithmic_lds_bytes_per_admm_iter": lds_iter}
    resident = CUS * (2 if nw == 4 else 1)
    cycles_per_qp = kernel_ms * 1e-3 * SHADER_CLOCK_GHZ * 1e9 * min(resident, batch) / batch
    stages = 2 * max(ocinfo["chain_e"], ocinfo["chain_f"])
    dep = iters_mean * stages * 4 * MFMA_F64_4X4_DEP_CYCLES
    return lds, {"value": dep / cycles_per_qp, "dependent_mfma_cycles_per_admm_iter": stages * 4 * MFMA_F64_4X4_DEP_CYCLES, "chain_positions": max(ocinfo["chain_e"], ocinfo["chain_f"]), "chain_pairs": ocinfo.get("chain_pairs", 1),
                 "resident_cycles_per_qp": cycles_per_qp, "note": "chains of the forward and backward sweep: 4 dependent v_mfma_f64_4x4x4_4b_f64 per position at %d cycles, over kernel time x QPs resident per CU (%d) at %.1f GHz" % (MFMA_F64_4X4_DEP_CYCLES, 2 if nw == 4 else 1, SHADER_CLOCK_GHZ)}


def host_cores():
    """CPU threads this process may actually use: min(logical CPUs, affinity mask, cgroup cpu quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0]); period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
        except (OSError, ValueError, IndexError):
            pass
    return n


def lib_sha16():
    """first 16 hex digits of sha256(libmpcqp.so): the build a committed PMC figure belongs to"""
    import hashlib
    from optimal_control_problem_amd import _lib
    try:
        return hashlib.sha256(open(_lib.lib_path(), "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def committed_traffic(key):
    """HBM bytes per launch of this workload from the committed rocprofv3 PMC passes -- only if they were taken on THIS build of the
    library (profiles/traffic_table.json keys each entry with the library's hash); anything else is reported as null, not as a stale number"""
    try:
        tbl = json.load(open(os.path.join(ROOT, "profiles", "traffic_table.json")))
    except (OSError, ValueError):
        return None
    e = tbl.get(key)
    if not isinstance(e, dict) or e.get("lib_sha16") is None or e.get("lib_sha16") != lib_sha16():
        return None
    return e.get("hbm_bytes_per_launch")


def one_config(workload, N, batch, seed, dev, steps=5):
    """another BASELINE configuration: one warm-up step, then `steps` timed ones (same step as the timed region of `value`: update + solve + get, instances in
    batch order); value from the MEDIAN step"""
    import torch
    from optimal_control_problem_amd import models
    from optimal_control_problem_amd.batch_qp import BatchQP
    t0 = time.time()
    mdl, ls, _ = models.make_workload(workload, batch, seed=seed, N=N)
    t_gen = time.time() - t0
    d = [torch.from_numpy(a).to(dev) for a in (ls.P, ls.q, ls.A, ls.l, ls.u)]
    ox = torch.empty(batch, ls.n, dtype=torch.float64, device=dev); oy = torch.empty(batch, ls.m, dtype=torch.float64, device=dev)
    ost = torch.empty(batch, dtype=torch.int32, device=dev); oit = torch.empty(batch, dtype=torch.int32, device=dev)
    qp = BatchQP(ls.n, ls.m, batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai, device=dev.index)
    qp.set_dispatch_hint(False)
    stream = torch.cuda.current_stream().cuda_stream
    def step():
        qp.update(*d); qp.solve(stream); qp.get_device(x=ox, y=oy, status=ost, iters=oit)
    step(); torch.cuda.synchronize()
    # every step between its own pair of events on the launch stream (whole step: update + solve + get), read after the loop -- waiting inside it
    # would park the host once per step; a second pair around the solve alone for the kernels' time
    evs = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(4)) for _ in range(steps)]
    for a, e0, e1, b in evs:
        a.record(); qp.update(*d); e0.record(); qp.solve(stream); e1.record(); qp.get_device(x=ox, y=oy, status=ost, iters=oit); b.record()
    torch.cuda.synchronize()
    dts = [a.elapsed_time(b) * 1e-3 for a, _, _, b in evs]
    dt = float(np.median(dts))
    kms = [e0.elapsed_time(e1) for _, e0, e1, _ in evs]
    pinfo = qp.plan_info(); ocinfo = qp.oc_info()
    abytes = algorithmic_bytes_per_solve(pinfo["nnzP_triu"], pinfo["nnzA"], ls.n, ls.m)
    k = float(np.median(kms))
    itm = float(oit.float().mean())
    lds_roof, mfma_frac = onchip_rooflines(pinfo, ocinfo, ls.n, ls.m, batch, itm, k)
    setup_ms, iter_ms = qp.last_phase_ms()
    out = {"workload": "%s horizon=%d batch=%d (n=%d m=%d)" % (mdl.name, N, batch, ls.n, ls.m), "value": batch / dt, "unit": "QP solves/s", "ms_per_step": dt * 1e3, "kernel_ms": k,
           "steps": steps, "ms_per_step_all": [x * 1e3 for x in dts], "kernels_ms_last_step": {"setup (initSolver)": setup_ms, "iteration (solve)": iter_ms},
           "mean_admm_iters": itm, "solved_frac": float((ost == 1).float().mean()), "variant": pinfo["variant"], "lds_bytes_per_qp": pinfo["lds_bytes"],
           "roofline": {"bound": "hbm", "achieved": abytes * batch / (k * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": abytes * batch / (k * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_solve": abytes, "traffic": committed_traffic("%s_N%d_b%d_variant%d" % (mdl.name, N, batch, pinfo["variant"])),
                        "lds": lds_roof, "mfma_issue_frac": mfma_frac},
           "workload_gen_s": t_gen}
    qp.close()
    del d, ox, oy
    torch.cuda.empty_cache()
    return out


def sqp_iteration_device(workload, N, batch, seed, dev, reps=5):
    """the complete device-resident SQP iteration on the headline workload (SURVEY.md section 8d's metric with the producer inside the step,
    reference SQPOptimizationSolver.cpp:137-198): mpcqp_stage_eval -> mpcqp_update -> mpcqp_solve -> mpcqp_get -> mpcqp_stage_step / merit,
    from the seeded iterate every time, nothing crossing PCIe"""
    import torch
    from optimal_control_problem_amd import models
    from optimal_control_problem_amd.sqp import DeviceSQPOptimizationSolver
    mdl, ls, meta = models.make_workload(workload, batch, seed=seed, N=N)
    arg = {k: torch.as_tensor(meta[k], dtype=torch.float64, device=dev) for k in ("lbx", "ubx", "lbg", "ubg", "p")}
    x0 = torch.as_tensor(meta["x_iterate"], dtype=torch.float64, device=dev)
    sq = DeviceSQPOptimizationSolver(mdl, {"max_iter": 1, "alpha": 0.5}, batch=batch, device=dev.index)
    sq.qp.set_dispatch_hint(False)
    sq.setInitialGuess(x0); sq.getOptimalSolution(arg, to_host=False); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        sq.setInitialGuess(x0); torch.cuda.synchronize()
        t0 = time.perf_counter(); sq.getOptimalSolution(arg, to_host=False); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    dt = float(np.median(ts))
    out = {"value": batch / dt, "unit": "QP solves/s (one SQP iteration each, evaluation included)", "ms_per_iteration": dt * 1e3, "repetitions": reps, "ms_all": [t * 1e3 for t in ts], "qp_kernel_ms": sq.qp.last_kernel_ms(),
           "mean_admm_iters": float(sq.iters.float().mean()), "solved_frac": float((sq.status == 1).float().mean()),
           "steps": "mpcqp_stage_eval + mpcqp_update + mpcqp_solve + mpcqp_get + mpcqp_stage_step + mpcqp_stage_merit, inputs and iterate resident in HBM, instances in batch order"}
    sq.close()
    torch.cuda.empty_cache()
    return out


def algorithmic_bytes_per_solve(nnzP_triu, nnzA, n, m, warm=False):
    """compulsory traffic with the iteration resident on-chip."""
    b = 8 * (nnzP_triu + nnzA + n + 2 * m) + 8 * (n + m) + 16
    if warm:
        b += 8 * (n + m)
    return b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="quadrotor", choices=["quadrotor", "double_integrator", "cartpole"])
    ap.add_argument("--horizon", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None, help="QP instances per GPU")
    ap.add_argument("--cpu-sample", type=int, default=None, help="QPs timed on the host for cpu_baseline (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="timed region only: skip the in-order, PCIe-inclusive and CPU-baseline legs (profiling runs)")
    # diagnostics (not used by the driver): force a kernel family / an exact ADMM iteration count
    ap.add_argument("--variant", default=None, choices=["stream", "res1", "res2", "res4", "res8", "gres4", "gres2", "oc4", "oc8"])
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal of the N > 1 flow on a one-GPU box: all ranks use cuda:0, collectives over gloo")
    ap.add_argument("--reduced", action="store_true", help="opt-in reduced form (mpcqp_create_reduced): the parameter rows dp = 0 named as fixed; not the headline configuration")
    ap.add_argument("--force-iters", type=int, default=None, help="run exactly this many ADMM iterations (eps = 0, no adaptive rho)")
    args = ap.parse_args()

    if args.variant:
        os.environ["MPCQP_VARIANT"] = args.variant
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, as fresh child processes, before this process has
        # imported torch or touched the GPU (the driver's torch.distributed.run launch sets WORLD_SIZE and skips this)
        from optimal_control_problem_amd import sharding
        raise SystemExit(sharding.launch_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    if int(os.environ.get("WORLD_SIZE", "1")) != max(args.gpus, 1):
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s" % (args.gpus, os.environ.get("WORLD_SIZE")))
    import torch
    from optimal_control_problem_amd import models, sharding
    from optimal_control_problem_amd.batch_qp import BatchQP

    rank, world, local, dist = sharding.init_distributed(args.gpus, backend="gloo" if args.share_gpu else None)
    if args.share_gpu:
        local = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    defaults = {"quadrotor": (20, 8192), "double_integrator": (20, 4096), "cartpole": (100, 16384)}
    N = args.horizon or defaults[args.workload][0]
    batch = args.batch or defaults[args.workload][1]
    seed0 = {"quadrotor": 2024, "double_integrator": 1234, "cartpole": 7}[args.workload]
    t0 = time.time()
    mdl, ls, _ = models.make_workload(args.workload, batch, seed=seed0 + rank, N=N)
    t_gen = time.time() - t0

    # inputs resident in HBM before the timed region
    dP, dq, dA, dl, du = [torch.from_numpy(a).to(dev) for a in (ls.P, ls.q, ls.A, ls.l, ls.u)]
    ox = torch.empty(batch, ls.n, dtype=torch.float64, device=dev)
    oy = torch.empty(batch, ls.m, dtype=torch.float64, device=dev)
    ost = torch.empty(batch, dtype=torch.int32, device=dev)
    oit = torch.empty(batch, dtype=torch.int32, device=dev)
    kw = dict(device=local)
    if args.force_iters:
        kw.update(max_iter=args.force_iters, eps_abs=0.0, eps_rel=0.0, eps_prim_inf=0.0, eps_dual_inf=0.0, adaptive_rho=0)
    if args.reduced:
        kw["fixed_rows"] = list(range(mdl.np))
    qp = BatchQP(ls.n, ls.m, batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai, **kw)
    # `value` is measured with the instances handed to workgroups in batch order.  The longest-first dispatch hint
    # (mpcqp_set_dispatch_hint, on by default in the library) predicts from the previous solve of the same handle; this
    # loop re-solves one batch, which would make that prediction exact, so it is reported separately (with_dispatch_hint).
    qp.set_dispatch_hint(False)
    pinfo = qp.plan_info(); ocinfo = qp.oc_info()
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        qp.update(dP, dq, dA, dl, du)
        qp.solve(stream)
        qp.get_device(x=ox, y=oy, status=ost, iters=oit)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    sharding.barrier(dist)
    torch.cuda.synchronize()
    # kernel duration per step: HIP events on the launch stream, recorded around mpcqp_solve (the solve kernel and its ~20 us validation kernel) and
    # read only after the timed region -- waiting for an event inside it would park the host once per step and put its wake-up and launch
    # latency (0.05 - 0.5 ms, depending on the box) into every step
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for e0, e1 in evs:
        qp.update(dP, dq, dA, dl, du)
        e0.record(); qp.solve(stream); e1.record()
        qp.get_device(x=ox, y=oy, status=ost, iters=oit)
    torch.cuda.synchronize()
    sharding.barrier(dist)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed, dist)
    kernel_ms = [e0.elapsed_time(e1) for e0, e1 in evs]
    kernel_ms_last_lib = qp.last_kernel_ms()         # the library's own events around the solve kernels alone, last step
    setup_ms_last, iter_ms_last = qp.last_phase_ms()  # ... and split by the reference's two calls (on-chip mode: set-up kernel / iteration kernel)

    iters = oit.cpu().numpy(); status = ost.cpu().numpy()
    solved_local = int((status == 1).sum())
    solved = sharding.sum_over_ranks(solved_local, dist)
    iters_sum = sharding.sum_over_ranks(float(iters.sum()), dist)
    kms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
    kms_max = sharding.max_over_ranks(kms, dist)

    # final gather of the solutions (the only collective that touches results; outside the timed region)
    tg = time.perf_counter()
    gathered = sharding.gather_rows(ox, dist, dst=0)
    if dist is not None:
        torch.cuda.synchronize()
    tg = time.perf_counter() - tg
    if dist is not None and rank == 0:
        assert tuple(gathered.shape) == (world * batch, ls.n)
    collective = sharding.collective_record(dist, kms, world * batch * ls.n * 8, tg * 1e3, local, args.share_gpu)

    if rank == 0:
        total = world * batch * args.steps
        value = total / elapsed
        abytes = algorithmic_bytes_per_solve(pinfo["nnzP_triu"], pinfo["nnzA"], ls.n, ls.m)
        achieved = abytes * batch / (kms * 1e-3) / 1e9
        flops_iter = 4 * pinfo["L_blocks"] * 256 + 2 * (2 * pinfo["nnzA"]) + 2 * (2 * pinfo["nnzP_triu"] - ls.n) + 12 * (ls.n + ls.m)
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command on this same build of the library, else null
        traffic = None if args.force_iters else committed_traffic("%s_N%d_b%d_variant%d" % (mdl.name, N, batch, pinfo["variant"]))
        lds_roof, mfma_frac = onchip_rooflines(pinfo, ocinfo, ls.n, ls.m, batch, iters_sum / (world * batch), kms)
        out = {
            "metric": "QP solves/sec (batched OSQP-ADMM, N=%d)" % N, "value": value, "unit": "QP solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s nx=%d nu=%d horizon=%d, reference formulation n=%d m=%d, batch=%d per GPU, "
                                   "eps_abs=eps_rel=1e-3, cold start%s" % (mdl.name, mdl.nx, mdl.nu, N, ls.n, ls.m, batch, ", REDUCED FORM (opt-in: parameter rows eliminated)" if args.reduced else ""),
                       "batch_per_gpu": batch, "parallelism": "batch-sharded x%d, no data-path collective" % world,
                       "dispatch": "batch order (the longest-first hint from the previous solve's iteration counts is off for `value`; see with_dispatch_hint)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         # the same launch priced on its measured HBM traffic instead of the algorithmic bytes
                         "achieved_from_traffic": None if traffic is None else traffic / (kms * 1e-3) / 1e9,
                         "frac_from_traffic": None if traffic is None else traffic / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "kernel": ("mpcqp_oc_setup_kernel + mpcqp_oc_admm_kernel (on-chip mode; one launch each per solve, priced together: `achieved` divides by the sum of their durations)"
                                    if pinfo["variant"] >= 200 and ocinfo["launch_pairs_for_rho_updates"] else "mpcqp_res_kernel (on-chip mode)" if pinfo["variant"] >= 200 else "mpcqp_res_kernel") if pinfo["variant"] else "mpcqp_admm_kernel",
                         "kernels_ms_last_step": {"setup (CuCaQP::initSolver)": setup_ms_last, "iteration (CuCaQP::solve)": iter_ms_last}, "kernel_ms": kms,
                         # the LDS-bandwidth fraction, and the share of a QP's residence spent in the dependent MFMA chains
                         "lds": lds_roof, "mfma_issue_frac": mfma_frac, "kernel_ms_max_over_ranks": kms_max, "kernel_ms_last_step_library_events": kernel_ms_last_lib,
                         "algorithmic_bytes_per_solve": abytes,
                         # flops of one ADMM iteration = two block-triangular solves + A x, A'y + P x + vector work
                         "algorithmic_flops_per_admm_iter": flops_iter,
                         "achieved_tflops_fp64": flops_iter * (iters_sum / world) / (kms * 1e-3) / 1e12,
                         "fp64_vector_peak_tflops": FP64_VEC_PEAK_TFLOPS,
                         "regime": ("factor on chip (LDS + registers, %s); A / A' values re-read from L2 / Infinity Cache / HBM every ADMM iteration" % ("one workgroup of eight waves per CU" if pinfo["variant"] == 208 else "two workgroups per CU") if pinfo["variant"] >= 200 else
                                    "factor blocks and A / A' values re-streamed from HBM every ADMM iteration (occupancy beats LDS residency at this size)"
                                    if pinfo["variant"] >= 100 or pinfo["variant"] == 0 else "factor resident in LDS; A / A' values re-read from L2 / HBM every ADMM iteration")},
            "solve_stats": {"solved_frac": solved / (world * batch), "mean_admm_iters": iters_sum / (world * batch),
                            "admm_iters_per_s": iters_sum / (kms_max * 1e-3),
                            "lds_bytes_per_qp": pinfo["lds_bytes"], "workspace_bytes_per_qp": pinfo["workspace_bytes_per_qp"],
                            "L_blocks": pinfo["L_blocks"], "workload_gen_s": t_gen,
                            "kernel_variant": ("stream (1 wave/QP, Cholesky factor streamed from HBM)" if pinfo["variant"] == 0 else
                                               "on-chip LDL' (%d waves/QP, factor in LDS + registers, solves on the matrix cores)" % (pinfo["variant"] - 200) if pinfo["variant"] >= 200 else
                                               "LDL' %d waves/QP, factor blocks streamed from HBM" % (pinfo["variant"] - 100) if pinfo["variant"] >= 100 else
                                               "resident (%d waves/QP, factor in LDS)" % pinfo["variant"])},
        }
        if collective is not None:
            out["collective"] = collective
        out["lib_sha16"] = lib_sha16()
        if world == 1 and not args.force_iters and not args.no_extras and not args.reduced and (args.workload, N, batch) == ("quadrotor", 20, 8192):
            # the other BASELINE configurations that fit one GPU, one warmed step each, and the complete SQP iteration -- under the driver's
            # clock, outside the timed region of `value`
            try:
                out["other_configs"] = {"config2_double_integrator_N20_b4096": one_config("double_integrator", 20, 4096, 1234, dev),
                                        "config3_quadrotor_N50_b8192": one_config("quadrotor", 50, 8192, 2024, dev),
                                        "config4_cartpole_N100_b16384_cold": one_config("cartpole", 100, 16384, 7, dev)}
            except Exception as e:      # a failure here must not take the headline down
                out["other_configs"] = {"error": repr(e)}
            try:
                out["sqp_iteration_device"] = sqp_iteration_device("quadrotor", 20, 8192, 2024, dev)
            except Exception as e:
                out["sqp_iteration_device"] = {"error": repr(e)}
        if world == 1 and not args.force_iters and not args.no_extras and not args.reduced:
            # the same step with the library's default dispatch hint on: instances ordered longest-first by the previous solve's
            # ADMM iteration counts.  On this repeated batch the prediction is exact, so this is the hint's upper bound.
            qp.set_dispatch_hint(True)
            for _ in range(2):
                step()
            torch.cuda.synchronize(); tn = time.perf_counter()
            for _ in range(5):
                step()
            torch.cuda.synchronize(); tn = (time.perf_counter() - tn) / 5
            out["with_dispatch_hint"] = {"value": batch / tn, "unit": "QP solves/s", "ms_per_step": tn * 1e3,
                                         "note": "mpcqp_set_dispatch_hint(h, 1) (library default): longest-first from the previous solve's iteration counts; exact predictor on a repeated batch"}
            qp.set_dispatch_hint(False)
            # the opt-in reduced form on the same batch (mpcqp_create_reduced: the rows dp = 0 named as fixed -> no parameter block, no arrow in the
            # KKT matrix): an equivalent QP, a different ADMM run -- reported beside `value`, never as `value`
            try:
                qr_ = BatchQP(ls.n, ls.m, batch, ls.Pp, ls.Pi, ls.Ap, ls.Ai, fixed_rows=list(range(mdl.np)), device=local)
                qr_.set_dispatch_hint(False)
                rit = torch.empty(batch, dtype=torch.int32, device=dev)
                def rstep():
                    qr_.update(dP, dq, dA, dl, du); qr_.solve(stream); qr_.get_device(x=ox, y=oy, status=ost, iters=rit)
                for _ in range(2):
                    rstep()
                torch.cuda.synchronize(); tr = time.perf_counter()
                for _ in range(5):
                    rstep()
                torch.cuda.synchronize(); tr = (time.perf_counter() - tr) / 5
                out["reduced_form"] = {"value": batch / tr, "unit": "QP solves/s", "ms_per_step": tr * 1e3, "kernel_ms": qr_.last_kernel_ms(), "variant": qr_.plan_info()["variant"],
                                       "mean_admm_iters": float(rit.float().mean()), "solved_frac": float((ost == 1).float().mean()),
                                       "note": "opt-in: variables fixed by equality singleton rows (the parameter block, dp = 0) substituted before the solve; presolve + solve + postsolve in the step "
                                               "(mpcqp_create_reduced; mpcqp_create_presolved finds such rows from the bounds for callers that cannot name them)"}
                qr_.close()
                step()      # restore ox / oy / ost of the full form for the legs below
                torch.cuda.synchronize()
            except Exception as e:
                out["reduced_form"] = {"error": repr(e)}
            # the same step with the boundary handing over HOST buffers (what a CuCaQP-style caller does: pageable inputs in,
            # x / status / iters out): H2D + kernel + D2H per step.  Reported beside `value`, never as `value`.
            hx = np.empty((batch, ls.n)); hst = np.empty(batch, np.int32); hit = np.empty(batch, np.int32)
            L = __import__("optimal_control_problem_amd._lib", fromlist=["lib"])
            def host_step():
                qp.update(ls.P, ls.q, ls.A, ls.l, ls.u)
                qp.solve(stream)
                L.check(L.lib().mpcqp_get(qp._h, hx.ctypes.data, None, None, hst.ctypes.data, hit.ctypes.data, None, L.MEM_HOST))
            host_step()
            th = time.perf_counter()
            for _ in range(3):
                host_step()
            th = (time.perf_counter() - th) / 3
            in_bytes = 8 * (ls.P.size + ls.q.size + ls.A.size + ls.l.size + ls.u.size); out_bytes = hx.nbytes + hst.nbytes + hit.nbytes
            out["pcie_inclusive"] = {"value": batch / th, "unit": "QP solves/s", "ms_per_step": th * 1e3,
                                     "host_bytes_in": int(in_bytes), "host_bytes_out": int(out_bytes),
                                     "note": "pageable NumPy inputs copied by mpcqp_update(MPCQP_MEM_HOST), x/status/iters copied back"}
            # the fused, pipelined host step (mpcqp_solve_host): pinned buffers, slices of the batch copied while others solve
            pin = [torch.from_numpy(a).pin_memory() for a in (ls.P, ls.q, ls.A, ls.l, ls.u)]
            best = None
            for chunks in (4, 6, 8):
                res = qp.solve_host(*pin, chunks=chunks)
                tp = time.perf_counter()
                for _ in range(3):
                    res = qp.solve_host(*pin, chunks=chunks, out=res)
                tp = (time.perf_counter() - tp) / 3
                if best is None or tp < best[0]:
                    best = (tp, chunks)
            assert np.array_equal(res["x"], hx) and np.array_equal(res["iters"], hit)       # same results as the unpipelined step
            out["pcie_inclusive_pipelined"] = {"value": batch / best[0], "unit": "QP solves/s", "ms_per_step": best[0] * 1e3, "chunks": best[1],
                                               "note": "mpcqp_solve_host: pinned host buffers; slices of the batch are copied in on one stream while earlier slices are solved and copied out on others"}
        if world == 1 and not args.no_cpu_baseline and not args.force_iters and not args.no_extras:
            # the oracle (CPU port of the same algorithm) on this box's host cores, bounded sample of the same workload
            from oracle import oracle as orc
            cores = int(os.environ.get("MPCQP_CPU_THREADS", "0")) or host_cores()
            ns = min(batch, args.cpu_sample or {"quadrotor": 8192, "double_integrator": 4096, "cartpole": 2048}[args.workload])
            pat = orc.Pattern(ls.n, ls.m, ls.Pp, ls.Pi, ls.Ap, ls.Ai)
            st = orc.default_settings()
            pat.solve(ls.P[:64], ls.q[:64], ls.A[:64], ls.l[:64], ls.u[:64], st, nthreads=cores)  # warm the threads
            times = []
            while (sum(times) * cores < 12.0 or len(times) < 10) and len(times) < 50:       # >= 10 repetitions, ~10-30 s of CPU work in total
                t1 = time.perf_counter()
                ref = pat.solve(ls.P[:ns], ls.q[:ns], ls.A[:ns], ls.l[:ns], ls.u[:ns], st, nthreads=cores)
                times.append(time.perf_counter() - t1)
            reps = len(times); tc = float(np.median(times))
            xg = ox[:ns].cpu().numpy()
            fin = np.isfinite(ref["x"])
            out["cpu_baseline"] = {"value": ns / tc, "unit": "QP solves/s", "cores": cores, "kind": "port",
                                   "sample": "first %d QPs of the same batch, OpenMP over instances, sparse LDL' per QP "
                                             "(symbolic analysis shared), median of %d repetitions, %.2f s wall each, logical CPUs on the box %d" % (ns, reps, tc, os.cpu_count() or 0)}
            # one host thread, one QP at a time -- how the reference itself runs (SURVEY.md section 8d baseline (a))
            n1 = min(ns, 512)
            t1 = time.perf_counter(); pat.solve(ls.P[:n1], ls.q[:n1], ls.A[:n1], ls.l[:n1], ls.u[:n1], st, nthreads=1); t1 = time.perf_counter() - t1
            out["cpu_baseline"]["single_thread"] = {"value": n1 / t1, "unit": "QP solves/s", "cores": 1, "sample": "first %d QPs, %.2f s" % (n1, t1)}
            # a real OSQP on this box, if there is one: the reference's true CPU path, one QP at a time with a fresh
            # setup each, and a pin for the oracle.  Absent from this image; the probe result is recorded either way.
            from oracle import osqp_probe
            found = osqp_probe.find()
            out["cpu_baseline"]["osqp_probe"] = {"python_module": found["version"] if found["python"] is not None else None, "libosqp": found["libs"]}
            if found["python"] is not None:
                try:
                    k = min(ns, 256)
                    r = osqp_probe.solve_batch(found["python"], ls.n, ls.m, ls.Pp, ls.Pi, ls.Ap, ls.Ai, ls.P, ls.q, ls.A, ls.l, ls.u, k)
                    fin2 = np.isfinite(ref["x"][:k]).all(axis=1) & np.isfinite(r["x"]).all(axis=1)
                    out["cpu_baseline"]["osqp"] = {"value": k / r["seconds"], "unit": "QP solves/s", "cores": 1, "kind": "osqp",
                                                   "sample": "first %d QPs, fresh osqp setup per QP, settings of SQPOptimizationSolver.cpp:81-85" % k,
                                                   "mean_iters": float(r["iters"].mean()),
                                                   "max_abs_x_diff_vs_oracle": float(np.abs(r["x"][fin2] - ref["x"][:k][fin2]).max()) if fin2.any() else None}
                except Exception as e:        # an unexpected binding version must not take the bench line down
                    out["cpu_baseline"]["osqp"] = {"error": repr(e)}
            out["parity"] = {"max_abs_x_err_vs_oracle": float(np.abs(xg[fin] - ref["x"][fin]).max()),
                             "iters_equal": bool((iters[:ns] == ref["iters"]).all()),
                             "status_equal": bool((status[:ns] == ref["status"]).all())}
        print(json.dumps(out))
    qp.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
