#!/usr/bin/env python3
"""The BASELINE.json configurations that fit one GPU (plus neighbours), one bench.py line each -> one JSON list.
usage: python tools/config_sweep.py > profiles/rNN_config_sweep.json   (GPU box; `value` is in batch order from round 2 on)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("double_integrator", 20, 4096, "BASELINE configs[1]"), ("quadrotor", 20, 8192, "north star"), ("quadrotor", 50, 8192, "BASELINE configs[2], one shard of configs[4]"),
         ("cartpole", 100, 16384, "BASELINE configs[3] (one QP of the SQP loop, cold start; the warm-started loop is profiles/r01_sqp_device_loop.json)"),
         ("cartpole", 30, 8192, ""), ("cartpole", 50, 8192, ""), ("double_integrator", 60, 8192, ""), ("quadrotor", 10, 8192, ""), ("quadrotor", 100, 2048, "")]
out = []
for w, n, b, note in CASES:
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", w, "--horizon", str(n), "--batch", str(b), "--no-cpu-baseline", "--no-extras",
                        "--steps", "6", "--warmup", "2"], capture_output=True, text=True, timeout=600)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        out.append({"workload": w, "horizon": n, "batch": b, "error": r.stderr[-500:]}); continue
    out.append({"workload": w, "horizon": n, "batch": b, "note": note, "qp_per_s": d["value"], "ms_per_step": d["ms_per_step"], "kernel_ms": d["roofline"].get("kernel_ms"),
                "kernel_variant": d["solve_stats"]["kernel_variant"], "lds_bytes_per_qp": d["solve_stats"]["lds_bytes_per_qp"], "mean_admm_iters": d["solve_stats"]["mean_admm_iters"],
                "solved_frac": d["solve_stats"]["solved_frac"]})
    print("%s N=%d x %d: %.0f QP/s, %.2f ms" % (w, n, b, d["value"], d["ms_per_step"]), file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
