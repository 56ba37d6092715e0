#!/usr/bin/env python3
"""Turn the output of tools/final_measure_r3.sh (gpurun_out/final3/) into the committed profiles/r03_final_* files and refresh
profiles/traffic_table.json with entries keyed by the hash of the library the passes were taken on.   usage: python tools/collect_profiles_r3.py"""
import csv
import glob
import hashlib
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "final3"); DST = os.path.join(ROOT, "profiles")
sha = hashlib.sha256(open(os.path.join(ROOT, "optimal_control_problem_amd", "libmpcqp.so"), "rb").read()).hexdigest()[:16]
bench = json.loads(open(os.path.join(SRC, "bench.json")).read().strip().splitlines()[-1])
if bench.get("lib_sha16") != sha:
    print("warning: bench.json was produced by library %s, the tree holds %s" % (bench.get("lib_sha16"), sha), file=sys.stderr)
sha = bench.get("lib_sha16") or sha
json.dump(bench, open(os.path.join(DST, "r03_final_bench.json"), "w"), indent=1)
WL = {"q20": ("quadrotor", 20, 8192), "q50": ("quadrotor", 50, 8192), "cp100": ("cartpole", 100, 16384)}
other = bench.get("other_configs", {})
ALG = {"q20": bench["roofline"]["algorithmic_bytes_per_solve"],
       "q50": other.get("config3_quadrotor_N50_b8192", {}).get("roofline", {}).get("algorithmic_bytes_per_solve"),
       "cp100": other.get("config4_cartpole_N100_b16384_cold", {}).get("roofline", {}).get("algorithmic_bytes_per_solve")}
summary = {"_doc": "final build of round 3 (libmpcqp.so sha256[:16] = %s): rocprofv3 --kernel-trace --pmc <counter> --output-format csv -- python3 bench.py <workload> --steps 3 --warmup 2 "
                   "--no-extras --no-cpu-baseline, one counter per run (tools/final_measure_r3.sh, summed per launch by tools/pmc_summary.py); read bytes = 2 x FETCH_SIZE x 1024 "
                   "(profiles/r02_traffic_counter_calibration.json), written = WRITE_SIZE x 1024; kernel_ms_rocprof = rocprofv3 --kernel-trace --stats average of the same command with --steps 6" % sha}
tbl_path = os.path.join(DST, "traffic_table.json"); tbl = json.load(open(tbl_path))
for tag, (name, N, B) in WL.items():
    vals = {}
    for line in open(os.path.join(SRC, "pmc_%s.txt" % tag)):
        m = re.match(r"(\S+) launches (\d+) mean (\S+) .* kernel \['void (mpcqp_res_kernel<[^>]*>)", line)
        if m:
            vals[m.group(1)] = float(m.group(3)); kernel = m.group(4); launches = int(m.group(2))
    ks = max(glob.glob(os.path.join(SRC, "kstats_%s" % tag, "*", "*kernel_stats.csv")), key=os.path.getmtime)      # (the newest pass)
    shutil.copy(ks, os.path.join(DST, "r03_final_%s_kernel_stats.csv" % tag))
    row = [r for r in csv.DictReader(open(ks)) if "mpcqp_res_kernel" in r["Name"]][0]
    hbm = 2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024
    e = {"workload": "%s N=%d x %d" % (name, N, B), "kernel": kernel, "launches": launches, "FETCH_SIZE_KiB_per_launch": vals["FETCH_SIZE"], "WRITE_SIZE_KiB_per_launch": vals["WRITE_SIZE"],
         "hbm_bytes_per_launch": hbm, "read_bytes_per_qp": 2 * vals["FETCH_SIZE"] * 1024 / B, "written_bytes_per_qp": vals["WRITE_SIZE"] * 1024 / B,
         "kernel_ms_rocprof": float(row["AverageNs"]) / 1e6, "kernel_calls_rocprof": int(row["Calls"]), "kernel_share_of_gpu_time_pct": float(row["Percentage"])}
    if ALG.get(tag):
        e["algorithmic_bytes_per_launch"] = ALG[tag] * B; e["traffic_over_algorithmic"] = hbm / (ALG[tag] * B)
        e["achieved_GBps_on_traffic"] = hbm / (e["kernel_ms_rocprof"] * 1e-3) / 1e9
    if "TCC_HIT_sum" in vals:
        e["TCC_HIT_per_launch"] = vals["TCC_HIT_sum"]; e["TCC_MISS_per_launch"] = vals["TCC_MISS_sum"]; e["L2_hit_rate"] = vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"])
    summary[tag] = e
    variant = {"q20": 204, "q50": 208, "cp100": 208}[tag]
    tbl["%s_N%d_b%d_variant%d" % (name, N, B, variant)] = {"hbm_bytes_per_launch": hbm, "lib_sha16": sha, "source": "profiles/r03_final_pmc_summary.json (%s)" % tag}
json.dump(summary, open(os.path.join(DST, "r03_final_pmc_summary.json"), "w"), indent=1)
json.dump(tbl, open(tbl_path, "w"), indent=1)
for tag in WL:
    txt = [l for l in open(os.path.join(SRC, "timing_breakdown_%s.txt" % tag)) if "amdgpu.ids" not in l]
    open(os.path.join(DST, "r03_final_timing_breakdown_%s.txt" % tag), "w").writelines(txt)
for a, b in (("bench_gpus2.json", "r03_bench_gpus2_share_gpu_rehearsal.json"), ("sqp_device_loop.json", "r03_sqp_device_loop.json"), ("config_sweep.json", "r03_config_sweep.json")):
    p = os.path.join(SRC, a)
    if os.path.exists(p) and os.path.getsize(p) > 2:
        txt = open(p).read().strip()
        try:
            cand = [l[l.index("{"):] for l in txt.splitlines() if "{" in l]
            json.dump(json.loads(txt) if txt.startswith("[\n") else json.loads(cand[-1]), open(os.path.join(DST, b), "w"), indent=1)
        except ValueError:
            print("skipped", a, file=sys.stderr)
for a, b in (("stageqp_bench.txt", "r03_stageqp_bench.txt"), ("chain_stage_probe.txt", "r03_chain_stage_probe.txt"), ("host_pipeline.txt", "r03_host_pipeline.txt")):
    p = os.path.join(SRC, a)
    if os.path.exists(p) and os.path.getsize(p) > 2:
        open(os.path.join(DST, b), "w").writelines(l for l in open(p) if "amdgpu.ids" not in l)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk in ("hbm_bytes_per_launch", "traffic_over_algorithmic", "kernel_ms_rocprof", "L2_hit_rate")} for k, v in summary.items() if k != "_doc"}, indent=1))
