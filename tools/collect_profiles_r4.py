#!/usr/bin/env python3
"""Turn the output of tools/final_measure_r4.sh (gpurun_out/final4/) into the committed profiles/r04_* files, every one of them carrying the hash of
the library it was taken on (lib_sha16 = sha256(libmpcqp.so)[:16], the key bench.py prints), and refresh profiles/traffic_table.json.
usage: python tools/collect_profiles_r4.py"""
import csv
import glob
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "final4"); DST = os.path.join(ROOT, "profiles")
sha_tree = hashlib.sha256(open(os.path.join(ROOT, "optimal_control_problem_amd", "libmpcqp.so"), "rb").read()).hexdigest()[:16]
bench = json.loads(open(os.path.join(SRC, "bench.json")).read().strip().splitlines()[-1])
sha = bench.get("lib_sha16")
if sha != sha_tree:
    print("warning: bench.json was produced by library %s, the tree holds %s" % (sha, sha_tree), file=sys.stderr)
hdr = "# lib_sha16 %s (sha256(optimal_control_problem_amd/libmpcqp.so)[:16] of the build these numbers were taken on)\n" % sha
json.dump(bench, open(os.path.join(DST, "r04_final_bench.json"), "w"), indent=1)
WL = {"q20": ("quadrotor", 20, 8192, 204), "q50": ("quadrotor", 50, 8192, 208), "cp100": ("cartpole", 100, 16384, 208)}
other = bench.get("other_configs", {})
ALG = {"q20": bench["roofline"]["algorithmic_bytes_per_solve"],
       "q50": other.get("config3_quadrotor_N50_b8192", {}).get("roofline", {}).get("algorithmic_bytes_per_solve"),
       "cp100": other.get("config4_cartpole_N100_b16384_cold", {}).get("roofline", {}).get("algorithmic_bytes_per_solve")}
summary = {"lib_sha16": sha,
           "_doc": "final build of round 4: rocprofv3 --kernel-trace --pmc <counter> --output-format csv -- python3 bench.py <workload> --steps 3 --warmup 2 --no-extras --no-cpu-baseline, one "
                   "counter per run (tools/final_measure_r4.sh 1); a solve of the on-chip mode is two kernels (mpcqp_oc_setup_kernel, mpcqp_oc_admm_kernel) plus the launches that serve "
                   "adaptive-rho steps: every figure is the total over ALL of them per solve (tools/pmc_summary.py <dir> <counter> mpcqp_oc_ 5), by_kernel says whose it is; read bytes = "
                   "2 x FETCH_SIZE x 1024 (profiles/r02_traffic_counter_calibration.json), written = WRITE_SIZE x 1024; kernels_ms_rocprof = rocprofv3 --kernel-trace --stats averages of "
                   "the same command with --steps 6 (the first, cold launch included)"}
tbl_path = os.path.join(DST, "traffic_table.json"); tbl = json.load(open(tbl_path))
for tag, (name, N, B, variant) in WL.items():
    vals, byk = {}, {}
    for line in open(os.path.join(SRC, "pmc_%s.txt" % tag)):
        m = re.match(r"(\S+) solves (\d+) per_solve (\S+) launches (\d+) by_kernel (.*)$", line)
        if m:
            vals[m.group(1)] = float(m.group(3)); byk[m.group(1)] = eval(m.group(5))
    ks = max(glob.glob(os.path.join(SRC, "kstats_%s" % tag, "*", "*kernel_stats.csv")), key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(ks))]
    with open(os.path.join(DST, "r04_final_%s_kernel_stats.csv" % tag), "w") as f:
        f.write(hdr); f.write(open(ks).read())
    kern = {}
    for r in rows:
        if "mpcqp_oc_" in r["Name"]:
            short = r["Name"].replace("void ", "").split("(")[0]
            kern[short] = {"calls": int(r["Calls"]), "average_ms": float(r["AverageNs"]) / 1e6, "min_ms": float(r["MinNs"]) / 1e6, "max_ms": float(r["MaxNs"]) / 1e6, "share_of_gpu_time_pct": float(r["Percentage"])}
    hbm = 2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024
    setup_ms = sum(v["average_ms"] * v["calls"] for k, v in kern.items() if "setup" in k) / 8.0      # 8 solves (6 steps + 2 warm-up); the resume-mode launches return at once
    admm_ms = sum(v["average_ms"] * v["calls"] for k, v in kern.items() if "admm" in k) / 8.0
    e = {"workload": "%s N=%d x %d" % (name, N, B), "solves_per_pmc_run": 5, "FETCH_SIZE_KiB_per_solve": vals["FETCH_SIZE"], "WRITE_SIZE_KiB_per_solve": vals["WRITE_SIZE"],
         "hbm_bytes_per_solve": hbm, "read_bytes_per_qp": 2 * vals["FETCH_SIZE"] * 1024 / B, "written_bytes_per_qp": vals["WRITE_SIZE"] * 1024 / B,
         "by_kernel_KiB_per_solve": {c: {k: v["per_solve"] for k, v in byk[c].items()} for c in ("FETCH_SIZE", "WRITE_SIZE")},
         "kernels_rocprof": kern, "setup_kernel_ms_per_solve": setup_ms, "iteration_kernel_ms_per_solve": admm_ms, "kernels_ms_per_solve": setup_ms + admm_ms}
    if ALG.get(tag):
        e["algorithmic_bytes_per_solve_batch"] = ALG[tag] * B; e["traffic_over_algorithmic"] = hbm / (ALG[tag] * B)
        e["achieved_GBps_on_traffic"] = hbm / ((setup_ms + admm_ms) * 1e-3) / 1e9
        e["roofline_frac_algorithmic"] = ALG[tag] * B / ((setup_ms + admm_ms) * 1e-3) / 8e12
    if "TCC_HIT_sum" in vals:
        e["TCC_HIT_per_solve"] = vals["TCC_HIT_sum"]; e["TCC_MISS_per_solve"] = vals["TCC_MISS_sum"]; e["L2_hit_rate"] = vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"])
    summary[tag] = e
    tbl["%s_N%d_b%d_variant%d" % (name, N, B, variant)] = {"hbm_bytes_per_launch": hbm, "lib_sha16": sha, "source": "profiles/r04_final_pmc_summary.json (%s): set-up + iteration kernels of one solve" % tag}
json.dump(summary, open(os.path.join(DST, "r04_final_pmc_summary.json"), "w"), indent=1)
json.dump(tbl, open(tbl_path, "w"), indent=1)


def keyed_text(src, dst):
    p = os.path.join(SRC, src)
    if os.path.exists(p) and os.path.getsize(p) > 2:
        open(os.path.join(DST, dst), "w").writelines([hdr] + [l for l in open(p) if "amdgpu.ids" not in l])


for tag in WL:
    keyed_text("timing_breakdown_%s.txt" % tag, "r04_final_timing_breakdown_%s.txt" % tag)
for a, b in (("fuzz_oc4.txt", "r04_fuzz_oc4.txt"), ("fuzz_oc8.txt", "r04_fuzz_oc8.txt"), ("fuzz_gpu.txt", "r04_fuzz_gpu.txt")):
    keyed_text(a, b)
# wave-cycle accounting from the counter groups (tools/pmc_groups.sh): totals per solve over both kernels
acc = {"lib_sha16": sha, "_doc": "rocprofv3 counter groups, one run each (tools/pmc_groups.sh <tag> ...; bench.py --steps 2 --warmup 1): totals per solve over the set-up and the iteration kernel"}
for tag in WL:
    p = os.path.join(SRC, "pmcg_%s.txt" % tag)
    if not os.path.exists(p):
        continue
    v = {}
    for line in open(p):
        m = re.match(r"(\S+) solves (\d+) per_solve (\S+) ", line)
        if m:
            v[m.group(1)] = float(m.group(3))
    d = {}
    if v.get("SQ_WAVE_CYCLES"):
        d["wave_cycles_waiting_on_waitcnt_frac"] = v.get("SQ_WAIT_INST_ANY", 0) / v["SQ_WAVE_CYCLES"]
        d["wave_cycles_issuing_frac"] = v.get("SQ_ACTIVE_INST_ANY", 0) / v["SQ_WAVE_CYCLES"]
        d["wave_cycles_neither_frac"] = 1 - d["wave_cycles_waiting_on_waitcnt_frac"] - d["wave_cycles_issuing_frac"]
    if v.get("TCP_TCC_READ_REQ_sum"):
        d["L1_to_L2_read_latency_cycles"] = v.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / v["TCP_TCC_READ_REQ_sum"]
    if v.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac_of_lds_active"] = v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"]
    v["derived"] = d; acc[tag] = v
json.dump(acc, open(os.path.join(DST, "r04_wave_cycle_accounting.json"), "w"), indent=1)
for a, b in (("bench_gpus2.json", "r04_bench_gpus2_share_gpu_rehearsal.json"), ("sqp_device_loop.json", "r04_sqp_device_loop.json")):
    p = os.path.join(SRC, a)
    if os.path.exists(p) and os.path.getsize(p) > 2:
        txt = open(p).read().strip()
        try:
            cand = [l[l.index("{"):] for l in txt.splitlines() if "{" in l]
            obj = json.loads(cand[-1]); obj.setdefault("lib_sha16", sha)
            json.dump(obj, open(os.path.join(DST, b), "w"), indent=1)
        except ValueError:
            print("skipped", a, file=sys.stderr)
# compiler's own account of the kernels, and their ISA hashes (CPU side: reproducible here)
res = subprocess.run(["make", "-C", os.path.join(ROOT, "optimal_control_problem_amd", "csrc"), "resource"], capture_output=True, text=True).stdout
open(os.path.join(DST, "r04_kernel_resources.txt"), "w").write(hdr + "# make -C optimal_control_problem_amd/csrc resource (hipcc -Rpass-analysis=kernel-resource-usage)\n" + "\n".join(l for l in res.splitlines() if "mpcqp_oc_" in l or l.startswith("kernel")) + "\n")
isa = subprocess.run(["bash", os.path.join(ROOT, "tools", "isa_guard.sh")], capture_output=True, text=True).stdout
open(os.path.join(DST, "r04_isa_hashes.txt"), "w").write(hdr + "# bash tools/isa_guard.sh: sha256[:16] of each kernel instance's device assembly (comments and labels stripped)\n" + isa)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk in ("hbm_bytes_per_solve", "traffic_over_algorithmic", "kernels_ms_per_solve", "L2_hit_rate", "roofline_frac_algorithmic")} for k, v in summary.items() if isinstance(v, dict)}, indent=1))
