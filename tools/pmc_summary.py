#!/usr/bin/env python3
"""Sum one rocprofv3 PMC counter over the kernels of a solve: python tools/pmc_summary.py <dir> <COUNTER> [kernel substring] [solves]
(the counter_collection CSV of `rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv -d <dir> -- python3 bench.py ...`).
Without `solves`: mean per launch of the matching kernels (one kernel per solve: the single-kernel families).  With `solves` (= steps + warm-up of the
bench command): the total over ALL matching launches divided by the number of solves -- the two-kernel on-chip mode runs a set-up and an iteration
kernel per solve plus the launches that serve adaptive-rho steps -- and the share of each kernel name."""
import collections
import csv
import glob
import sys

d, ctr = sys.argv[1], sys.argv[2]
sub = sys.argv[3] if len(sys.argv) > 3 else "mpcqp_"
solves = int(sys.argv[4]) if len(sys.argv) > 4 else 0
per = collections.defaultdict(float); name = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != ctr or sub not in r.get("Kernel_Name", ""):
            continue
        k = (f, r["Dispatch_Id"]); per[k] += float(r["Counter_Value"]); name[k] = r["Kernel_Name"]
vals = sorted(per.values())
if not solves:
    print(ctr, "launches", len(vals), "mean", sum(vals) / max(len(vals), 1), "min", vals[0] if vals else None, "max", vals[-1] if vals else None, "kernel", sorted(set(name.values()))[:2])
else:
    by = collections.defaultdict(float); cnt = collections.Counter()
    for k, v in per.items():
        short = name[k].replace("void ", "").split("(")[0]
        by[short] += v; cnt[short] += 1
    tot = sum(by.values())
    print(ctr, "solves", solves, "per_solve", tot / solves, "launches", len(vals), "by_kernel", {k: {"per_solve": v / solves, "launches": cnt[k]} for k, v in sorted(by.items())})
