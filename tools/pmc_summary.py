#!/usr/bin/env python3
"""Sum one rocprofv3 PMC counter per kernel launch: python tools/pmc_summary.py <dir> <COUNTER> [kernel substring]
(the counter_collection CSV of `rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv -d <dir> -- python3 bench.py ...`)"""
import csv, glob, sys, collections
d, ctr = sys.argv[1], sys.argv[2]
sub = sys.argv[3] if len(sys.argv) > 3 else "mpcqp_res_kernel"
per = collections.defaultdict(float); name = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != ctr or sub not in r.get("Kernel_Name", ""): continue
        k = (f, r["Dispatch_Id"]); per[k] += float(r["Counter_Value"]); name[k] = r["Kernel_Name"]
vals = sorted(per.values())
print(ctr, "launches", len(vals), "mean", sum(vals) / max(len(vals), 1), "min", vals[0] if vals else None, "max", vals[-1] if vals else None, "kernel", sorted(set(name.values()))[:2])
