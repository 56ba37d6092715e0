#!/bin/bash
# One hash per kernel instance of the on-chip mode's two kernels (and of the single-kernel form kept for A/B), from the device assembly hipcc
# writes for the k_oc_*.hip units (comments and labels stripped): "the ISA of instance X did not change" as a file that can be diffed
# (profiles/r04_isa_hashes.txt) instead of a sentence.  usage: bash tools/isa_guard.sh [unit ...]   (default: k_oc_setup k_oc_admm k_oc_admm_rf)
set -o pipefail
root=$(cd "$(dirname "$0")/.." && pwd); src=$root/optimal_control_problem_amd/csrc; tmp=$(mktemp -d)
units=("$@"); [ ${#units[@]} -eq 0 ] && units=(k_oc_setup k_oc_admm k_oc_admm_rf)
for u in "${units[@]}"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 --cuda-device-only -S -o $tmp/$u.s $src/$u.hip 2>/dev/null &
done
wait
for u in "${units[@]}"; do
  python3 - "$tmp/$u.s" <<'PY'
import hashlib, re, subprocess, sys
name, body, out = None, [], []
for line in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        name, body = m.group(1), []
        continue
    if name is None:
        continue
    t = line.split(";")[0].strip()
    if t and not t.startswith("."):
        body.append(t)
    if t.startswith(".Lfunc_end"):
        try:
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "") or name
        except OSError:
            dem = name
        out.append("%s  %6d instructions  %s" % (hashlib.sha256("\n".join(body).encode()).hexdigest()[:16], len(body), dem))
        name = None
print("\n".join(out))
PY
done
rm -rf $tmp
