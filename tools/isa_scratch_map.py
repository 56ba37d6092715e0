"""Where a kernel's scratch traffic sits: per basic block of the device assembly (hipcc -S --cuda-device-only), the number of scratch loads / stores,
MFMA instructions (v_mfma_f64_4x4x4: the solve's chains and hub phases; 16x16x4: the factorisation) and global loads.  A block that has both
4x4x4 MFMAs or the sweeps' global loads AND scratch loads is a reload inside the ADMM iteration -- the thing to look for.
usage: isa_scratch_map.py file.s [kernel-substring]"""
import re
import sys

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None
blocks = {}
order = []
kern = None
for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        kern = m.group(1)
        continue
    if kern is None or want not in kern:
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", line)
    if m:
        cur = (kern, m.group(1))
        blocks[cur] = dict(sl=0, ss=0, m4=0, m16=0, gl=0, gs=0, ds=0, n=0, bar=0)
        order.append(cur)
        continue
    if cur is None or cur[0] != kern:
        cur = (kern, "entry")
        if cur not in blocks:
            blocks[cur] = dict(sl=0, ss=0, m4=0, m16=0, gl=0, gs=0, ds=0, n=0, bar=0)
            order.append(cur)
    t = line.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    b = blocks[cur]
    b["n"] += 1
    if t.startswith("scratch_load"): b["sl"] += 1
    elif t.startswith("scratch_store"): b["ss"] += 1
    elif t.startswith("v_mfma_f64_4x4x4"): b["m4"] += 1
    elif t.startswith("v_mfma_f64_16x16x4"): b["m16"] += 1
    elif t.startswith("global_load"): b["gl"] += 1
    elif t.startswith("global_store"): b["gs"] += 1
    elif t.startswith("ds_"): b["ds"] += 1
    elif t.startswith("s_barrier"): b["bar"] += 1
    if t.startswith("s_endpgm"):
        cur = None
last = None
for k in order:
    b = blocks[k]
    if k[0] != last:
        print("==", k[0]); last = k[0]
    if b["sl"] or b["ss"] or b["m4"] or b["m16"]:
        print("%-12s instr %5d  scratch ld %3d st %3d | mfma4 %3d mfma16 %3d | global ld %3d st %3d | ds %3d bar %d" % (k[1], b["n"], b["sl"], b["ss"], b["m4"], b["m16"], b["gl"], b["gs"], b["ds"], b["bar"]))
