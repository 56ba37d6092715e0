"""Scratch accesses per phase of a kernel: the device assembly of a build with -DMPCQP_ASM_MARKS carries the phase boundaries (the TS(k) stamps) as
comments; this counts scratch loads / stores between consecutive marks, in program order.
usage: isa_phase_scratch.py file.s"""
import re
import sys

kern = None
seg = "start"
acc = {}
order = []
for line in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        kern = m.group(1); seg = "start"
        continue
    if kern is None:
        continue
    t = line.strip()
    m = re.match(r"; TS (\d+)", t)
    if m:
        seg = "after TS%s" % m.group(1)
        continue
    key = (kern, seg)
    if key not in acc:
        acc[key] = [0, 0, 0]; order.append(key)
    if t.startswith("scratch_load"): acc[key][0] += 1
    elif t.startswith("scratch_store"): acc[key][1] += 1
    elif t and not t.startswith(";") and not t.startswith("."): acc[key][2] += 1
    if t.startswith("s_endpgm"):
        kern = None
last = None
for k in order:
    if k[0] != last:
        print("==", k[0]); last = k[0]
    print("  %-12s instr %6d  scratch loads %4d stores %4d" % (k[1], acc[k][2], acc[k][0], acc[k][1]))
