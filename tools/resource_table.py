"""One line per kernel from the compiler's -Rpass-analysis=kernel-resource-usage remarks (csrc/Makefile `resource`)."""
import re
import subprocess
import sys


def rows(path):
    txt = open(path).read()
    for blk in re.split(r"remark: Function Name: ", txt)[1:]:
        name = blk.split()[0]
        g = lambda k: int(re.search(k + r": (\d+)", blk).group(1))
        try:
            name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0] or name
        except OSError:
            pass
        yield dict(kernel=name.replace("void ", ""), vgpr=g("VGPRs"), agpr=g("AGPRs"), scratch=g(r"ScratchSize \[bytes/lane\]"),
                   sgpr_spill=g("SGPRs Spill"), vgpr_spill=g("VGPRs Spill"), occupancy=g(r"Occupancy \[waves/SIMD\]"))


if __name__ == "__main__":
    print("%-70s %5s %5s %8s %11s %11s %4s" % ("kernel", "VGPR", "AGPR", "scratch", "SGPR spill", "VGPR spill", "occ"))
    for p in sys.argv[1:]:
        for r in rows(p):
            print("%-70s %5d %5d %8d %11d %11d %4d" % (r["kernel"], r["vgpr"], r["agpr"], r["scratch"], r["sgpr_spill"], r["vgpr_spill"], r["occupancy"]))
