#!/usr/bin/env python3
"""Readable per-kernel table from a `rocprofv3 --kernel-trace --stats --output-format csv` directory.
usage: kstats_summary.py DIR [steps]   (steps: divide total time by it to get ms per step)"""
import csv, glob, re, sys
d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:60]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).replace("void ", "")
    n = n[:n.index(">(") + 1] if ">(" in n else n.split("(")[0]
    print("%-64s calls %5d avg %8.1f us  %5.2f%%" % (n[:64], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
print("total %.3f ms over all calls" % (tot / 1e6) + (", %.4f ms per step" % (tot / 1e6 / steps) if steps else ""))
