#!/usr/bin/env python3
"""Timeline of ONE steady-state training step from a `rocprofv3 --kernel-trace --output-format csv` directory:
every dispatch of the step in order with its duration and the idle gap before it, and the totals (busy / gaps).
A step is found as the span between two consecutive dispatches of the marker kernel (default: adam_kernel).
usage: step_timeline.py DIR [marker] [which]   (which: index of the step from the end, default 3)"""
import csv, glob, re, sys

d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "adam_kernel"
which = int(sys.argv[3]) if len(sys.argv) > 3 else 3
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n).replace("void ", "")
    return (n[:n.index(">(") + 1] if ">(" in n else n.split("(")[0])[:60]


marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = marks[-which - 1], marks[-which]
step = rows[a + 1:b + 1]
busy = gaps = 0
prev_end = int(rows[a]["End_Timestamp"])
small = 0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - prev_end
    busy += e - s
    gaps += max(gap, 0)
    if e - s < 20000:
        small += e - s
    print("%-60s %8.1f us   gap %6.1f us" % (short(r["Kernel_Name"]), (e - s) / 1e3, gap / 1e3))
    prev_end = max(prev_end, e)
span = int(step[-1]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])
print("dispatches %d  span %.1f us  busy %.1f us  gaps %.1f us  (kernels under 20 us: %.1f us)" % (len(step), span / 1e3, busy / 1e3, gaps / 1e3, small / 1e3))
