#!/usr/bin/env python3
"""Per (kernel, grid size) summary of a `rocprofv3 --kernel-trace --output-format csv` directory:
count, average / min / max duration in us.  usage: ktrace_summary.py DIR [name-substring ...]"""
import csv, glob, re, sys
from collections import defaultdict

d = sys.argv[1]
want = sys.argv[2:]
rows = defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        n = re.sub(r"\(anonymous namespace\)::", "", n).replace("void ", "")
        n = re.sub(r"\(.*", "", n) if "<" not in n.split("(")[0] else n[:n.index(">(") + 1] if ">(" in n else n
        if want and not any(w in n for w in want):
            continue
        g = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)))
        rows[(n, g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"{'kernel':70s} {'grid':>9s} {'n':>5s} {'avg us':>9s} {'min us':>9s} {'max us':>9s}")
for (n, g), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n[:70]:70s} {g:9d} {len(v):5d} {sum(v) / len(v):9.1f} {min(v):9.1f} {max(v):9.1f}")
