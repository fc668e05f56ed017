#!/usr/bin/env python3
"""Average of every counter per kernel from `rocprofv3 --pmc ... --output-format csv` directories.
usage: pmc_dump.py name-substring DIR [DIR ...]"""
import csv, glob, re, sys
from collections import defaultdict
want = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if want not in n:
                continue
            n = re.sub(r"\(anonymous namespace\)::", "", n).replace("void ", "")
            n = n[:n.index(">(") + 1] if ">(" in n else n.split("(")[0]
            agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in agg.items():
    print(n)
    for c, v in sorted(cs.items()):
        print(f"    {c:36s} {sum(v) / len(v):16.0f}   (n={len(v)}, max {max(v):.0f})")
