"""One step of a kernel trace, launch by launch: python tools/exp/step_launches.py <dir with *_kernel_trace.csv> [marker kernel] [min us]"""
import csv, glob, sys
d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "adam_kernel"
floor = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
tot = 0.0
for r in rows[a + 1:b + 1]:
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += us
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:52]
    if us >= floor:
        print(f"{n:54s} {us:8.1f} us  grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} wg {r['Workgroup_Size_X']} lds {r['LDS_Block_Size']}")
print(f"sum of kernels {tot:.1f} us, wall {(int(rows[b]['End_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e3:.1f} us")
