"""Median duration of the MFMA VQ kernel per shape from a kernel trace of `vqbench.py A B C` (launch order = shape order)."""
import csv, glob, statistics as st, sys
d, nshape = sys.argv[1], int(sys.argv[2])
v = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "vq_forward_mfma" in r["Kernel_Name"]:
            v.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
v.sort()
dur = [x[1] for x in v]
n = len(dur) // nshape
print([f"{st.median(dur[i * n + 4:(i + 1) * n]):.1f}" for i in range(nshape)])
