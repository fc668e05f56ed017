#!/usr/bin/env python3
"""Per-kernel summary of two `rocprofv3 --pmc` SQ passes (see HISTORY.md section 4): instructions per wave, share of
wave time spent issuing / parked / stalled, MFMA pipe share.  usage: sq_summary.py gpurun_out/sq1 gpurun_out/sq2 [out.json]
With a third argument the MFMA-busy share and the instruction counts per kernel are also written as JSON (bench.py quotes
them next to its own timings: it cannot run under the profiler itself)."""
import csv, glob, re, sys
from collections import defaultdict


def load(d):
    agg = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "anonymous namespace" not in n:
                continue
            n = re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0].replace("void ", "")
            agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] in ("SQ_WAVES", "SQ_INSTS_VALU"):
                dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}, {k: sum(v) / len(v) for k, v in dur.items()}


a, dur = load(sys.argv[1]); b, _ = load(sys.argv[2])
rows = []
for k in a:
    if k not in b:
        continue
    A, Bc = a[k], b[k]
    waves = A["SQ_WAVES"]; wc = A["SQ_WAVE_CYCLES"]
    rows.append((dur.get(k, 0), k, waves, Bc["SQ_INSTS_VALU"] / waves, Bc["SQ_INSTS_SALU"] / waves, Bc["SQ_INSTS_LDS"] / waves,
                 (Bc["SQ_INSTS_VMEM_RD"] + Bc["SQ_INSTS_VMEM_WR"]) / waves, 4 * wc / waves,
                 100 * A["SQ_ACTIVE_INST_ANY"] / wc, 100 * A["SQ_WAIT_ANY"] / wc, 100 * A["SQ_WAIT_INST_ANY"] / wc,
                 Bc["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (4 * wc / waves) * 100 if wc else 0))
print(f"{'kernel':58s} {'us':>6s} {'waves':>6s} {'valu/w':>8s} {'salu/w':>7s} {'lds/w':>7s} {'vmem/w':>6s} {'cyc/w':>9s} {'act%':>5s} {'wait%':>5s} {'stall%':>6s} {'mfma%':>5s}")
for r in sorted(rows, reverse=True):
    print(f"{r[1][:58]:58s} {r[0]:6.0f} {r[2]:6.0f} {r[3]:8.0f} {r[4]:7.0f} {r[5]:7.0f} {r[6]:6.0f} {r[7]:9.0f} {r[8]:5.1f} {r[9]:5.1f} {r[10]:6.1f} {r[11]:5.1f}")
if len(sys.argv) > 3:
    import json
    out = {r[1]: {"us_under_profiler": round(r[0], 1), "waves": int(r[2]), "valu_per_wave": round(r[3]), "mfma_busy_pct_of_wave_cycles": round(r[11], 1),
                  "issue_active_pct": round(r[8], 1), "wait_pct": round(r[9], 1), "stall_pct": round(r[10], 1)} for r in rows}
    json.dump({"note": "two rocprofv3 --pmc passes (SQ counters only) over bench.py --steps 3; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES per SIMD / cycles a wave is resident",
               "kernels": out}, open(sys.argv[3], "w"), indent=1)
