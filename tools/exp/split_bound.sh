#!/bin/bash
# what could ANY bf16-piece arithmetic buy the convolution family?  The two-piece split (fewer matrix
# instructions and less LDS than a three-piece one) is an upper bound: C3 / C2 timed with it on the backward products (the
# shipped opt-in), and -- measurement library only (make -C dynamorph_amd/csrc measure) -- on the forward convolutions too.
#   tools/exp/split_bound.sh OUTDIR      (from the repo root on the GPU box)
out=${1:-gpurun_out/split_bound}
mkdir -p $out
M=$PWD/dynamorph_amd/libdynamorph_hip_measure.so
common="--no-cpu-baseline --no-roofline --no-targets --steps 100"
python3 bench.py $common > $out/c3_f32.json 2> $out/err.log
DM_BACKWARD_PRECISION=split python3 bench.py $common > $out/c3_bwd_split.json 2>> $out/err.log
DM_LIB_PATH=$M python3 bench.py $common > $out/c3_measure_lib_f32.json 2>> $out/err.log
DM_LIB_PATH=$M DM_FORWARD_SPLIT=1 python3 bench.py $common > $out/c3_fwd_split.json 2>> $out/err.log
DM_LIB_PATH=$M DM_FORWARD_SPLIT=1 DM_BACKWARD_PRECISION=split python3 bench.py $common > $out/c3_fwd_bwd_split.json 2>> $out/err.log
python3 bench.py --workload c2 $common > $out/c2_f32.json 2>> $out/err.log
DM_LIB_PATH=$M DM_FORWARD_SPLIT=1 python3 bench.py --workload c2 $common > $out/c2_fwd_split.json 2>> $out/err.log
for f in c3_f32 c3_bwd_split c3_measure_lib_f32 c3_fwd_split c3_fwd_bwd_split c2_f32 c2_fwd_split; do
  python3 - $out/$f.json $f <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-22s %8.4f ms/step  %10.0f patches/s  losses %s" % (sys.argv[2], r["ms_per_step"], r["value"], r.get("final_losses")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done | tee $out/table.txt
