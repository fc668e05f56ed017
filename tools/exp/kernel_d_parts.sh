#!/bin/bash
# Experiment: where kernel D's time goes -- the step with parts of bwd_s2_split_kernel switched off (measurement library only;
# the results of such a step are wrong, its time is what is read).  DM_FUSED_BWD_DBG: 1 no weight-gradient products, 2 no
# data-gradient products, 4 no loads / commits after the first tile.   usage: tools/exp/kernel_d_parts.sh OUTDIR
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/${1:-gpurun_out/kd_parts}; mkdir -p $out
export DM_LIB_PATH=$root/dynamorph_amd/libdynamorph_hip_measure.so TMPDIR=/tmp
for dbg in 0 1 2 4 3 7; do
  cd /tmp
  DM_FUSED_BWD_DBG=$dbg rocprofv3 --kernel-trace --stats --output-format csv -d $out/p$dbg -- python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-targets --no-roofline > /dev/null 2> $out/p$dbg.err
  cd $root
  echo "dbg=$dbg $(python3 tools/kstats_summary.py $out/p$dbg 2>/dev/null | grep bwd_s2_split)"
  rm -rf $out/p$dbg
done > $out/table.txt
cat $out/table.txt
