#!/bin/bash
# Experiment: where the decoder tail's time goes -- dec_tail_backward_kernel<2, true, false> with parts switched off (measurement
# library only; the results of such a launch are wrong, its time is what is read).  DM_DEC_TAIL_DBG bits: 1 no phase A products
# (d4 recompute), 2 no phase B (g4, loss, dW6 sums), 4 no phase 3 (data gradient), 8 no phase 4 (weight-gradient matrix
# instructions), 16 no loads / commits after the first tile, 32 no workgroup barriers in the tile loop.
#   usage: tools/exp/dec_tail_parts.sh OUTFILE
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/${1:-gpurun_out/dec_tail_parts.txt}
export DM_LIB_PATH=$root/dynamorph_amd/libdynamorph_hip_measure.so
: > $out
for dbg in 0 1 2 4 8 16 32 15 31 63 14 13 11 7 48 ${EXTRA_DBG}; do
  DM_DEC_TAIL_DBG=$dbg python3 $root/tools/exp/dec_tail_parts.py 2>/dev/null | grep "^dbg" >> $out
done
cat $out
