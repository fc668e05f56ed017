#!/bin/bash
# VectorQuantizer headline kernel: registers bound (waves per SIMD) x workgroups per CU, both filters (tools/vqbench.py)
root=${GRAFT_REPO_ROOT:-$(pwd)}
for occ in 3 4; do
  for wgs in 2 3 4 6 8; do
    echo "== DM_VQ_OCC=$occ DM_VQ_WGS=$wgs"
    DM_VQ_OCC=$occ DM_VQ_WGS=$wgs python3 $root/tools/vqbench.py headline c2 2>&1 | grep -E "mfma|bf16"
  done
done
