#!/bin/bash
# vq_cells_kernel (csrc/vq_cells.h) with parts switched off -- measurement build of vq.hip only:
#   cd dynamorph_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DDM_MEASURE -c vq.hip -o build_measure/vq_m.o &&
#   hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdm_vqm.so build_measure/vq_m.o $(ls *.o | grep -v "^vq.o")
# DM_VQ_DBG bits: 4 no exact re-checks, 8 no exact evaluation of the best cell, 16 no group ends, 32 no cell minima
# (results are wrong with any bit set; the time is what is read)
export VQBENCH_ONLY=bf16
for shape in ${SHAPES:-c5model}; do
  for prod in 3 4; do
    for d in 0 4 12 28 60; do
      echo -n "$shape prod=$prod dbg=$d  "
      DM_VQ_CELLS_PROD=$prod DM_LIB_PATH=$PWD/dynamorph_amd/libdm_vqm.so DM_VQ_DBG=$d python3 tools/vqbench.py $shape 2>/dev/null | grep " bf16 " | cut -c30-190
    done
  done
done
