#!/bin/bash
# vq_cells_kernel (csrc/vq_cells.h) with parts compiled out (results are wrong with any bit set; the time is what is read).
# VQC_OFF bits: 4 no exact re-checks, 8 no exact evaluation of the best cell, 16 no group ends, 32 no cell minima,
# 64 operand ring never refilled, 128 norms read once per pass (16 / 32 make every position degenerate: not timeable).
#   tools/exp/vq_cells_parts.sh build     here (hipcc cross-compiles): one library per variant next to the package's
#   gpurun -- tools/exp/vq_cells_parts.sh on the GPU box
cd "$(dirname "$0")/../.."
VARIANTS="${VARIANTS:-0 4 12 64 128 192}"
if [ "$1" = build ]; then
  cd dynamorph_amd/csrc && mkdir -p build_measure
  for v in $VARIANTS; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DVQC_OFF=$v -c vq.hip -o build_measure/vq_off$v.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdm_vqoff$v.so build_measure/vq_off$v.o $(ls *.o | grep -v "^vq.o") &
  done; wait; ls -la ../libdm_vqoff*.so; exit 0
fi
export VQBENCH_ONLY=bf16
for shape in ${SHAPES:-c5model}; do
  for prod in ${PRODS:-3}; do
    for v in $VARIANTS; do
      echo -n "$shape prod=$prod off=$v  "
      DM_VQ_CELLS_PROD=$prod DM_LIB_PATH=$PWD/dynamorph_amd/libdm_vqoff$v.so python3 tools/vqbench.py $shape 2>/dev/null | grep " bf16 " | cut -c30-190
    done
  done
done
