#!/bin/bash
# The large-codebook VectorQuantizer kernel with parts switched off (measurement builds of csrc/vq.hip; results are wrong with
# any switch on, the time is what is read).
#   run time  DM_VQ_DBG: 1 nothing is copied into the LDS buffers after the prologue (the barriers stay), 4 no exact re-checks
#   compile   -DVQ_DBG_NOEMBED no index bits in the scores, -DVQ_DBG_MINONLY in-lane minimum only (no runner-up)
# Build (from dynamorph_amd/csrc, after `make measure` for the other objects):
#   for v in "" "-DVQ_DBG_NOEMBED" "-DVQ_DBG_MINONLY"; do n=$(echo "$v" | tr -d ' -');
#     hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DDM_MEASURE $v -c vq.hip -o build_measure/vq_m$n.o &&
#     hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdm_vqm$n.so build_measure/vq_m$n.o $(ls build_measure/*.o | grep -v /vq); done
export VQBENCH_ONLY=bf16
for shape in ${SHAPES:-stress c5model}; do
for lib in "" DVQ_DBG_NOEMBED DVQ_DBG_MINONLY; do
  [ -f dynamorph_amd/libdm_vqm$lib.so ] || continue
  for d in 0 4 5; do
    echo -n "$shape lib=${lib:-plain} dbg=$d  "
    DM_LIB_PATH=$PWD/dynamorph_amd/libdm_vqm$lib.so DM_VQ_DBG=$d python3 tools/vqbench.py $shape 2>/dev/null | grep " bf16 " | cut -c30-175
  done
done; done
