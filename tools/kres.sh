#!/bin/bash
# VGPRs / scratch / LDS / occupancy of every kernel in dynamorph_amd/csrc, as hipcc reports them for gfx950
# (-Rpass-analysis=kernel-resource-usage; no GPU needed):   tools/kres.sh > profiles/r03_kernel_resources.txt
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root/dynamorph_amd/csrc
printf "%-100s %6s %8s %8s %5s\n" kernel VGPRs scratchB LDS_B occ
for f in *.hip; do
    extra=""; [ "$f" = vq.hip ] && extra="-ffp-contract=off"; [ "$f" = dec_tail.hip ] && extra="-fno-slp-vectorize"      # as the Makefile builds them
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $extra -I../../include -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/kres_$$.o 2>&1 |
        grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* *//; s/ \[-Rpass.*//' | paste - - - - - |
        while IFS=$'\t' read -r name vg sc occ lds; do
            n=$(echo "${name#*Name: }" | c++filt | sed 's/(anonymous namespace):://; s/(.*//; s/^void //')
            printf "%-100s %6s %8s %8s %5s\n" "${n:0:100}" "${vg##*: }" "${sc##*: }" "${lds##*: }" "${occ##*: }"
        done
done
rm -f /tmp/kres_$$.o
