#!/bin/bash
# Two `rocprofv3 --pmc` passes (SQ counters only, no trace domains) over a command, summarised by tools/sq_summary.py.
#   tools/sqprof.sh OUTTAG python3 tools/vqbench.py headline        (run from the repo root on the GPU box)
# The program after `--` is python3 itself (never env/bash -c: the profiler initialises the GPU before the program starts).
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
# the profiler runs from /tmp: a repo-relative script argument (tools/vqbench.py, bench.py) is rewritten to $root/...
args=()
for a in "$@"; do
    if [[ "$a" != /* && -f "$root/$a" ]]; then args+=("$root/$a"); else args+=("$a"); fi
done
set -- "${args[@]}"
mkdir -p $root/gpurun_out
trap 'rc=$?; if [ $rc -ne 0 ]; then echo "sqprof.sh failed ($rc); log tails:"; tail -n 20 $root/gpurun_out/${tag}_sq?.log 2>/dev/null; fi' EXIT
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $root/gpurun_out/${tag}_sqA -- "$@" > $root/gpurun_out/${tag}_sqA.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $root/gpurun_out/${tag}_sqB -- "$@" > $root/gpurun_out/${tag}_sqB.log 2>&1
cd $root
python3 tools/sq_summary.py gpurun_out/${tag}_sqA gpurun_out/${tag}_sqB gpurun_out/${tag}_sq.json | tee gpurun_out/${tag}_sq.txt
