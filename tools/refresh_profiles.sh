#!/bin/bash
# Regenerates the round's judged artefacts under gpurun_out/refresh (run from the repo root on the GPU box, copy into profiles/ afterwards):
#   tools/refresh_profiles.sh r02
set -e
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/refresh
mkdir -p $out
export TMPDIR=/tmp
python3 $root/bench.py > $out/${tag}_c3_b2048_bench.json 2> $out/bench.err
python3 $root/bench.py --workload c2 --no-roofline > $out/${tag}_c2_b1024_bench.json 2>> $out/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c3 -- python3 $root/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-targets > $out/${tag}_c3_b2048_bench_under_profiler.json 2> $out/c3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c2 -- python3 $root/bench.py --workload c2 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-targets > $out/c2_under_profiler.json 2> $out/c2.err
cd $root
cp $(ls $out/c3/*/*kernel_stats.csv | head -1) $out/${tag}_c3_b2048_kernel_stats.csv
cp $(ls $out/c2/*/*kernel_stats.csv | head -1) $out/${tag}_c2_b1024_kernel_stats.csv
python3 tools/kstats_summary.py $out/c3 35 > $out/${tag}_c3_b2048_kernel_table.txt
python3 tools/kstats_summary.py $out/c2 35 > $out/${tag}_c2_b1024_kernel_table.txt
rm -rf $out/c3 $out/c2
DM_DIST_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 > $out/${tag}_bench_dp2_gloo_rehearsal.json 2> $out/dp2.err
tail -c 600 $out/${tag}_c3_b2048_bench.json; echo; tail -3 $out/${tag}_c3_b2048_kernel_table.txt; tail -c 400 $out/${tag}_bench_dp2_gloo_rehearsal.json
