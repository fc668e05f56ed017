#!/bin/bash
# Regenerates the round's judged artefacts under gpurun_out/refresh (run from the repo root on the GPU box, copy into profiles/ afterwards):
#   tools/refresh_profiles.sh r06
set -e
tag=${1:-r06}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/refresh
mkdir -p $out
export TMPDIR=/tmp
python3 $root/bench.py > $out/${tag}_c3_b2048_bench.json 2> $out/bench.err
python3 $root/bench.py --workload c2 --no-roofline > $out/${tag}_c2_b1024_bench.json 2>> $out/bench.err
python3 $root/bench.py --workload c5 > $out/${tag}_c5_k4096_256px_b1024_bench.json 2>> $out/bench.err
python3 $root/bench.py --workload z32ex > $out/${tag}_z32ex_b768_bench.json 2>> $out/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c3 -- python3 $root/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-targets > $out/${tag}_c3_b2048_bench_under_profiler.json 2> $out/c3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c2 -- python3 $root/bench.py --workload c2 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-targets > $out/c2_under_profiler.json 2> $out/c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c5 -- python3 $root/bench.py --workload c5 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $out/c5_under_profiler.json 2> $out/c5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/z32 -- python3 $root/bench.py --workload z32ex --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $out/z32_under_profiler.json 2> $out/z32.err
cd $root
cp $(ls $out/c3/*/*kernel_stats.csv | head -1) $out/${tag}_c3_b2048_kernel_stats.csv
cp $(ls $out/c2/*/*kernel_stats.csv | head -1) $out/${tag}_c2_b1024_kernel_stats.csv
cp $(ls $out/c5/*/*kernel_stats.csv | head -1) $out/${tag}_c5_k4096_256px_b1024_kernel_stats.csv
cp $(ls $out/z32/*/*kernel_stats.csv | head -1) $out/${tag}_z32ex_b768_kernel_stats.csv
python3 tools/kstats_summary.py $out/c3 41 > $out/${tag}_c3_b2048_kernel_table.txt
python3 tools/kstats_summary.py $out/c2 35 > $out/${tag}_c2_b1024_kernel_table.txt
python3 tools/kstats_summary.py $out/c5 19 > $out/${tag}_c5_k4096_256px_b1024_kernel_table.txt
python3 tools/kstats_summary.py $out/z32 19 > $out/${tag}_z32ex_b768_kernel_table.txt
python3 tools/exp/step_launches.py $out/z32 adam_kernel 30 > $out/${tag}_z32ex_b768_step_launches.txt
rm -rf $out/c3 $out/c2 $out/c5 $out/z32
# HBM bytes per launch of the roofline kernels: the two counters in their own passes, no trace domains with them
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $root/tools/kbench.py > $out/kbench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $root/tools/kbench.py > $out/kbench_write.log 2>&1
cd $root
python3 tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write 2048 > $out/${tag}_pmc_traffic.json
rm -rf $out/pmc_fetch $out/pmc_write
# ... and of the example configuration's rebuilt kernels, layer by layer at B = 768
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/zf -- python3 $root/tools/exp/z32_layers.py > $out/z32_layers_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/zw -- python3 $root/tools/exp/z32_layers.py > $out/z32_layers_write.log 2>&1
cd $root
python3 tools/exp/z32_pmc.py $out/zf $out/zw > $out/${tag}_z32ex_pmc_traffic.json
rm -rf $out/zf $out/zw
python3 tools/vqbench.py > $out/${tag}_vq_kernels.txt 2>&1
# train() end to end (resident / streaming / synchronous feeds) and the kernels of the resident loop
python3 tools/trainbench.py --full > $out/${tag}_trainbench.jsonl 2> $out/trainbench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/tb -- python3 $root/tools/trainbench.py --feeds resident --epochs 2 > $out/tb_under_profiler.jsonl 2> $out/tb.err
cd $root
python3 tools/kstats_summary.py $out/tb > $out/${tag}_trainloop_resident_kernel_table.txt
rm -rf $out/tb
# SQ counters of the step's kernels (two --pmc passes, no trace domains)
tools/sqprof.sh ${tag}_c3 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-targets --no-roofline > /dev/null 2>&1
cp gpurun_out/${tag}_c3_sq.txt $out/${tag}_c3_b2048_sq_counters.txt 2>/dev/null
cp gpurun_out/${tag}_c3_sq.json $out/${tag}_c3_b2048_sq_counters.json 2>/dev/null
DM_DIST_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 2> $out/dp2.err | grep '^{' > $out/${tag}_bench_dp2_gloo_rehearsal.json   # (gloo prints its connection lines on stdout)
tail -c 600 $out/${tag}_c3_b2048_bench.json; echo; tail -3 $out/${tag}_c3_b2048_kernel_table.txt; tail -c 400 $out/${tag}_bench_dp2_gloo_rehearsal.json
