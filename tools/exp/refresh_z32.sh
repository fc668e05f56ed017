set -e
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/refresh2; mkdir -p $out; export TMPDIR=/tmp
python3 $root/bench.py > $out/r06_c3_b2048_bench.json 2> $out/bench.err
python3 $root/bench.py --workload z32ex > $out/r06_z32ex_b768_bench.json 2>> $out/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/z32 -- python3 $root/bench.py --workload z32ex --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $out/z32_under_profiler.json 2> $out/z32.err
cd $root
cp $(ls $out/z32/*/*kernel_stats.csv | head -1) $out/r06_z32ex_b768_kernel_stats.csv
python3 tools/kstats_summary.py $out/z32 24 > $out/r06_z32ex_b768_kernel_table.txt
python3 tools/exp/step_launches.py $out/z32 adam_kernel 30 > $out/r06_z32ex_b768_step_launches.txt
rm -rf $out/z32
tail -c 300 $out/r06_z32ex_b768_bench.json; echo; tail -2 $out/r06_z32ex_b768_kernel_table.txt
