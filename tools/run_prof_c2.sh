cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_c2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c2 -- python3 bench.py --config c2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_prof_c2.log 2>&1
cp $(find gpurun_out/prof_c2 -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats_c2.csv
cut -c1-150 gpurun_out/kernel_stats_c2.csv | head -20
