cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_c5
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5 -- python3 bench.py --config c5 --steps 3 --warmup 1 > gpurun_out/bench_prof_c5.log 2>&1
cp $(find gpurun_out/prof_c5 -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats_c5.csv
cp $(find gpurun_out/prof_c5 -name "*kernel_trace.csv" | head -1) gpurun_out/kernel_trace_c5.csv
cut -c1-150 gpurun_out/kernel_stats_c5.csv | head -24
