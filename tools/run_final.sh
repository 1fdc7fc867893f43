# Round artefacts: tests, bench (c3 headline, c2, c5), rocprofv3 kernel stats, PMC passes -> gpurun_out/final_*
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r01}
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/final_smoke.txt 2>&1 || { tail -5 gpurun_out/final_smoke.txt; exit 1; }
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/final_pytest_gpu.txt 2>&1 || { tail -30 gpurun_out/final_pytest_gpu.txt; exit 1; }
tail -1 gpurun_out/final_pytest_gpu.txt
timeout -k 10 600 python bench.py --steps 5 --warmup 2 > gpurun_out/final_bench_c3.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
timeout -k 10 300 python bench.py --config c2 --steps 20 --warmup 3 --cpu-sample-m 65536 > gpurun_out/final_bench_c2.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
timeout -k 10 300 python bench.py --config c5 --steps 5 --warmup 2 > gpurun_out/final_bench_c5.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
echo "bench done"
rm -rf gpurun_out/prof_$TAG
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1 || { tail -5 gpurun_out/bench_prof.log; exit 1; }
cp $(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/final_kernel_stats_c3.csv
echo "kernel trace done"
bash tools/run_pmc.sh ${TAG}f c3 > gpurun_out/final_pmc_run.log 2>&1 || { tail -5 gpurun_out/final_pmc_run.log; exit 1; }
cp gpurun_out/pmc_${TAG}f_summary.txt gpurun_out/final_pmc_summary.txt
echo "pmc done"
python3 tools/pmc_traffic_json.py gpurun_out/final_pmc_summary.txt 8192 16384 > gpurun_out/final_pmc_traffic.json
bash tools/run_prof_c5.sh > gpurun_out/final_prof_c5.log 2>&1 || { tail -5 gpurun_out/final_prof_c5.log; exit 1; }
echo "c5 trace done"
