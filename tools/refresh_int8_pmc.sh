# the int8 PMC passes + the bench lines that read them (after an edit of ozaki.hip / abo_oz_dev.h / abo_kernels.h: the traffic figure is
# only printed while the sources still hash to what the PMC pass recorded)
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r04}
bash tools/run_pmc_int8.sh ${TAG}i > gpurun_out/final_pmc_int8_run.log 2>&1 || { tail -5 gpurun_out/final_pmc_int8_run.log; exit 1; }
cp gpurun_out/pmc_${TAG}i_summary.txt gpurun_out/final_pmc_int8_summary.txt
MC=$(grep '^{' gpurun_out/pmc_${TAG}i_fetch.log | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(int(d['config']['M_per_gpu'] // d['roofline']['launches_per_step']))")
python3 tools/pmc_traffic_json.py gpurun_out/final_pmc_int8_summary.txt 8192 $MC int8 14 > gpurun_out/final_c3_int8_pmc_traffic.json
cp gpurun_out/final_c3_int8_pmc_traffic.json profiles/${TAG}_c3_int8_pmc_traffic.json
echo "pmc c3 int8 done"
timeout -k 10 900 python bench.py > gpurun_out/final_bench_default.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
timeout -k 10 600 python bench.py --gpus 2 --share-device --steps 3 --warmup 1 > gpurun_out/final_bench_c3_2shards_one_device.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
rm -rf gpurun_out/prof_${TAG}_c3
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_c3 -- python3 bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/bench_prof_c3.log 2>&1 || { tail -5 gpurun_out/bench_prof_c3.log; exit 1; }
cp $(find gpurun_out/prof_${TAG}_c3 -name "*kernel_stats.csv" | head -1) gpurun_out/final_kernel_stats_c3.csv
grep '^{' gpurun_out/bench_prof_c3.log > gpurun_out/final_bench_under_rocprof_c3.json || true
echo "bench done"
