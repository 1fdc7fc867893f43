# smoke + tests + bench + rocprofv3 kernel stats for one round tag
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r01}
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 || { tail -5 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log | cut -c1-120
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/test.log 2>&1 || { tail -30 gpurun_out/test.log; exit 1; }
tail -1 gpurun_out/test.log
timeout -k 10 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
cut -c1-250 gpurun_out/bench_$TAG.json
rm -rf gpurun_out/prof_$TAG
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1 || { tail -5 gpurun_out/bench_prof.log; exit 1; }
cp $(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats_$TAG.csv
head -8 gpurun_out/kernel_stats_$TAG.csv | cut -c1-160
