set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 || { tail -5 gpurun_out/smoke.log; exit 1; }
tail -2 gpurun_out/smoke.log
rm -rf gpurun_out/prof_r01
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1 || { tail -5 gpurun_out/bench_prof.log; exit 1; }
tail -1 gpurun_out/bench_prof.log | cut -c1-300
find gpurun_out/prof_r01 -name "*stats*" | head
