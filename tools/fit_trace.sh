# Kernel trace of fits at two sizes, summarised launch by launch: bash tools/fit_trace.sh  → gpurun_out/fit_trace_summary.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
: > gpurun_out/fit_trace_summary.txt
for cfg in "1024 4" "8192 8"; do
  set -- $cfg
  rm -rf gpurun_out/fit_trace_$1
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fit_trace_$1 -- python3 tools/fit_only.py $1 $2 2 > gpurun_out/fit_trace_$1.log 2>&1 || { tail -5 gpurun_out/fit_trace_$1.log; exit 1; }
  python3 tools/fit_trace_summary.py $(find gpurun_out/fit_trace_$1 -name "*kernel_trace.csv" | head -1) $1 5 >> gpurun_out/fit_trace_summary.txt
done
cat gpurun_out/fit_trace_summary.txt
