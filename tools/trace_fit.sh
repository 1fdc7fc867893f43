# kernel trace of the fit alone: tools/trace_fit.sh <tag> <N> <d>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; N=$2; D=$3
rm -rf gpurun_out/trace_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_$TAG -- python3 tools/fit_only.py $N $D 3 > gpurun_out/trace_$TAG.log 2>&1 || { tail -5 gpurun_out/trace_$TAG.log; exit 1; }
python3 tools/trace_summary.py $(find gpurun_out/trace_$TAG -name "*kernel_trace.csv" | head -1) > gpurun_out/trace_${TAG}_summary.txt
tail -3 gpurun_out/trace_$TAG.log
