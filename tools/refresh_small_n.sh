# the artefacts the small-N host path feeds: default bench line (its C1-shaped leg), tools/small_n_latency.py in the three phase-event
# modes, the plain-C host's latency, optimize_acquisition's
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python bench.py > gpurun_out/final_bench_default.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
(echo "tools/small_n_latency.py: refit + EI over M + top-100 at the sizes the reference's own loops live at"; echo "--- default (phase events automatic: off for N <= 128)"; timeout -k 10 200 python tools/small_n_latency.py 2>&1 | grep "^N="; echo "--- ABO_PHASE_EVENTS=1"; ABO_PHASE_EVENTS=1 timeout -k 10 200 python tools/small_n_latency.py 2>&1 | grep "^N="; echo "--- ABO_PHASE_EVENTS=0"; ABO_PHASE_EVENTS=0 timeout -k 10 200 python tools/small_n_latency.py 2>&1 | grep "^N=") > gpurun_out/final_small_n_latency.txt
(timeout -k 10 600 python tools/optimize_acquisition_latency.py 2>/dev/null | grep -v amdgpu.ids; echo; echo "--- the one-launch kernel at every size (ABO_REFINE_LOCKSTEP_NP=0)"; ABO_REFINE_LOCKSTEP_NP=0 timeout -k 10 600 python tools/optimize_acquisition_latency.py 2>/dev/null | grep "^N="; echo "--- lockstep rounds at every size (ABO_REFINE_LOCKSTEP_NP=128)"; ABO_REFINE_LOCKSTEP_NP=128 timeout -k 10 600 python tools/optimize_acquisition_latency.py 2>/dev/null | grep "^N=") > gpurun_out/final_optimize_acquisition_latency.txt
(echo "tools/c_host_latency.sh: per-step latency from the plain-C host (tests/c_abi_harness.c latency; system HIP runtime, no interpreter, host arrays in, top-100 out)"; bash tools/c_host_latency.sh 2>&1 | grep "^latency") > gpurun_out/final_c_host_latency.txt
echo done
