# Round artefacts, in stages (a gpurun call is limited to 20 minutes): bash tools/run_final.sh <tag> <stage>
#   tests    smoke() + pytest -m gpu
#   pmc      PMC passes of the dominant kernels (first: the bench lines of the next stage read their traffic figures)
#   bench    default line (C3 headline + C2 / C5 / C1-shape secondaries), fp64 engine, C2, C5, shards on one device (2 x C3, 2 x C5,
#            8 x C4 at its own size, 4 x C5 at its own size: rehearsals of the fan-out, not scaling figures)
#   latency  small-N / optimize_acquisition / plain-C host / fit / gradient-engine / hyper-parameter latencies, soaks
#   drift    config 5 without its refresh: 64 appends against an oracle refit every 16 (tools/c5_refresh_drift.py)
#   traces   rocprofv3 --kernel-trace --stats per configuration
# everything lands in gpurun_out/final_*; tools/collect_final.sh copies it into profiles/ afterwards (run here, not on the box)
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r06}
STAGE=${2:-all}
want() { [ "$STAGE" = all ] || [ "$STAGE" = "$1" ]; }
if want tests; then
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/final_smoke.txt 2>&1 || { tail -5 gpurun_out/final_smoke.txt; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/final_pytest_gpu.txt 2>&1 || { tail -30 gpurun_out/final_pytest_gpu.txt; exit 1; }
tail -1 gpurun_out/final_pytest_gpu.txt
fi
if want pmc; then
bash tools/run_pmc_int8.sh ${TAG}i > gpurun_out/final_pmc_int8_run.log 2>&1 || { tail -5 gpurun_out/final_pmc_int8_run.log; exit 1; }
cp gpurun_out/pmc_${TAG}i_summary.txt gpurun_out/final_pmc_int8_summary.txt
MC=$(grep '^{' gpurun_out/pmc_${TAG}i_fetch.log | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(int(d['config']['M_per_gpu'] // d['roofline']['launches_per_step']))")
python3 tools/pmc_traffic_json.py gpurun_out/final_pmc_int8_summary.txt 8192 $MC int8 14 > gpurun_out/final_c3_int8_pmc_traffic.json
cp gpurun_out/final_c3_int8_pmc_traffic.json profiles/${TAG}_c3_int8_pmc_traffic.json    # the bench lines below read it (same source hash)
echo "pmc c3 int8 done"
ABO_CONTRACTION=fp64 bash tools/run_pmc.sh ${TAG}f c3 > gpurun_out/final_pmc_run.log 2>&1 || { tail -5 gpurun_out/final_pmc_run.log; exit 1; }
cp gpurun_out/pmc_${TAG}f_summary.txt gpurun_out/final_pmc_summary.txt
python3 tools/pmc_traffic_json.py gpurun_out/final_pmc_summary.txt 8192 16384 > gpurun_out/final_c3_pmc_traffic.json
cp gpurun_out/final_c3_pmc_traffic.json profiles/${TAG}_c3_pmc_traffic.json
echo "pmc c3 fp64 done"
bash tools/run_pmc_c5.sh > gpurun_out/final_pmc_c5.log 2>&1 || { tail -5 gpurun_out/final_pmc_c5.log; exit 1; }
cp gpurun_out/c5_pmc_traffic.json profiles/${TAG}_c5_pmc_traffic.json
echo "pmc c5 done"
fi
if want bench; then
timeout -k 10 900 python bench.py > gpurun_out/final_bench_default.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
timeout -k 10 300 python bench.py --contraction fp64 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/final_bench_c3_fp64_engine.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
timeout -k 10 300 python bench.py --config c2 --steps 30 --warmup 5 --cpu-sample-m 65536 --cpu-reps 3 > gpurun_out/final_bench_c2.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
timeout -k 10 300 python bench.py --config c5 --steps 128 --warmup 2 > gpurun_out/final_bench_c5.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
timeout -k 10 600 python bench.py --gpus 2 --share-device --steps 3 --warmup 1 > gpurun_out/final_bench_c3_2shards_one_device.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
timeout -k 10 300 python bench.py --gpus 2 --share-device --steps 16 --warmup 1 --config c5 > gpurun_out/final_bench_c5_2shards_one_device.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
# BASELINE config 4 at its own size through the 8-shard path, all shards on this one GPU (a rehearsal of the fan-out, NOT a scaling figure)
timeout -k 10 600 python bench.py --gpus 8 --share-device --config c4 --steps 2 --warmup 1 > gpurun_out/final_bench_c4_8shards_one_device.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
timeout -k 10 600 python bench.py --gpus 4 --share-device --config c5 --steps 16 --warmup 1 > gpurun_out/final_bench_c5_4shards_one_device.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
# BASELINE config 4's candidate count on ONE GPU (one rank, M = 2^23: the denominator of the >= 6x scaling target — a real single-GPU line)
timeout -k 10 300 python bench.py --config c4 --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/final_bench_c4_1gpu.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
echo "bench done"
fi
if want latency; then
(echo "tools/small_n_latency.py: refit + EI over M + top-100 at the sizes the reference's own loops live at"; echo "--- default (phase events automatic: off for N <= 128)"; timeout -k 10 200 python tools/small_n_latency.py 2>&1 | grep "^N="; echo "--- ABO_PHASE_EVENTS=1"; ABO_PHASE_EVENTS=1 timeout -k 10 200 python tools/small_n_latency.py 2>&1 | grep "^N="; echo "--- ABO_PHASE_EVENTS=0"; ABO_PHASE_EVENTS=0 timeout -k 10 200 python tools/small_n_latency.py 2>&1 | grep "^N=") > gpurun_out/final_small_n_latency.txt
(timeout -k 10 600 python tools/optimize_acquisition_latency.py 2>/dev/null | grep -v amdgpu.ids; echo; echo "--- the one-launch kernel at every size (ABO_REFINE_LOCKSTEP_NP=0)"; ABO_REFINE_LOCKSTEP_NP=0 timeout -k 10 600 python tools/optimize_acquisition_latency.py 2>/dev/null | grep "^N="; echo "--- lockstep rounds at every size (ABO_REFINE_LOCKSTEP_NP=128)"; ABO_REFINE_LOCKSTEP_NP=128 timeout -k 10 600 python tools/optimize_acquisition_latency.py 2>/dev/null | grep "^N="; echo "--- lockstep rounds without the split-k / skinny products (ABO_REFINE_KSPLIT=0: round 3)"; ABO_REFINE_KSPLIT=0 timeout -k 10 600 python tools/optimize_acquisition_latency.py 2048 8192 2>/dev/null | grep "^N=") > gpurun_out/final_optimize_acquisition_latency.txt
(echo "tools/c_host_latency.sh: per-step latency from the plain-C host (tests/c_abi_harness.c latency; system HIP runtime, no interpreter, host arrays in, top-100 out)"; bash tools/c_host_latency.sh 2>&1 | grep "^latency") > gpurun_out/final_c_host_latency.txt
timeout -k 10 300 bash tools/fit_times.sh > gpurun_out/final_fit_times.txt 2>&1 || true
(timeout -k 10 300 python tools/grad_engine_latency.py 2>&1 | grep -v amdgpu; timeout -k 10 300 python tools/grad_engine_latency.py 400 16 2048 2>&1 | grep -v amdgpu) > gpurun_out/final_grad_engine_latency.txt
timeout -k 10 600 python tools/oz_soak.py 40 2>&1 | grep "^N=" > gpurun_out/final_oz_soak.txt || true
(timeout -k 10 300 python tools/c5_cycle.py 512 16 32 64 2>&1 | grep -v amdgpu.ids) > gpurun_out/final_c5_cycle.txt
timeout -k 10 600 python tools/soak.py > gpurun_out/final_soak.txt 2>&1 || { tail -5 gpurun_out/final_soak.txt; exit 1; }
(timeout -k 10 200 python tools/ensemble_latency.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/final_ensemble_latency.txt
(timeout -k 10 300 python tools/hyperparameter_latency.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/final_hyperparameter_latency.txt
(timeout -k 10 300 python tools/refine_diag.py c3 2>&1 | grep -v amdgpu.ids) > gpurun_out/final_refine_diag_c3.txt
echo "latency tools done"
fi
if want drift; then
(timeout -k 10 900 python tools/c5_refresh_drift.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/final_c5_refresh_drift.txt
tail -3 gpurun_out/final_c5_refresh_drift.txt
(timeout -k 10 900 python tools/c5_refresh_drift.py 512 128 2>&1 | grep -v amdgpu.ids) > gpurun_out/final_c5_refresh_drift_512.txt
tail -3 gpurun_out/final_c5_refresh_drift_512.txt
fi
if want traces; then
rm -rf gpurun_out/prof_${TAG}_c3fp64
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_c3fp64 -- python3 bench.py --config c3 --contraction fp64 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/bench_prof_c3fp64.log 2>&1 || { tail -5 gpurun_out/bench_prof_c3fp64.log; exit 1; }
cp $(find gpurun_out/prof_${TAG}_c3fp64 -name "*kernel_stats.csv" | head -1) gpurun_out/final_kernel_stats_c3_fp64_engine.csv
for cfg in c3 c2 c5; do
  rm -rf gpurun_out/prof_${TAG}_$cfg
  steps=5; [ $cfg = c2 ] && steps=30
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_$cfg -- python3 bench.py --config $cfg --steps $steps --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/bench_prof_$cfg.log 2>&1 || { tail -5 gpurun_out/bench_prof_$cfg.log; exit 1; }
  cp $(find gpurun_out/prof_${TAG}_$cfg -name "*kernel_stats.csv" | head -1) gpurun_out/final_kernel_stats_$cfg.csv
  grep '^{' gpurun_out/bench_prof_$cfg.log > gpurun_out/final_bench_under_rocprof_$cfg.json || true
done
echo "kernel traces done"
timeout -k 10 120 tools/mfma_i8_power_probe > gpurun_out/final_mfma_i8_power_probe.txt 2>&1 || true
fi
