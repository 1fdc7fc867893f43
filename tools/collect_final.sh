# copy what tools/run_final.sh left under gpurun_out/ into profiles/ (run here, after the gpurun call): bash tools/collect_final.sh r04
TAG=${1:-r06}
cd gpurun_out
cp final_bench_default.json ../profiles/${TAG}_bench_default.json
cp final_bench_c2.json ../profiles/${TAG}_c2_bench.json
cp final_bench_c5.json ../profiles/${TAG}_c5_bench.json
cp final_bench_c3_fp64_engine.json ../profiles/${TAG}_c3_bench_fp64_engine.json
cp final_bench_c3_2shards_one_device.json ../profiles/${TAG}_bench_c3_2shards_one_device.json
cp final_bench_c5_2shards_one_device.json ../profiles/${TAG}_bench_c5_2shards_one_device.json
cp final_bench_c4_8shards_one_device.json ../profiles/${TAG}_bench_c4_8shards_one_device_rehearsal.json
cp final_bench_c5_4shards_one_device.json ../profiles/${TAG}_bench_c5_4shards_one_device_rehearsal.json
cp final_bench_c4_1gpu.json ../profiles/${TAG}_bench_c4_1gpu.json
cp final_hyperparameter_latency.txt ../profiles/${TAG}_hyperparameter_latency.txt
[ -f final_ensemble_latency.txt ] && cp final_ensemble_latency.txt ../profiles/${TAG}_ensemble_latency.txt
cp final_refine_diag_c3.txt ../profiles/${TAG}_refine_diag_c3.txt
[ -f final_c5_refresh_drift.txt ] && cp final_c5_refresh_drift.txt ../profiles/${TAG}_c5_refresh_drift.txt
[ -f final_c5_refresh_drift_512.txt ] && cp final_c5_refresh_drift_512.txt ../profiles/${TAG}_c5_refresh_drift_512.txt
cp final_grad_engine_latency.txt ../profiles/${TAG}_grad_engine_latency.txt
cp final_oz_soak.txt ../profiles/${TAG}_oz_soak.txt
[ -f final_c5_cycle.txt ] && cp final_c5_cycle.txt ../profiles/${TAG}_c5_cycle.txt
for c in c2 c3 c5; do cp final_bench_under_rocprof_$c.json ../profiles/${TAG}_${c}_bench_under_rocprof.json; cp final_kernel_stats_$c.csv ../profiles/${TAG}_${c}_kernel_stats.csv; done
cp final_kernel_stats_c3_fp64_engine.csv ../profiles/${TAG}_c3_kernel_stats_fp64_engine.csv
cp final_pmc_int8_summary.txt ../profiles/${TAG}_c3_int8_pmc_summary.txt
cp final_pmc_summary.txt ../profiles/${TAG}_c3_pmc_summary.txt
cp final_c3_pmc_traffic.json ../profiles/${TAG}_c3_pmc_traffic.json
cp final_c3_int8_pmc_traffic.json ../profiles/${TAG}_c3_int8_pmc_traffic.json
cp c5_pmc_traffic.json ../profiles/${TAG}_c5_pmc_traffic.json
cp pmc_c5_fetch_summary.txt ../profiles/${TAG}_c5_pmc_fetch_summary.txt
cp final_mfma_i8_power_probe.txt ../profiles/${TAG}_mfma_i8_power_probe.txt
cp final_small_n_latency.txt ../profiles/${TAG}_small_n_latency.txt
cp final_optimize_acquisition_latency.txt ../profiles/${TAG}_optimize_acquisition_latency.txt
cp final_c_host_latency.txt ../profiles/${TAG}_c_host_latency.txt
cp final_fit_times.txt ../profiles/${TAG}_fit_times.txt
cp fit_trace_summary.txt ../profiles/${TAG}_fit_trace_summary.txt
cp final_soak.txt ../profiles/${TAG}_soak.txt
cp final_pytest_gpu.txt ../profiles/${TAG}_pytest_gpu.txt
cp final_smoke.txt ../profiles/${TAG}_smoke.txt
cd ..
python tools/update_parity_bounds.py
