for c in 2048 4096 8192 16384 32768 65536; do
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --chunk $c > gpurun_out/ch_$c.log 2>&1
  python - <<PY
import json
j=json.loads(open("gpurun_out/ch_$c.log").read().strip().splitlines()[-1]); print("chunk=$c", round(j["value"],1), round(j["roofline"]["achieved"],2), j["roofline"]["launches_per_step"], round(j["phases_ms"]["acq_var_gemm_ms"],1), round(j["phases_ms"]["acq_kxz_ms"],1))
PY
done
