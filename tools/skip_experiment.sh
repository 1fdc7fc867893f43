# A/B: var_gemm256s_kernel (diagonal sub-tile skipping) vs var_gemm256_kernel (ABO_VAR_NOSKIP=1)
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/skip_test.log 2>&1 || { tail -30 gpurun_out/skip_test.log; exit 1; }
tail -1 gpurun_out/skip_test.log
for rep in 1 2; do
for v in skip noskip; do
  if [ $v = noskip ]; then export ABO_VAR_NOSKIP=1; else unset ABO_VAR_NOSKIP; fi
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/sk_c3.json 2>/dev/null
  python bench.py --config c2 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/sk_c2.json 2>/dev/null
  python - <<PY
import json
a=json.load(open("gpurun_out/sk_c3.json")); b=json.load(open("gpurun_out/sk_c2.json"))
print("$v  C3 %.1f ms  var_gemm %.1f ms  %.2f TF/s   |  C2 %.3f ms  var_gemm %.3f ms  %.2f TF/s" % (a["value"], a["phases_ms"]["acq_var_gemm_ms"], a["roofline"]["achieved"], b["value"], b["phases_ms"]["acq_var_gemm_ms"], b["roofline"]["achieved"]))
PY
done
done
