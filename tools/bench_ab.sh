# C3 and C2 bench lines, compact:  tools/bench_ab.sh [label]
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); p=j['phases_ms']
print('$1 c3: step %.1f ms (median %.1f)  kxz %.2f  var_gemm %.1f (%.4f of peak)  fit %.2f' % (j['value'], j['median_ms_per_step'], p['acq_kxz_ms'], p['acq_var_gemm_ms'], j['roofline']['frac'], p['fit_total_ms']))"
timeout -k 10 200 python bench.py --config c2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); p=j['phases_ms']
print('$1 c2: step %.3f ms  kxz %.3f  var_gemm %.3f (%.3f of peak)  fit %.3f  acq_total %.3f' % (j['value'], p['acq_kxz_ms'], p['acq_var_gemm_ms'], j['roofline']['frac'], p['fit_total_ms'], p['acq_total_ms']))"
done
