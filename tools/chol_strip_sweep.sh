# Cholesky strip width sweep (ABO_CHOL_STRIP): fit phases at N = 1024 (C2), 8192 (C3), 16384 (C5 refresh)
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/strip_test.log 2>&1 || { tail -20 gpurun_out/strip_test.log; exit 1; }
tail -1 gpurun_out/strip_test.log
for sw in 128 256 512 1024 2048; do
  export ABO_CHOL_STRIP=$sw
  python bench.py --config c2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/sw_c2.json 2>/dev/null
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/sw_c3.json 2>/dev/null
  python bench.py --config c5 --steps 1 --warmup 1 > gpurun_out/sw_c5.json 2>/dev/null
  python - <<PY
import json
a=json.load(open("gpurun_out/sw_c2.json"))["phases_ms"]["fit_cholesky_ms"]
b=json.load(open("gpurun_out/sw_c3.json"))["phases_ms"]["fit_cholesky_ms"]
c=json.load(open("gpurun_out/sw_c5.json"))["refresh_phases_ms"]["fit_cholesky_ms"]
print("strip $sw: cholesky ms  N=1024 %.3f  N=8192 %.2f  N=16384 %.2f" % (a,b,c))
PY
done
