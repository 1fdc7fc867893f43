cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/test.log 2>&1; tail -3 gpurun_out/test.log
for v in 256 128; do
  if [ $v = 128 ]; then export ABO_TILE128=1; else unset ABO_TILE128; fi
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/t_$v.log 2>&1
  python - <<PY
import json
j=json.loads(open("gpurun_out/t_$v.log").read().strip().splitlines()[-1]); print("tile=$v", round(j["value"],1), round(j["roofline"]["achieved"],2), round(j["roofline"]["avg_launch_ms"],3))
PY
  rm -rf gpurun_out/pmc_t$v
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_t$v -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_t$v.log 2>&1
  python3 - <<PY
import csv,glob
v=[float(r["Counter_Value"]) for f in glob.glob("gpurun_out/pmc_t$v/**/*counter_collection.csv",recursive=True) for r in csv.DictReader(open(f)) if "var_gemm" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
print("tile=$v FETCH_SIZE mean KB", sum(v)/len(v), "n", len(v))
PY
done
