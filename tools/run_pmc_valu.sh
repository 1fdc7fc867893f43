cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_valu
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d gpurun_out/pmc_valu -- python3 bench.py --config c3 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_valu.log 2>&1 || { tail -5 gpurun_out/pmc_valu.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_valu/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "kgen" in k or "var_gemm" in k:
        print(k, {c: sum(x)/len(x) for c, x in v.items()}, "n=", len(next(iter(v.values()))))
PY
