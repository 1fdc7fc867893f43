# L2 hit rate and L2-miss-side read traffic of the residue GEMM: persistent kernel against the one-tile kernel (ABO_OZ_ONE_TILE=1)
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in persistent onetile; do
  [ $v = onetile ] && export ABO_OZ_ONE_TILE=1
  for pass in tcc fetch; do
    rm -rf gpurun_out/pmc_ab_${v}_$pass
    if [ $pass = tcc ]; then C="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; else C="FETCH_SIZE"; fi
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmc_ab_${v}_$pass -- python3 bench.py --config c3 --contraction int8 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > gpurun_out/pmc_ab_${v}_$pass.log 2>&1 || { tail -5 gpurun_out/pmc_ab_${v}_$pass.log; exit 1; }
  done
done
python3 - <<'PY'
import csv, glob, collections
for v in ("persistent", "onetile"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in ("tcc", "fetch"):
        for f in glob.glob(f"gpurun_out/pmc_ab_{v}_{p}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "oz_gemm16" in r["Kernel_Name"]:
                    acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        m = {n: sum(x) / len(x) for n, x in c.items()}
        print(v, k, "launches", len(c.get("FETCH_SIZE", [])), "L2 hit", round(m.get("TCC_HIT_sum", 0) / max(m.get("TCC_REQ_sum", 1), 1), 4),
              "FETCH_SIZE KB", round(m.get("FETCH_SIZE", 0)), "-> read traffic GB per launch (x2 correction)", round(m.get("FETCH_SIZE", 0) * 2 * 1024 / 1e9, 1))
PY
