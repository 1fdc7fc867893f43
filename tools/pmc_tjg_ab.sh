# int8 residue GEMM: patch shape (4 row blocks x tjg/8 column blocks per XCD) against L2-miss-side traffic and time, one box.
# ABO_OZ_TJG = 32 / 64 (shipped) / 128; per setting: the C3 step (2 steps) for the HIP-event time of the residue GEMMs, then a
# FETCH_SIZE pass and a TCC hit/miss pass (own runs, --kernel-trace only).
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for T in 32 64 128; do
  export ABO_OZ_TJG=$T
  python3 bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/tjg_${T}_bench.json 2> gpurun_out/tjg_${T}_bench.err
  for C in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    name=$(echo $C | cut -d' ' -f1)
    rm -rf gpurun_out/pmc_tjg_${T}_$name
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmc_tjg_${T}_$name -- python3 bench.py --config c3 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > gpurun_out/pmc_tjg_${T}_$name.log 2>&1 || { tail -5 gpurun_out/pmc_tjg_${T}_$name.log; exit 1; }
  done
  echo "tjg $T done"
done
python3 - <<'PY'
import csv, glob, json, collections
print("tjg  step_ms  oz_gemm_ms/step  FETCH GB/launch  L2 hit   (oz_gemm16p_kernel, C3: N = 8192, 65536 candidates per launch)")
for T in (32, 64, 128):
    b = json.loads([l for l in open(f"gpurun_out/tjg_{T}_bench.json") if l.startswith("{")][-1])
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_tjg_{T}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "oz_gemm16p" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    hit = m.get("TCC_HIT_sum", 0) / max(1.0, m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0))
    print(f"{T:4d} {b['value']:8.1f} {b.get('phases_ms', {}).get('oz_gemm_ms', float('nan')):10.1f} {m.get('FETCH_SIZE', 0) * 2048 / 1e9:14.1f} {hit:8.3f}")
PY
