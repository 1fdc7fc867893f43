# HBM-side bytes of the C5 down-date mat-vec (cand_gemv_kernel) and the append's triangular mat-vecs: FETCH_SIZE pass
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_c5_fetch
timeout -k 10 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_c5_fetch -- python3 bench.py --config c5 --steps 2 --warmup 0 > gpurun_out/pmc_c5_fetch.log 2>&1 || { tail -5 gpurun_out/pmc_c5_fetch.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list); dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_c5_fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/pmc_c5_fetch/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
lines = ["rocprofv3 --kernel-trace --pmc FETCH_SIZE (tools/run_pmc_c5.sh), C5: N = 16384(+), grid 131072, means per dispatch;",
         "bytes = FETCH_SIZE (KB) x 1024 x 2 (gfx950: 64 B counted per 128-B request of a 16-B/lane stream)"]
for k in ("abo::cand_gemv_kernel", "abo::trmv_kernel"):
    if k in acc:
        kb = sum(acc[k]) / len(acc[k]); d = sum(dur[k]) / len(dur[k]) / 1e3
        lines.append(f"{k}: n={len(acc[k])} avg {d:.1f} us  FETCH_SIZE {kb:.6g} KB -> {kb*2048/1e9:.2f} GB per launch, {kb*2048/(d*1e-6)/1e12:.2f} TB/s")
open("gpurun_out/pmc_c5_fetch_summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
import hashlib, json
k = "abo::cand_gemv_kernel"
if k in acc:
    kb = sum(acc[k]) / len(acc[k]); d = sum(dur[k]) / len(dur[k]) / 1e6
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE, tools/run_pmc_c5.sh", "kernel": "cand_gemv_kernel", "N": 16384, "M": 131072,
               "kernel_source": "abstractbayesopt.jl_amd/csrc/misc.hip",
               "kernel_sources": ["misc.hip", "abo_kernels.h", "abo_kappa.h"],     # = bench.py PMC_SOURCES["c5"]
               "kernel_source_sha": hashlib.sha256(b"".join(open("abstractbayesopt.jl_amd/csrc/" + n, "rb").read() for n in ("misc.hip", "abo_kernels.h", "abo_kappa.h"))).hexdigest()[:16],
               "FETCH_SIZE_KB_mean": kb, "correction": "gfx950: x2 (64 B counted per 128-B request of a 16-B/lane stream)",
               "traffic_bytes_per_launch": kb * 2048, "algorithmic_bytes_per_launch": 8.0 * 16384 * 131072,
               "avg_launch_ms_under_pmc": d}, open("gpurun_out/c5_pmc_traffic.json", "w"), indent=1)
PY
