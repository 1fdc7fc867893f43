# HBM-side bytes of the C5 step's dominant kernel — the ONE pass over the resident K_ZX per block of greedy q-EI picks
# (qei_pass_kernel, gemm.hip) — and of the bordered append's triangular mat-vecs: FETCH_SIZE pass (own run, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r06}
rm -rf gpurun_out/pmc_c5_fetch
timeout -k 10 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_c5_fetch -- python3 bench.py --config c5 --steps 3 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_c5_fetch.log 2>&1 || { tail -5 gpurun_out/pmc_c5_fetch.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections, hashlib, json
acc = collections.defaultdict(list); dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_c5_fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            acc[r["Kernel_Name"].split("(")[0]].append((float(r["Counter_Value"]), r.get("Grid_Size", "")))
for f in glob.glob("gpurun_out/pmc_c5_fetch/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Grid_Size_X", r.get("Grid_Size", ""))))
lines = ["rocprofv3 --kernel-trace --pmc FETCH_SIZE (tools/run_pmc_c5.sh), C5: N = 16384(+), grid 131072, means per dispatch;",
         "bytes = FETCH_SIZE (KB) x 1024 x 2 (gfx950: 64 B counted per 128-B request of a 16-B/lane stream)",
         "qei_passd_kernel<T/16> is launched three times per block: two split-k products against L^-1 / L^-T (grid z > 1, ~1.1 GB each) and the pass over K_ZX (the largest)"]
out = {}
for k in sorted(acc):
    if "qei_pass" in k or "trmv_kernel" in k or "cand_gemv" in k:
        vals = sorted(v for v, _ in acc[k])
        ds = sorted(d for d, _ in dur.get(k, []))
        big = [v for v in vals if v > 0.5 * vals[-1]]               # the pass over K_ZX: the launches with the most traffic
        dbig = ds[-len(big):] if ds else []
        kb = sum(big) / len(big); dm = (sum(dbig) / len(dbig) / 1e3) if dbig else float("nan")
        lines.append(f"{k}: {len(vals)} dispatches; largest class n={len(big)} avg {dm:.1f} us  FETCH_SIZE {kb:.6g} KB -> {kb*2048/1e9:.2f} GB per launch, {kb*2048/(dm*1e-6)/1e12:.2f} TB/s")
        out[k] = (kb, dm)
open("gpurun_out/pmc_c5_fetch_summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
key = [k for k in out if "qei_passd_kernel" in k]
tk = [k for k in out if "trmv_kernel" in k]
if key:
    kb, dm = out[key[0]]
    srcs = ("gemm.hip", "abo_kernels.h")            # = bench.py PMC_SOURCES["c5"]
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE, tools/run_pmc_c5.sh", "kernel": "qei_passd_kernel (the pass over the resident K_ZX)",
               "N": 16384, "M": 131072, "kernel_sources": list(srcs),
               "kernel_source_sha": hashlib.sha256(b"".join(open("abstractbayesopt.jl_amd/csrc/" + n, "rb").read() for n in srcs)).hexdigest()[:16],
               "FETCH_SIZE_KB_mean": kb, "correction": "gfx950: x2 (64 B counted per 128-B request of a 16-B/lane stream)",
               "traffic_bytes_per_launch": kb * 2048, "algorithmic_bytes_per_launch": 8.0 * 16384 * 131072,
               "avg_launch_ms_under_pmc": dm / 1e3,
               # the timed step's dominant kernel (bench.py: roofline of the C5 line): the bordered append's two triangular mat-vecs
               "trmv": ({"kernel": "trmv_kernel", "kernel_sources": ["chol.hip"],
                         "kernel_source_sha": hashlib.sha256(open("abstractbayesopt.jl_amd/csrc/chol.hip", "rb").read()).hexdigest()[:16],
                         "traffic_bytes_per_launch": out[tk[0]][0] * 2048, "avg_launch_ms_under_pmc": out[tk[0]][1] / 1e3}
                        if tk else None)},
              open("gpurun_out/c5_pmc_traffic.json", "w"), indent=1)
PY
