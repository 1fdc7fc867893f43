"""profiles/<tag>_c3_pmc_traffic.json from a pmc summary (tools/pmc_summary.py output): per-launch HBM-side traffic of
the dominant kernel with the gfx950 correction the micro-architecture guide prescribes (FETCH_SIZE counts 64 B per
128-B request of a 16-B/lane stream -> x2; WRITE_SIZE exact; both in KB), plus MFMA-busy fraction and held clock.
usage: python tools/pmc_traffic_json.py <summary.txt> <N> <Mc_per_launch> [int8 <moduli>] > profiles/r01_c3_pmc_traffic.json
With `int8 <moduli>` the kernel is the residue GEMM of the int8 engine (csrc/ozaki.hip)."""
import hashlib, json, os, re, sys

path, N, Mc = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
INT8 = len(sys.argv) > 4 and sys.argv[4] == "int8"
NMOD = int(sys.argv[5]) if INT8 else 0
PREFIX, SRC = ("abo::oz_gemm16p", "ozaki.hip") if INT8 else ("abo::var_gemm", "gemm.hip")
# every file the kernel is compiled from — the same lists as bench.py's PMC_SOURCES (its staleness guard compares the hash)
SRCS = ["ozaki.hip", "abo_oz_dev.h", "abo_kernels.h"] if INT8 else ["gemm.hip", "abo_kernels.h"]
blocks, cur = {}, None
for line in open(path):
    if line.startswith("== "):
        cur = line[3:].strip()
        blocks[cur] = {}
    elif line.startswith("dur["):
        m = re.match(r"dur\[(.*?)\] (.*): n=\d+ mean=([0-9.]+) us", line)
        if m:
            blocks.setdefault("_dur", {})[(m.group(1), m.group(2))] = float(m.group(3))
    elif cur and "mean=" in line:
        name = line.split()[0]
        blocks[cur][name] = float(re.search(r"mean=([0-9.e+-]+)", line).group(1))
kern = next(k for k in blocks if k.startswith(PREFIX))
c = blocks[kern]
fetch_kb, write_kb = c["FETCH_SIZE"], c["WRITE_SIZE"]
traffic = fetch_kb * 1024 * 2 + write_kb * 1024
busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0)      # per-SIMD busy / per-XCD active
extra = {}
if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
    extra["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
    if k in c and "SQ_WAVE_CYCLES" in c:
        extra[k.lower() + "_frac_of_wave_cycles"] = c[k] / c["SQ_WAVE_CYCLES"]
durs = [v for (p, k), v in blocks.get("_dur", {}).items() if k == kern]
dur_us = sum(durs) / len(durs) if durs else None
def _sha(names):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for name in names:
        h.update(open(os.path.join(root, "abstractbayesopt.jl_amd", "csrc", name), "rb").read())
    return h.hexdigest()[:16]


out = {
    "kernel_source": "abstractbayesopt.jl_amd/csrc/" + SRC, "kernel_sources": SRCS, "kernel_source_sha": _sha(SRCS),   # bench.py drops the figure when the file changes
    "source": f"{path} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ / GRBM, separate passes, tools/run_pmc.sh)",
    "kernel": kern.replace("abo::", ""), "N": N, "Mc_per_launch": Mc,
    "FETCH_SIZE_KB_mean": fetch_kb, "WRITE_SIZE_KB_mean": write_kb,
    "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request of a 16-B/lane stream -> x2 (MI355X_MICROARCH.md HBM section); WRITE_SIZE exact",
    "traffic_bytes_per_launch": traffic, "traffic_bytes_per_candidate": traffic / Mc,
    # fp64 engine: W (lower half) + K_XZ chunk + partial sums; int8 engine: residue planes of W (lower half) and of the chunk read
    # once, U written once — n bytes per entry
    "algorithmic_bytes_per_launch": (NMOD * (N * N / 2 + 2.0 * Mc * N)) if INT8 else 8.0 * (N * N / 2 + Mc * N + (N / 128) * Mc),
    "avg_launch_ms_under_pmc": dur_us / 1e3 if dur_us else None,
    "mfma_busy_frac": busy,
    "clock_ghz": (c["GRBM_GUI_ACTIVE"] / 8.0) / (dur_us * 1e3) if dur_us else None,
    "lds_bank_conflict": c.get("SQ_LDS_BANK_CONFLICT"),
}
out.update(extra)
print(json.dumps(out, indent=1))
