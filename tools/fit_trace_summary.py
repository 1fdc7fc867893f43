"""Per-launch summary of a fit's kernel trace: python tools/fit_trace_summary.py <kernel_trace.csv> <N> [fits in the trace]
(rocprofv3 --kernel-trace --output-format csv -- python3 tools/fit_only.py N d reps; the last fit of the trace is summarised:
every gemm_nt launch above 150 µs with its tile count and rate, totals per kernel, idle time between kernels)."""
import csv, sys
path, N = sys.argv[1], int(sys.argv[2])
fits = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rows = list(csv.DictReader(open(path)))
last = rows[-(len(rows) // fits):]
t0 = int(last[0]["Start_Timestamp"])
tot, prev_end, gaps = {}, None, 0.0
print(f"N = {N}: {len(last)} kernels in the last fit of the trace")
for r in last:
    name = r["Kernel_Name"].split("(")[0].replace("abo::", "").replace("void ", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d = (e - s) / 1e3
    tot.setdefault(name, [0, 0.0]); tot[name][0] += 1; tot[name][1] += d
    if prev_end is not None:
        gaps += max(0, s - prev_end) / 1e3
    prev_end = e
    if "gemm_nt_kernel" in name and d > 150:
        gx, gy, gz = int(r["Grid_Size_X"]) // 256, int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])
        print(f"  gemm_nt_kernel grid {gx} x {gy} x {gz}: {d:8.1f} us")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:10]:
    print(f"  {k:44s} {v[0]:4d} launches {v[1]:9.1f} us")
print(f"  idle between kernels {gaps:.1f} us; first start to last end {(int(last[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
