"""Per-launch timeline of the LAST fit in a rocprofv3 kernel trace: start offset, duration, gap to the previous kernel,
kernel name, grid.  usage: trace_summary.py kernel_trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last fit starts at the last scale_points_kernel
starts = [i for i, r in enumerate(rows) if "scale_points" in r["Kernel_Name"]]
seg = rows[starts[-1]:]
t0 = int(seg[0]["Start_Timestamp"])
prev_end = t0
tot = {}
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("abo::", "").replace("void ", "")
    grid = "x".join(str(int(r[k]) // max(1, int(r[w]))) for k, w in (("Grid_Size_X", "Workgroup_Size_X"), ("Grid_Size_Y", "Workgroup_Size_Y"), ("Grid_Size_Z", "Workgroup_Size_Z")))
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {name:28s} wg {grid}  q{r.get('Queue_Id', '?')}")
    prev_end = max(prev_end, e)
    d = tot.setdefault(name, [0, 0.0]); d[0] += 1; d[1] += (e - s) / 1e3
print(f"TOTAL span {(prev_end - t0) / 1e3:.1f} us")
for k, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:30s} n={n:4d} sum={t:9.1f} us")
