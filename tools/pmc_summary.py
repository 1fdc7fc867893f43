"""Summarise rocprofv3 --pmc CSVs per kernel: mean counter value per dispatch."""
import csv, glob, os, sys, collections
root, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sorted(glob.glob(os.path.join(root, f"pmc_{tag}_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[(os.path.basename(d), r["Kernel_Name"].split("(")[0])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in sorted(acc, key=lambda k: -sum(sum(v) for v in acc[k].values())):
    print(f"== {k}")
    for c, v in sorted(acc[k].items()):
        print(f"   {c:34s} n={len(v):5d} mean={sum(v)/len(v):.6g} sum={sum(v):.6g}")
for (p, k), v in sorted(dur.items()):
    if "var_gemm" in k or "kgen" in k or "chol" in k or "oz_" in k:
        print(f"dur[{p}] {k}: n={len(v)} mean={sum(v)/len(v)/1e3:.1f} us")
