# one PMC pass over tools/oz_dev: oz_pmc1.sh <tag> <nmod> <counters...>
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; NMOD=$2; shift; shift
rm -rf gpurun_out/pmc_${TAG}
timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${TAG} -- tools/oz_dev $NMOD > gpurun_out/pmc_${TAG}.log 2>&1 || { tail -5 gpurun_out/pmc_${TAG}.log; exit 1; }
tail -4 gpurun_out/pmc_${TAG}.log
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_${TAG}/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'oz_gemm' in r['Kernel_Name'] and int(r['Grid_Size'])>1000000:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print(f"{k:30s} n={len(v)} mean={sum(v)/len(v):.5g}")
PY
