# PMC passes for the dominant kernel (separate runs, --kernel-trace only, as the pool requires)
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r01}
CFG=${2:-c3}
run() {  # name, counters...
  name=$1; shift
  rm -rf gpurun_out/pmc_${TAG}_$name
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${TAG}_$name -- python3 bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > gpurun_out/pmc_${TAG}_$name.log 2>&1 || { tail -5 gpurun_out/pmc_${TAG}_$name.log; exit 1; }
  echo "pass $name done"
}
run sq SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
run grbm GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS SQ_INSTS_VMEM_RD
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 tools/pmc_summary.py gpurun_out $TAG > gpurun_out/pmc_${TAG}_summary.txt
cat gpurun_out/pmc_${TAG}_summary.txt
