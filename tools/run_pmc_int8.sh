# PMC passes for the int8 engine's residue GEMM under the C3 bench step (separate runs, --kernel-trace only, as the pool requires)
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r02i}
run() {  # name, counters...
  name=$1; shift
  rm -rf gpurun_out/pmc_${TAG}_$name
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${TAG}_$name -- python3 bench.py --config c3 --contraction int8 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > gpurun_out/pmc_${TAG}_$name.log 2>&1 || { tail -5 gpurun_out/pmc_${TAG}_$name.log; exit 1; }
  echo "pass $name done"
}
run sq SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
run grbm GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 tools/pmc_summary.py gpurun_out $TAG > gpurun_out/pmc_${TAG}_summary.txt
grep -A22 "oz_gemm16p" gpurun_out/pmc_${TAG}_summary.txt | head -24
