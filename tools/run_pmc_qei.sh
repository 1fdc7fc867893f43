# PMC passes over the block pass of greedy q-EI (qei_pass_kernel) at config-5 size: separate runs, --kernel-trace only
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r05qei}
export QEI_AB_VARIANTS=${2:-pass}
export QEI_AB_T=${3:-32}
run() {
  name=$1; shift
  rm -rf gpurun_out/pmc_${TAG}_$name
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${TAG}_$name -- python3 tools/qei_pass_ab.py > gpurun_out/pmc_${TAG}_$name.log 2>&1 || { tail -5 gpurun_out/pmc_${TAG}_$name.log; exit 1; }
  echo "pass $name done"
}
run sq SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
run grbm GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run fetch FETCH_SIZE
python3 tools/pmc_summary.py gpurun_out $TAG > gpurun_out/pmc_${TAG}_summary.txt
grep -A14 "qei_pass\|gemm_skinny" gpurun_out/pmc_${TAG}_summary.txt | head -60
