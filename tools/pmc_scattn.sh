#!/bin/bash
# PMC passes over one kernel of the encoder (GPU box):  bash tools/pmc_scattn.sh VARIANT TAG [KERNEL_NAME_PATTERN]
# One rocprofv3 run per counter group (counters only with --kernel-trace, as the pool requires); summary by tools/pmc_summary.py
V=${1:-18}; TAG=${2:-pmc}; PAT=${3:-k_scattn}
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
GROUPS_=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL"
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAIT_ANY SQ_INSTS_SALU"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCC_HIT_sum TCC_MISS_sum"
)
i=0
for g in "${GROUPS_[@]}"; do
  out=gpurun_out/${TAG}_g$i
  rm -rf "$out"
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d "$out" -o run -- python3 tools/run_scattn_once.py "$V" > "$out.log" 2>&1 || { echo "group $i failed"; tail -5 "$out.log"; }
  i=$((i+1))
done
python3 tools/pmc_summary.py "gpurun_out/${TAG}_g*" "$PAT" > "gpurun_out/${TAG}_summary.json"
cat "gpurun_out/${TAG}_summary.json"
