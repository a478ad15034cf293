#!/bin/bash
# Instruction-mix / memory-pipeline counters of fs_encode_streams on the PPMd micro-benchmark (3072 copies of one
# 100 k-symbol quality stream = one full wave set).  Separate rocprofv3 --pmc passes, nothing else traced.
set -u
tag=${1:-r01}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp COPIES=3072
mkdir -p gpurun_out
rocprofv3 -L > gpurun_out/${tag}_counters_list.txt 2>&1
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $line --output-format csv -d gpurun_out/${tag}_mb_pass$i -- python3 tools/ppmd_microbench.py 100000 > gpurun_out/${tag}_mb_pass$i.log 2> gpurun_out/${tag}_mb_pass$i.err
  python3 tools/pmc_summary.py pmc gpurun_out/${tag}_mb_pass$i > gpurun_out/${tag}_mb_pass${i}_summary.json
  rm -rf gpurun_out/${tag}_mb_pass$i
done <<'PASSES'
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_VALU_MFMA_I8 SQ_IFETCH
TA_TA_BUSY_sum TA_BUSY_avr
TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
GRBM_GUI_ACTIVE GRBM_COUNT
PASSES
grep -h "copies" gpurun_out/${tag}_mb_pass*.log | head -3
