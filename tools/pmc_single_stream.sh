#!/bin/bash
# PMC passes over fs_encode_streams on ONE long quality stream (the per-stream latency case).
#   tools/pmc_single_stream.sh <tag> [symbols]
set -u
tag=${1:-r02}; n=${2:-3000000}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp COPIES=1
mkdir -p gpurun_out
i=0
for line in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
            "SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_INST_CYCLES_SALU SQ_INSTS_VALU_MFMA_I8 SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $line --output-format csv -d gpurun_out/${tag}_ss_pass$i -- python3 tools/ppmd_microbench.py $n > gpurun_out/${tag}_ss_pass$i.log 2> gpurun_out/${tag}_ss_pass$i.err
  python3 tools/pmc_summary.py pmc gpurun_out/${tag}_ss_pass$i > gpurun_out/${tag}_ss_pass${i}_summary.json
  rm -rf gpurun_out/${tag}_ss_pass$i
  cat gpurun_out/${tag}_ss_pass${i}_summary.json
done
grep -h "copies" gpurun_out/${tag}_ss_pass*.log | head -3
