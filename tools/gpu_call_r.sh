#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/call_r
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $R
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_WAVES"; do
  i=$((i+1))
  MI355_TUNE=${MI355_TUNE:-} B32_STEPS=3 timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_$i --output-format csv -- python tools/exp_b32.py > $OUT/pmc_$i.log 2>&1
  python tools/pmc_summary.py /tmp/pmc_$i $OUT/pmc_$i.json | grep -E "qmm_gemm" | cut -c1-600
done
