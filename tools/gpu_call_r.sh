#!/bin/bash
# SQ counter passes (three --pmc sets) over one command; CMD / TAG from the environment
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/call_r
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $R
CMD=${CMD:-"python tools/exp_b32.py"}
TAG=${TAG:-b32}
FILT=${FILT:-"qmm_|paged_attn"}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" \
           "SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  B32_STEPS=3 timeout 400 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_${TAG}_$i --output-format csv -- $CMD > $OUT/pmc_${TAG}_$i.log 2>&1
  python tools/pmc_summary.py /tmp/pmc_${TAG}_$i $OUT/pmc_${TAG}_$i.json | grep -E "$FILT" | cut -c1-520
done
