#!/bin/bash
# scratch: profile the hand-written prefill GEMM (kernel stats + SQ counters)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/call_d
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
PF_T=2048 PF_MODES=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_pf --output-format csv -- python tests/bench_prefill.py > $OUT/pf_stats.log 2>&1
cp $(find /tmp/rp_pf -name "*kernel_stats.csv" | head -1) $OUT/pf_kernel_stats.csv
head -14 $OUT/pf_kernel_stats.csv | cut -c1-160
PF_T=2048 PF_MODES=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d /tmp/pmc_pf --output-format csv -- python tests/bench_prefill.py > $OUT/pf_pmc.log 2>&1
python tools/pmc_summary.py /tmp/pmc_pf $OUT/pf_pmc_sq.json > $OUT/pf_pmc_summary.txt 2>&1
grep -E "qpg_|prefill_attn" $OUT/pf_pmc_summary.txt | cut -c1-600
PF_T=2048 PF_MODES=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM -d /tmp/pmc_pf2 --output-format csv -- python tests/bench_prefill.py > $OUT/pf_pmc2.log 2>&1
python tools/pmc_summary.py /tmp/pmc_pf2 $OUT/pf_pmc_sq2.json > $OUT/pf_pmc_summary2.txt 2>&1
grep -E "qpg_" $OUT/pf_pmc_summary2.txt | cut -c1-600
tail -3 $OUT/pf_pmc2.log
