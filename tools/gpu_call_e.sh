#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/call_e
mkdir -p $OUT
cd $R
PF_T=512,2048,4096 PF_MODES=1,2 PF_QPG=0 timeout 400 python tests/bench_prefill.py > $OUT/bench_prefill.log 2>&1
grep prefill $OUT/bench_prefill.log
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_prefill.py tests/test_gpu_engine.py -m gpu -q > $OUT/ops.log 2>&1
tail -3 $OUT/ops.log | cut -c1-300
MI355_FULLSIZE_PROMPT_T=512 timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k prompt > $OUT/fullsize_prompt.log 2>&1
grep -o "{'tokens[^}]*}" $OUT/fullsize_prompt.log; tail -1 $OUT/fullsize_prompt.log
cd /tmp && export TMPDIR=/tmp && cd $R
PF_T=2048 PF_MODES=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_pf --output-format csv -- python tests/bench_prefill.py > $OUT/pf_stats.log 2>&1
cp $(find /tmp/rp_pf -name "*kernel_stats.csv" | head -1) $OUT/pf_kernel_stats.csv
head -8 $OUT/pf_kernel_stats.csv | cut -c1-140
PF_T=2048 PF_MODES=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d /tmp/pmc_pf --output-format csv -- python tests/bench_prefill.py > $OUT/pf_pmc.log 2>&1
python tools/pmc_summary.py /tmp/pmc_pf $OUT/pf_pmc_sq.json > $OUT/pf_pmc_summary.txt 2>&1
grep -E "qpg_gemm" $OUT/pf_pmc_summary.txt | cut -c1-500
