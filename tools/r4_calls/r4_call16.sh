#!/bin/bash
# prompt GEMM through LDS (default build, Q4_K + Q6_K) vs the 128-token tile variant (build_probe/libmi355vllm_m8.so)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4c16
mkdir -p $OUT
cd $R
M8=$R/build_probe/libmi355vllm_m8.so
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_prefill.py -m gpu -q -x > $OUT/pytest_default.log 2>&1
tail -3 $OUT/pytest_default.log
MI355_LIB_PATH=$M8 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_prefill.py -m gpu -q -x > $OUT/pytest_m8.log 2>&1
tail -3 $OUT/pytest_m8.log
for i in 1 2; do
  PF_T=512,2048,4096 PF_MODES=1 timeout 200 python tests/bench_prefill.py 2>&1 | grep "prefill" | sed 's/^/default pf /' | tee -a $OUT/ab.log
  MI355_LIB_PATH=$M8 PF_T=512,2048,4096 PF_MODES=1 timeout 200 python tests/bench_prefill.py 2>&1 | grep "prefill" | sed 's/^/m8      pf /' | tee -a $OUT/ab.log
done
cd /tmp && export TMPDIR=/tmp
for lib in default m8; do
  L=""; [ $lib = m8 ] && L=$M8
  rm -rf /tmp/prof_pf
  MI355_LIB_PATH=$L PF_T=2048 PF_MODES=1 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_pf --output-format csv -- python $R/tests/bench_prefill.py > /tmp/pf_trace.log 2>&1
  f=$(find /tmp/prof_pf -name "*kernel_stats.csv" | head -1)
  (echo "# $lib"; head -9 "$f" | cut -c1-170) | tee $OUT/pf_stats_$lib.csv
done
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rm -rf /tmp/pmc_pf_$i
  PF_T=2048 PF_MODES=1 timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_pf_$i --output-format csv -- python $R/tests/bench_prefill.py > $OUT/pmc_pf_$i.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pmc_pf_$i $OUT/pf_pmc_set$i.json r4c16 > $OUT/pf_pmc_summary_$i.txt 2>&1
  grep -i "qpg_gemm" $OUT/pf_pmc_summary_$i.txt | cut -c1-400 | head -8
done
