#!/bin/bash
# last call of the round (commit f65d023): targeted GPU tests on the two-workgroup prompt GEMM, the traffic passes for bench.py's roofline.traffic,
# the prompt step's kernel stats
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd $R
timeout 120 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_prefill.py -m gpu -q -x > $OUT/pytest_last.log 2>&1
grep "passed\|failed" $OUT/pytest_last.log | tail -1
timeout 150 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "test_prompt_step or chunked_prefill_16k" > $OUT/pytest_last_fullsize.log 2>&1
grep "passed\|failed" $OUT/pytest_last_fullsize.log | tail -1
cd /tmp && export TMPDIR=/tmp && cd $R
PMC="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-batch32 --no-graph --parity off --legs none"
timeout 100 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f --output-format csv -- $PMC > $OUT/pmc_f.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w --output-format csv -- $PMC > $OUT/pmc_w.log 2>&1
python tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w $OUT/r04_pmc_traffic.json f65d023
PF_T=2048 PF_MODES=1 timeout 100 rocprofv3 --kernel-trace --stats -d /tmp/rp_pf --output-format csv -- python tests/bench_prefill.py > $OUT/pf_stats.log 2>&1
(echo "# commit f65d023 : PF_T=2048 PF_MODES=1 rocprofv3 --kernel-trace --stats -- python tests/bench_prefill.py"; cat $(find /tmp/rp_pf -name "*kernel_stats.csv" | head -1)) > $OUT/r04_prefill_kernel_stats.csv
grep prefill $OUT/pf_stats.log; head -5 $OUT/r04_prefill_kernel_stats.csv | cut -c1-140
PF_T=512,1024,2048,4096 PF_MODES=1 timeout 100 python tests/bench_prefill.py 2>&1 | grep "prefill" | tee $OUT/pf_last.log
