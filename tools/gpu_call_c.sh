#!/bin/bash
# scratch: hand-written prefill GEMM: per-op parity, model parity, full-size prompt parity, speed
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/call_c
mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "qmatmul or matmul or fused" > $OUT/ops.log 2>&1
tail -12 $OUT/ops.log | cut -c1-300
timeout 300 python -m pytest tests/test_gpu_model.py tests/test_gpu_prefill.py tests/test_gpu_engine.py -m gpu -q > $OUT/model.log 2>&1
tail -6 $OUT/model.log | cut -c1-300
timeout 300 python -m pytest tests/test_gpu_tp2.py -m gpu -q > $OUT/tp2.log 2>&1
tail -4 $OUT/tp2.log | cut -c1-300
timeout 400 python tests/bench_prefill.py > $OUT/bench_prefill.log 2>&1
cat $OUT/bench_prefill.log | grep prefill
MI355_FULLSIZE_PROMPT_T=1024 timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k prompt > $OUT/fullsize_prompt.log 2>&1
grep -o "{'tokens[^}]*}" $OUT/fullsize_prompt.log; tail -2 $OUT/fullsize_prompt.log
