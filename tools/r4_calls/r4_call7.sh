#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4c7
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "faithful or chunked_prefill_16k" > $OUT/pytest_fullsize.log 2>&1
grep -E "max_rel_err|passed|failed|Error|assert|chunks" $OUT/pytest_fullsize.log | cut -c1-700
timeout 900 python bench_legs.py mixtral_prompt_16k engine_b32 > $OUT/legs1.log 2>&1; tail -1 $OUT/legs1.log | cut -c1-3000
timeout 900 python bench_legs.py gptq_qwen2_b32 > $OUT/legs2.log 2>&1; tail -1 $OUT/legs2.log | cut -c1-1500
