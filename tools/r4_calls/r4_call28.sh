#!/bin/bash
# 16-bit prompt GEMM: 256 x 128 tile (8 waves, tuning key 30 bit 32) against the 128 x 128 tile; bf16 prompt step of Llama-3-8B, T = 2048
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4c28
mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests/test_gpu_linear.py -m gpu -q -x -k "tall_tile or matches_oracle or gate_up" > $OUT/pytest.log 2>&1
grep "passed\|failed" $OUT/pytest.log | tail -1; grep -n "Error\|assert " $OUT/pytest.log | head -5
for i in 1 2; do
  timeout 200 python bench_legs.py bf16_prompt --no-parity 2>/dev/null | grep '^{' | tail -1 | grep -o '"value": [0-9.]*' | sed 's/^/128x128 /' | tee -a $OUT/ab.log
  MI355_TUNING=30:32 timeout 200 python bench_legs.py bf16_prompt --no-parity 2>/dev/null | grep '^{' | tail -1 | grep -o '"value": [0-9.]*' | sed 's/^/256x128 /' | tee -a $OUT/ab.log
done
