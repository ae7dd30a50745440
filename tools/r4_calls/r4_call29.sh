#!/bin/bash
# 16-bit prompt GEMM: ring of three K steps (tuning key 30 bit 128) at both tile heights against the double buffer; bf16 prompt step, T = 2048
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4c29
mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests/test_gpu_linear.py -m gpu -q -x -k "tall_tile" > $OUT/pytest.log 2>&1
grep "passed\|failed" $OUT/pytest.log | tail -1; grep -n "Error\|assert " $OUT/pytest.log | head -5
for i in 1 2; do
  for kv in 0 128 160; do
    MI355_TUNING=30:$kv timeout 200 python bench_legs.py bf16_prompt --no-parity 2>/dev/null | grep '^{' | tail -1 | grep -o '"value": [0-9.]*' | sed "s/^/key30=$kv /" | tee -a $OUT/ab.log
  done
done
