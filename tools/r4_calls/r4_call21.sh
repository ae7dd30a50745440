#!/bin/bash
# wide GEMM image ring: three stages (default build) against four (build_probe/libmi355vllm_lead3.so, -DQW1_LEAD=3)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4c21
mkdir -p $OUT
cd $R
L3=$R/build_probe/libmi355vllm_lead3.so
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_prefill.py -m gpu -q -x > $OUT/pytest_default.log 2>&1
grep -n "passed\|failed" $OUT/pytest_default.log | tail -1
MI355_LIB_PATH=$L3 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x > $OUT/pytest_lead3.log 2>&1
grep -n "passed\|failed" $OUT/pytest_lead3.log | tail -1
for i in 1 2 3; do
  B32_STEPS=16 timeout 120 python tools/exp_b32.py 2>&1 | grep -o "'value': [0-9.]*" | sed "s/^/default b32 /" | tee -a $OUT/ab.log
  MI355_LIB_PATH=$L3 B32_STEPS=16 timeout 120 python tools/exp_b32.py 2>&1 | grep -o "'value': [0-9.]*" | sed "s/^/lead3   b32 /" | tee -a $OUT/ab.log
done
