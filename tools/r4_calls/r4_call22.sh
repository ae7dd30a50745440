#!/bin/bash
# wide GEMM: weight tiles two ahead in registers (build_probe/libmi355vllm_wpf2.so, -DQW1_WPF=2) on top of the three-stage image ring
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4c22
mkdir -p $OUT
cd $R
V=$R/build_probe/libmi355vllm_wpf2.so
MI355_LIB_PATH=$V timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x > $OUT/pytest_wpf2.log 2>&1
grep -n "passed\|failed" $OUT/pytest_wpf2.log | tail -1
for i in 1 2 3; do
  B32_STEPS=16 timeout 120 python tools/exp_b32.py 2>&1 | grep -o "'value': [0-9.]*" | sed "s/^/default b32 /" | tee -a $OUT/ab.log
  MI355_LIB_PATH=$V B32_STEPS=16 timeout 120 python tools/exp_b32.py 2>&1 | grep -o "'value': [0-9.]*" | sed "s/^/wpf2    b32 /" | tee -a $OUT/ab.log
done
