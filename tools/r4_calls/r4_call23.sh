#!/bin/bash
# attention stream: waves split the TOKENS of a stage (default build) against the channel split (build_probe/libmi355vllm_chsplit.so, -DPAS_TS=0)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4c23
mkdir -p $OUT
cd $R
V=$R/build_probe/libmi355vllm_chsplit.so
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_dense_model.py tests/test_gpu_engine.py -m gpu -q -x > $OUT/pytest_default.log 2>&1
grep -n "passed\|failed" $OUT/pytest_default.log | tail -1; grep -n "Error\|assert" $OUT/pytest_default.log | head -5
for i in 1 2 3; do
  MI355_LIB_PATH=$V B32_STEPS=16 timeout 120 python tools/exp_b32.py 2>&1 | grep -o "'value': [0-9.]*" | sed "s/^/chsplit b32 /" | tee -a $OUT/ab.log
  B32_STEPS=16 timeout 120 python tools/exp_b32.py 2>&1 | grep -o "'value': [0-9.]*" | sed "s/^/default b32 /" | tee -a $OUT/ab.log
done
