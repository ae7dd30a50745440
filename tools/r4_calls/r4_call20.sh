#!/bin/bash
# wide GEMM: three-stage image ring only (weights in registers, one tile ahead): build_probe/libmi355vllm_img3.so (-DQW1_LEAD=2 -DQW1_WLDS=0)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4c20
mkdir -p $OUT
cd $R
IMG3=$R/build_probe/libmi355vllm_img3.so
MI355_LIB_PATH=$IMG3 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x > $OUT/pytest_img3.log 2>&1
tail -2 $OUT/pytest_img3.log
for i in 1 2 3; do
  B32_STEPS=16 timeout 120 python tools/exp_b32.py 2>&1 | grep -o "'value': [0-9.]*" | sed "s/^/default b32 /" | tee -a $OUT/ab.log
  MI355_LIB_PATH=$IMG3 B32_STEPS=16 timeout 120 python tools/exp_b32.py 2>&1 | grep -o "'value': [0-9.]*" | sed "s/^/img3    b32 /" | tee -a $OUT/ab.log
done
