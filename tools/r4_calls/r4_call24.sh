#!/bin/bash
# token-split attention stream: ring of 2 / two workgroups per CU (default) against rings of 3 and 4 with one workgroup per CU
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4c24
mkdir -p $OUT
cd $R
for v in ts_r3 ts_r4; do
  MI355_LIB_PATH=$R/build_probe/libmi355vllm_$v.so timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "stream" > $OUT/pytest_$v.log 2>&1
  grep -n "passed\|failed" $OUT/pytest_$v.log | tail -1
done
for i in 1 2 3; do
  for v in default ts_r3 ts_r4; do
    L=$R/build_probe/libmi355vllm_$v.so; [ $v = default ] && L=""
    MI355_LIB_PATH=$L B32_STEPS=16 timeout 120 python tools/exp_b32.py 2>&1 | grep -o "'value': [0-9.]*" | sed "s/^/$v b32 /" | tee -a $OUT/ab.log
  done
done
