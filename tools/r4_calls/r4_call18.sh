#!/bin/bash
# cheaper Q4_K unpack (high nibbles in place, one shift by 8 per word, scale pairs once per k-block) in the wide GEMM and the LDS-fed prompt GEMM,
# wide consumer loop unrolled by two (no tile copies): the tree's library against the previous commit's (build_probe/libmi355vllm_prev.so)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4c18
mkdir -p $OUT
cd $R
PREV=$R/build_probe/libmi355vllm_prev.so
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_prefill.py -m gpu -q -x > $OUT/pytest_default.log 2>&1
tail -3 $OUT/pytest_default.log
for i in 1 2 3; do
  MI355_LIB_PATH=$PREV B32_STEPS=16 timeout 120 python tools/exp_b32.py 2>&1 | grep -o "'value': [0-9.]*" | sed 's/^/prev    b32 /' | tee -a $OUT/ab.log
  B32_STEPS=16 timeout 120 python tools/exp_b32.py 2>&1 | grep -o "'value': [0-9.]*" | sed 's/^/default b32 /' | tee -a $OUT/ab.log
done
for i in 1 2; do
  MI355_LIB_PATH=$PREV PF_T=2048,4096 PF_MODES=1 timeout 200 python tests/bench_prefill.py 2>&1 | grep "prefill" | sed 's/^/prev    pf /' | tee -a $OUT/ab.log
  PF_T=2048,4096 PF_MODES=1 timeout 200 python tests/bench_prefill.py 2>&1 | grep "prefill" | sed 's/^/default pf /' | tee -a $OUT/ab.log
done
