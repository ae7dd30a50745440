#!/bin/bash
# prompt GEMM: A-fragment prefetch across a bare barrier (pfa) and the 64-token tile everywhere (m4) against the default build
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4c17
mkdir -p $OUT
cd $R
PFA=$R/build_probe/libmi355vllm_pfa.so
M4=$R/build_probe/libmi355vllm_m4.so
MI355_LIB_PATH=$PFA timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_prefill.py -m gpu -q -x > $OUT/pytest_pfa.log 2>&1
tail -3 $OUT/pytest_pfa.log
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "prompt_gemm" > $OUT/pytest_default.log 2>&1
tail -2 $OUT/pytest_default.log
for i in 1 2; do
  PF_T=1024,2048,4096 PF_MODES=1 timeout 200 python tests/bench_prefill.py 2>&1 | grep "prefill" | sed 's/^/default pf /' | tee -a $OUT/ab.log
  MI355_LIB_PATH=$PFA PF_T=1024,2048,4096 PF_MODES=1 timeout 200 python tests/bench_prefill.py 2>&1 | grep "prefill" | sed 's/^/pfa     pf /' | tee -a $OUT/ab.log
  MI355_LIB_PATH=$M4 PF_T=1024 PF_MODES=1 timeout 200 python tests/bench_prefill.py 2>&1 | grep "prefill" | sed 's/^/m4      pf /' | tee -a $OUT/ab.log
done
