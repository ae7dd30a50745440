#!/bin/bash
# prompt step: row statistics + image in one launch (default build) against the two launches (build_probe/libmi355vllm_prep2.so, -DQPG_ROWPREP=0)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4c25
mkdir -p $OUT
cd $R
V=$R/build_probe/libmi355vllm_prep2.so
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_prefill.py -m gpu -q -x > $OUT/pytest_default.log 2>&1
grep -n "passed\|failed" $OUT/pytest_default.log | tail -1
for i in 1 2; do
  MI355_LIB_PATH=$V PF_T=512,2048,4096 PF_MODES=1 timeout 200 python tests/bench_prefill.py 2>&1 | grep "prefill" | sed 's/^/prep2   pf /' | tee -a $OUT/ab.log
  PF_T=512,2048,4096 PF_MODES=1 timeout 200 python tests/bench_prefill.py 2>&1 | grep "prefill" | sed 's/^/default pf /' | tee -a $OUT/ab.log
done
