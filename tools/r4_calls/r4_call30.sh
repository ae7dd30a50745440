#!/bin/bash
# prompt GEMM, 64-token tile as TWO workgroups per CU (qpg_gemm_lds2_kernel; build_probe/libmi355vllm_l2all.so: at every T) against the default
# (one workgroup per CU; 128-token tile from 2048 tokens)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4c30
mkdir -p $OUT
cd $R
V=$R/build_probe/libmi355vllm_l2all.so
MI355_LIB_PATH=$V timeout 200 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "qmatmul_vs_oracle or prompt_gemm" > $OUT/pytest.log 2>&1
grep "passed\|failed" $OUT/pytest.log | tail -1; grep -n "Error\|assert " $OUT/pytest.log | head -5
for i in 1 2; do
  PF_T=512,1024,2048,4096 PF_MODES=1 timeout 200 python tests/bench_prefill.py 2>&1 | grep "prefill" | sed 's/^/default pf /' | tee -a $OUT/ab.log
  MI355_LIB_PATH=$V PF_T=512,1024,2048,4096 PF_MODES=1 timeout 200 python tests/bench_prefill.py 2>&1 | grep "prefill" | sed 's/^/l2all   pf /' | tee -a $OUT/ab.log
done
