#!/bin/bash
# prompt attention through the LDS ring: 64-query workgroups (default; grid now heaviest block first) against 32-query workgroups
# (build_probe/libmi355vllm_pq2.so, -DPFL_QTILES=2) and 32-query workgroups on a ring of two stages (pq2r2)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4c26
mkdir -p $OUT
cd $R
for v in default pq2 pq2r2; do
  L=$R/build_probe/libmi355vllm_$v.so; [ $v = default ] && L=""
  MI355_LIB_PATH=$L timeout 600 python -m pytest tests/test_gpu_prefill.py tests/test_gpu_model.py -m gpu -q -x > $OUT/pytest_$v.log 2>&1
  echo $v $(grep "passed\|failed" $OUT/pytest_$v.log | tail -1)
done
for i in 1 2; do
  for v in default pq2 pq2r2; do
    L=$R/build_probe/libmi355vllm_$v.so; [ $v = default ] && L=""
    MI355_LIB_PATH=$L PF_T=2048,4096 PF_MODES=1 timeout 200 python tests/bench_prefill.py 2>&1 | grep "prefill" | sed "s/^/$v pf /" | tee -a $OUT/ab.log
  done
done
