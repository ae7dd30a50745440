#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/call_i
mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_tp2.py -m gpu -q -k "moe or Moe or mixtral or expert" > $OUT/moe.log 2>&1
tail -3 $OUT/moe.log | cut -c1-200
timeout 300 python bench_legs.py mixtral_fp8 > $OUT/legs.json 2> $OUT/legs.err
cut -c1-700 $OUT/legs.json
