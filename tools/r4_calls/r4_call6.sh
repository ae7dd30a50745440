#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4c6
mkdir -p $OUT
cd $R
timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_prefill.py -m gpu -q -x -k "prompt or prefill or reference_numerics or gemm or moe" > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "faithful or prompt_step" > $OUT/pytest_faithful.log 2>&1
grep -E "max_rel_err|passed|failed|Error|assert" $OUT/pytest_faithful.log | cut -c1-600
PF_T=2048 PF_MODES=1 PF_QPG=2,0,2 timeout 300 python tests/bench_prefill.py 2>&1 | grep "tok/s" > $OUT/prefill_ab.log; cat $OUT/prefill_ab.log
timeout 600 python bench_legs.py engine_b32 > $OUT/engine.log 2>&1; tail -1 $OUT/engine.log | cut -c1-2500
