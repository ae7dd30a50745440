#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/call_s
mkdir -p $OUT
cd $R
timeout 200 python tools/exp_chain_det.py 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_engine.py tests/test_gpu_prefill.py -q -m gpu > $OUT/pytest.log 2>&1; grep -E "passed|failed|Error|error" $OUT/pytest.log | tail -5
MODES=${MODES:-"15=0"} bash tools/gpu_call_n.sh | grep -E "^==|value|qmm_gemm"
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -x -m gpu -k "batch32" > $OUT/pytest_full.log 2>&1; grep -E "passed|failed|Error|error|assert" $OUT/pytest_full.log | tail -8
