#!/bin/bash
# gpurun --timeout 1500 -- 'bash tools/gpu_fullsize_call.sh'   : the full-size parity test (+ whatever else is being validated)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/fullsize
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_linear.py -m gpu -q -x > $OUT/linear.log 2>&1
tail -5 $OUT/linear.log
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s > $OUT/pytest.log 2>&1
grep -o "{'[a-z_]*': [^}]*}" $OUT/pytest.log | cut -c1-700
grep -E "passed|failed" $OUT/pytest.log | tail -1
