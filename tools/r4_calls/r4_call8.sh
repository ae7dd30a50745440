#!/bin/bash
# round 4, call 8: the whole suite on the ten-key surface + the new full-size tests
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4c8
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
