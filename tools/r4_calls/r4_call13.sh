#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4c13
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
ROUND_TAG=r04 SKIP_MICRO=1 bash tools/refresh_profiles.sh > $OUT/refresh.log 2>&1
tail -40 $OUT/refresh.log | cut -c1-220
