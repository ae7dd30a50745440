#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/call_t
mkdir -p $OUT
cd $R
timeout 2400 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log | cut -c1-200
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
