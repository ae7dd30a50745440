#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/call_f
mkdir -p $OUT
cd $R
timeout 600 python bench_legs.py > $OUT/legs.json 2> $OUT/legs.err
cat $OUT/legs.json | cut -c1-2500; tail -3 $OUT/legs.err
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fullsize.py > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log | cut -c1-300
