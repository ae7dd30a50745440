#!/bin/bash
# round 4, call 2: the suite with the new defaults + the Q6_K read fix; k-split A/B at batch 32; the bench line
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4c2
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
B32_STEPS=16 B32_AB="10=1024;10=-256;10=-512;10=1024;10=-256;10=-512" timeout 200 python tools/exp_b32.py 2>&1 | grep "tok/s" > $OUT/b32_ks.log
cat $OUT/b32_ks.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
head -c 1500 $OUT/bench.json; tail -3 $OUT/bench.err
