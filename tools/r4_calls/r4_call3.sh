#!/bin/bash
# round 4, call 3: the in-launch split-K epilogue of the wide path + the 16-bit layer's native step driver through the suite; A/B; trace
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4c3
mkdir -p $OUT
cd $R
timeout 700 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
B32_STEPS=16 B32_AB="49=1;49=0;49=1;49=0" timeout 200 python tools/exp_b32.py 2>&1 | grep "tok/s" > $OUT/b32_fuse.log
cat $OUT/b32_fuse.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b32
B32_STEPS=6 timeout 200 rocprofv3 --kernel-trace -d /tmp/prof_b32 --output-format csv -- python $R/tools/exp_b32.py > /tmp/b32_trace.log 2>&1
f=$(find /tmp/prof_b32 -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_groups.py "$f" > $OUT/b32_groups.txt 2>&1
head -30 $OUT/b32_groups.txt
cd $R
timeout 300 python bench_legs.py bf16_b32 --no-parity > $OUT/leg_bf16.log 2>&1; tail -1 $OUT/leg_bf16.log | cut -c1-400
timeout 300 python bench_legs.py gptq_qwen2 --no-parity > $OUT/leg_gptq.log 2>&1; tail -1 $OUT/leg_gptq.log | cut -c1-400
