#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4c5
mkdir -p $OUT
cd $R
timeout 400 python -m pytest tests/test_gpu_engine.py tests/test_gpu_dense_model.py tests/test_gpu_ops.py -m gpu -q -x -k "engine or loop or fused_epilogue or wide" > $OUT/pytest.log 2>&1
tail -12 $OUT/pytest.log
B32_STEPS=16 timeout 100 python tools/exp_b32.py 2>&1 | grep "value" | cut -c1-200 > $OUT/b32.log; cat $OUT/b32.log
PF_T=2048 PF_MODES=1 PF_QPG=0,2,0,2 timeout 300 python tests/bench_prefill.py 2>&1 | grep "tok/s" > $OUT/prefill_ab.log
PF_T=4096 PF_MODES=1 PF_QPG=0,2 timeout 300 python tests/bench_prefill.py 2>&1 | grep "tok/s" >> $OUT/prefill_ab.log
cat $OUT/prefill_ab.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_pf
PF_T=2048 PF_MODES=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_pf --output-format csv -- python $R/tests/bench_prefill.py > /tmp/pf.log 2>&1
f=$(find /tmp/prof_pf -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/prefill_kernel_stats.csv; head -12 "$f" | cut -c1-180
cd $R
timeout 600 python bench_legs.py engine_b32 > $OUT/engine.log 2>&1; tail -1 $OUT/engine.log | cut -c1-1800
