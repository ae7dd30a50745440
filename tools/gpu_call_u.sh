#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/call_u
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_engine.py -q -m gpu > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -3
TUNES="0=0;0=0" PF_BATCHES=1 NO_PROBE=1 timeout 300 python tools/exp_decode_sweep.py 2>&1 | grep mode=
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_u --output-format csv -- python bench.py --steps 32 --warmup 4 --no-batch32 --no-cpu-baseline --parity off --legs none > $OUT/bench.json 2>/dev/null
grep -E "reduce|paged_attn_mfma" $(find /tmp/rp_u -name "*kernel_stats.csv" | head -1) | cut -c1-200
