#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/call_v
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $R
for m in ${MODES:-"18=1" "18=0"}; do
MI355_TUNE=$m timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_v_$m --output-format csv -- python bench.py --steps 32 --warmup 4 --no-batch32 --no-cpu-baseline --parity off --legs none > $OUT/bench_$m.json 2>/dev/null
echo "== $m"; grep -o '"value": [0-9.]*' $OUT/bench_$m.json | head -1
grep -E "qmm_|paged" $(find /tmp/rp_v_$m -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4,6 | cut -c1-150
done
