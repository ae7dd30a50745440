#!/bin/bash
# rocprofv3 kernel stats of the batch-1 GGUF bench loop: tools/prof_b1.sh <out-name> [head-lines]  (MI355_TUNING=key:value,... applies)
R=${GRAFT_REPO_ROOT:-$PWD}
NAME=${1:-b1}
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$NAME
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$NAME --output-format csv -- python $R/bench.py --no-batch32 --no-cpu-baseline --parity off --legs none > /tmp/$NAME.log 2>&1
tail -1 /tmp/$NAME.log | cut -c1-200
f=$(find /tmp/prof_$NAME -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $R/gpurun_out/prof/${NAME}_kernel_stats.csv; head -${2:-10} "$f" | cut -c1-150; fi
