#!/bin/bash
# rocprofv3 kernel stats of one bench_legs.py leg: tools/prof_leg.sh <leg> <out-name>  (run on the GPU box through gpurun)
R=${GRAFT_REPO_ROOT:-$PWD}
LEG=${1:-gptq_qwen2}; NAME=${2:-leg}
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$NAME
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$NAME --output-format csv -- python $R/bench_legs.py $LEG --no-parity > /tmp/$NAME.log 2>&1
tail -1 /tmp/$NAME.log | cut -c1-330
f=$(find /tmp/prof_$NAME -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $R/gpurun_out/prof/${NAME}_kernel_stats.csv; head -${3:-12} "$f" | cut -c1-170; fi
