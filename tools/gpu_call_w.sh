#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/call_w
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $R

timeout 400 rocprofv3 --kernel-trace -d /tmp/rp_w --output-format csv -- python bench.py --steps 4 --warmup 1 --no-batch32 --no-cpu-baseline --parity off --legs ${LEGS:-gptq_qwen2} > $OUT/bench.json 2>/dev/null
python tools/trace_groups.py $(find /tmp/rp_w -name "*kernel_trace.csv" | head -1) | grep -v "at::native" | head -22
python - <<'P'
import json
d=json.loads(open('gpurun_out/call_w/bench.json').read().strip().splitlines()[-1])
for k,v in d.get('configs',{}).items(): print(k, v.get('value'), v.get('ms_per_step'))
P
