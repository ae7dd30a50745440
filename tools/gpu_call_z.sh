#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd $R
for v in ${VARS:-new prev}; do
  if [ $v = new ]; then L=""; else L=$R/ab_lib/$v.so; fi
  MI355_LIB_PATH=$L timeout 300 rocprofv3 --kernel-trace -d /tmp/rp_z_$v --output-format csv -- python bench.py --steps 32 --warmup 4 --no-batch32 --no-cpu-baseline --parity off --legs none > /tmp/bench_$v.json 2>/dev/null
  echo "== $v $(grep -o '"value": [0-9.]*' /tmp/bench_$v.json | head -1)"
  python tools/trace_groups.py $(find /tmp/rp_z_$v -name "*kernel_trace.csv" | head -1) | grep -E "qmm_kernel|paged" | head -12
done
