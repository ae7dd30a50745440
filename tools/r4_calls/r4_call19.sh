#!/bin/bash
# wide GEMM: which of the two changes costs the 1 %?  prev = neither, u = cheaper unpack only, r = loop unrolled by two only, default = both;
# then the skeleton of the gate/up launch with today's 18 KB image (tools/probe_wide_stream.hip)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4c19
mkdir -p $OUT
cd $R
for i in 1 2 3; do
  for v in prev u r default; do
    L=$R/build_probe/libmi355vllm_$v.so; [ $v = default ] && L=""
    MI355_LIB_PATH=$L B32_STEPS=16 timeout 120 python tools/exp_b32.py 2>&1 | grep -o "'value': [0-9.]*" | sed "s/^/$v b32 /" | tee -a $OUT/ab.log
  done
done
hipcc --offload-arch=gfx950 -O3 tools/probe_wide_stream.hip -o /tmp/probe_wide_stream.bin && timeout 200 /tmp/probe_wide_stream.bin 2>&1 | tee $OUT/probe_wide_stream.txt | tail -16
