#!/bin/bash
# A second build of the library with extra -D switches on the mat-mul translation unit, for same-box A/B measurements:
#   bash tools/build_variant_lib.sh lead2 -DQW1_LEAD=2 && MI355_LIB_PATH=$PWD/build_probe/libmi355vllm_lead2.so python tools/exp_b32.py
# (run after `python __graft_entry__.py`; build_probe/ is git-ignored and travels to the GPU box)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
TU=${TU:-qmatmul}                                            # translation unit that takes the switches (TU=paged_attention bash tools/build_variant_lib.sh ...)
mkdir -p $R/build_probe
tmp=$R/candle_vllm_amd/csrc/${TU}_${name}_tmp.hip            # (same directory: the .inc files resolve)
cp $R/candle_vllm_amd/csrc/$TU.hip $tmp
trap "rm -f $tmp" EXIT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fopenmp -Wno-unused-value "$@" -I$R/include -c $tmp -o $R/build_probe/${TU}_$name.o
objs=$(ls $R/build/*.o | grep -v "/$TU.o")
hipcc --offload-arch=gfx950 -shared -fPIC -fopenmp -o $R/build_probe/libmi355vllm_$name.so $objs $R/build_probe/${TU}_$name.o -ldl
rm -f $R/build_probe/${TU}_$name.o
echo built $R/build_probe/libmi355vllm_$name.so
