#!/bin/bash
# Library with the probe / ablation modes and the per-wave timestamp hooks compiled in (mi355_set_tuning(2, v),
# mi355_debug_set_timestamps); the production library has none of them.  Run after `python __graft_entry__.py`:
#   bash tools/build_probe_lib.sh && MI355_LIB_PATH=$PWD/build_probe/libmi355vllm_probes.so python tools/exp_wave_times.py
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/build_probe
cp $R/candle_vllm_amd/csrc/qmatmul.hip $R/candle_vllm_amd/csrc/qmatmul_probe_tmp.hip   # (same directory: the .inc files resolve)
trap "rm -f $R/candle_vllm_amd/csrc/qmatmul_probe_tmp.hip" EXIT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fopenmp -Wno-unused-value -DMI355_QMM_PROBES -DMI355_QMM_TIMESTAMPS -I$R/include \
      -c $R/candle_vllm_amd/csrc/qmatmul_probe_tmp.hip -o $R/build_probe/qmatmul.o
objs=$(ls $R/build/*.o | grep -v "/qmatmul.o")
hipcc --offload-arch=gfx950 -shared -fPIC -fopenmp -o $R/build_probe/libmi355vllm_probes.so $objs $R/build_probe/qmatmul.o -ldl
echo built $R/build_probe/libmi355vllm_probes.so
