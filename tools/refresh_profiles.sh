#!/bin/bash
# Regenerates the judged artefacts under profiles/ on the GPU box (run through gpurun; results land in gpurun_out/prof, copy
# them to profiles/).  Every file carries the commit it was measured on: write it to ./COMMIT_SHA before the call
#   git rev-parse --short HEAD > COMMIT_SHA && gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh'
#   0. two --pmc passes (FETCH_SIZE, WRITE_SIZE)         -> rNN_pmc_traffic.json   (tools/pmc_traffic.py; copied into profiles/ for step 1)
#   1. python bench.py                                   -> rNN_bench_default.json
#   2. rocprofv3 --kernel-trace --stats -- bench (B=1)   -> rNN_bench_b1_kernel_stats.csv (+ the bench line under rocprof)
#   3. rocprofv3 --kernel-trace --stats -- bench (B=32)  -> rNN_bench_b32_kernel_stats.csv
#   5. one --pmc pass of the SQ busy counters            -> rNN_pmc_sq_b1.json     (MFMA-busy / VALU-busy of every kernel)
set -x
RN=${ROUND_TAG:-r03}
R=${GRAFT_REPO_ROOT:-$PWD}
SHA=$(cat $R/COMMIT_SHA 2>/dev/null || echo unknown)
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
echo "$SHA" > $OUT/${RN}_commit.txt
# the two HBM-traffic passes FIRST, and their summary straight into profiles/ of this copy of the tree: bench.py quotes `roofline.traffic`
# only from a pass whose `kernel_source_sha256` matches the sources it runs
PMC="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-batch32 --no-graph --parity off --legs none"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f --output-format csv -- $PMC > $OUT/pmc_f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w --output-format csv -- $PMC > $OUT/pmc_w.log 2>&1
python tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w $OUT/${RN}_pmc_traffic.json $SHA
cp $OUT/${RN}_pmc_traffic.json $R/profiles/${RN}_pmc_traffic.json
timeout 900 python bench.py > $OUT/${RN}_bench_default.json 2> $OUT/bench_default.err
B1="python bench.py --no-batch32 --no-cpu-baseline --parity off --legs none"
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/rp_b1 --output-format csv -- $B1 > $OUT/${RN}_bench_b1_under_rocprof.json 2> $OUT/rp_b1.err
(echo "# commit $SHA : rocprofv3 --kernel-trace --stats -- $B1"; cat $(find /tmp/rp_b1 -name "*kernel_stats.csv" | head -1)) > $OUT/${RN}_bench_b1_kernel_stats.csv
B32="python bench.py --batch 32 --steps 16 --warmup 4 --no-cpu-baseline --no-batch32 --parity off --legs none"
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/rp_b32 --output-format csv -- $B32 > $OUT/${RN}_bench_b32_under_rocprof.json 2> $OUT/rp_b32.err
(echo "# commit $SHA : rocprofv3 --kernel-trace --stats -- $B32"; cat $(find /tmp/rp_b32 -name "*kernel_stats.csv" | head -1)) > $OUT/${RN}_bench_b32_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d /tmp/pmc_sq --output-format csv -- $PMC > $OUT/pmc_sq.log 2>&1
python tools/pmc_summary.py /tmp/pmc_sq $OUT/${RN}_pmc_sq_b1.json $SHA > $OUT/pmc_sq_summary.txt 2>&1
head -c 1800 $OUT/${RN}_bench_default.json
head -14 $OUT/${RN}_bench_b1_kernel_stats.csv | cut -c1-150
head -10 $OUT/${RN}_bench_b32_kernel_stats.csv | cut -c1-150
head -12 $OUT/pmc_sq_summary.txt | cut -c1-400
# 6. prompt-step GEMM: kernel stats + MFMA-busy
PF="python tests/bench_prefill.py"
PF_T=2048 PF_MODES=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_pf --output-format csv -- $PF > $OUT/pf_stats.log 2>&1
(echo "# commit $SHA : PF_T=2048 PF_MODES=1 rocprofv3 --kernel-trace --stats -- $PF"; cat $(find /tmp/rp_pf -name "*kernel_stats.csv" | head -1)) > $OUT/${RN}_prefill_kernel_stats.csv
PF_T=2048 PF_MODES=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d /tmp/pmc_pf --output-format csv -- $PF > $OUT/pf_pmc.log 2>&1
python tools/pmc_summary.py /tmp/pmc_pf $OUT/${RN}_prefill_pmc_sq.json $SHA > $OUT/pf_pmc_summary.txt 2>&1
grep prefill $OUT/pf_stats.log
# 7. ragged batch-32 step: durations per (kernel, grid) from a kernel trace
B32_STEPS=6 timeout 300 rocprofv3 --kernel-trace -d /tmp/rp_b32r --output-format csv -- python tools/exp_b32.py > $OUT/b32_ragged.log 2>&1
(echo "# commit $SHA : B32_STEPS=6 rocprofv3 --kernel-trace -- python tools/exp_b32.py ; tools/trace_groups.py (mean / min us per launch shape)"; grep -o "'value': [0-9.]*" $OUT/b32_ragged.log; python tools/trace_groups.py $(find /tmp/rp_b32r -name "*kernel_trace.csv" | head -1) | grep -v "at::native") > $OUT/${RN}_b32_ragged_launch_groups.txt
# 8. SQ counters of the wide GEMM at batch 32 (two --pmc passes)
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA"; do
  i=$((i+1))
  B32_STEPS=3 timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_b32_$i --output-format csv -- python tools/exp_b32.py > $OUT/pmc_b32_$i.log 2>&1
  python tools/pmc_summary.py /tmp/pmc_b32_$i $OUT/${RN}_pmc_sq_b32_set$i.json $SHA > /dev/null 2>&1
done
# 8b. kernel stats of the secondary legs (bench_legs.py)
for leg in bf16_b32 gptq_qwen2 gptq_qwen2_b32 mixtral_fp8 mixtral_fp8_b32; do
  rm -rf /tmp/rp_leg_$leg
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_leg_$leg --output-format csv -- python bench_legs.py $leg --no-parity > $OUT/leg_$leg.log 2>&1
  f=$(find /tmp/rp_leg_$leg -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && (echo "# commit $SHA : rocprofv3 --kernel-trace --stats -- python bench_legs.py $leg --no-parity ; $(grep '^{' $OUT/leg_$leg.log | tail -1 | grep -o '"value": [0-9.]*' | head -1) under rocprofv3"; head -16 "$f") > $OUT/${RN}_leg_${leg}_kernel_stats.csv
done
[ -n "$SKIP_MICRO" ] && { head -12 $OUT/${RN}_b32_ragged_launch_groups.txt | cut -c1-170; exit 0; }
# 9. micro-benchmarks: bare weight stream of the wide GEMM; per-wave timelines of the single-token launches
[ -x tools/probe_wide_stream.bin ] || hipcc --offload-arch=gfx950 -O3 tools/probe_wide_stream.hip -o tools/probe_wide_stream.bin
(echo "# commit $SHA : tools/probe_wide_stream.bin"; timeout 120 tools/probe_wide_stream.bin) > $OUT/${RN}_probe_wide_stream.txt 2>&1
(echo "# commit $SHA : MI355_LIB_PATH=build_probe/libmi355vllm_probes.so python tools/exp_wave_times.py (probe build: tools/build_probe_lib.sh; us since the first wave's entry; 100 MHz clock)"; MI355_LIB_PATH=$R/build_probe/libmi355vllm_probes.so timeout 300 python tools/exp_wave_times.py 2>&1 | grep -v amdgpu.ids) > $OUT/${RN}_b1_wave_times.txt
head -12 $OUT/${RN}_b32_ragged_launch_groups.txt | cut -c1-170
