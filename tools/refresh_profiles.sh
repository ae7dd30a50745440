#!/bin/bash
# Regenerates the judged artefacts under profiles/ on the GPU box (run through gpurun; results land in gpurun_out/prof).
#   1. python bench.py                                   -> r01_bench_default.json
#   2. rocprofv3 --kernel-trace --stats -- bench (B=1)   -> r01_bench_b1_kernel_stats.csv (+ the bench line under rocprof)
#   3. rocprofv3 --kernel-trace --stats -- bench (B=32)  -> r01_bench_b32_kernel_stats.csv
#   4. two --pmc passes (FETCH_SIZE, WRITE_SIZE)         -> r01_pmc_traffic.json   (tools/pmc_traffic.py)
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
timeout 600 python bench.py > $OUT/r01_bench_default.json 2> $OUT/bench_default.err
B1="python bench.py --no-batch32 --no-cpu-baseline"
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/rp_b1 --output-format csv -- $B1 > $OUT/r01_bench_b1_under_rocprof.json 2> $OUT/rp_b1.err
cp $(find /tmp/rp_b1 -name "*kernel_stats.csv" | head -1) $OUT/r01_bench_b1_kernel_stats.csv
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/rp_b32 --output-format csv -- python bench.py --batch 32 --steps 16 --warmup 4 --no-cpu-baseline --no-batch32 > $OUT/r01_bench_b32_under_rocprof.json 2> $OUT/rp_b32.err
cp $(find /tmp/rp_b32 -name "*kernel_stats.csv" | head -1) $OUT/r01_bench_b32_kernel_stats.csv
PMC="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-batch32 --no-graph"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f --output-format csv -- $PMC > $OUT/pmc_f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w --output-format csv -- $PMC > $OUT/pmc_w.log 2>&1
python tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w $OUT/r01_pmc_traffic.json
head -c 1500 $OUT/r01_bench_default.json
head -12 $OUT/r01_bench_b1_kernel_stats.csv | cut -c1-150
head -8 $OUT/r01_bench_b32_kernel_stats.csv | cut -c1-150
