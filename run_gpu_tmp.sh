mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 500 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cd /tmp && export TMPDIR=/tmp
timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b1 -o b1 -- python $R/bench.py --no-cpu-baseline --no-batch32 > $R/gpurun_out/bench_under_rocprof.json 2>/dev/null
cp /tmp/prof_b1/*kernel_stats.csv $R/gpurun_out/b1_kernel_stats.csv 2>/dev/null
cd $R
python -c "
import json; d=json.load(open('gpurun_out/bench_default.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_us'], d['batch32']['value'], d['prefill']['value'])
for g in d['roofline']['groups']: print(g['kernel'], g['avg_us'])"
head -8 gpurun_out/b1_kernel_stats.csv | cut -c1-120
