mkdir -p gpurun_out
timeout 500 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python -c "
import json; d=json.load(open('gpurun_out/bench_default.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['batch32']['value'], d['prefill'], d['cpu_baseline']['value'])"
