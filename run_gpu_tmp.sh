mkdir -p gpurun_out
timeout 280 python bench.py > gpurun_out/bench2.json 2> gpurun_out/bench2.err; tail -2 gpurun_out/bench2.err
python -c "
import json
d=json.load(open('gpurun_out/bench2.json')); print(d['value'],'tok/s', d['ms_per_step'],'ms'); print(d['batch32']); print(d['cpu_baseline']); print(d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['frac'])"
