mkdir -p gpurun_out
timeout 200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -8
timeout 200 python bench.py --steps 64 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'],'tok/s', d['ms_per_step'],'ms'); print(d['batch32'])
for g in d['roofline']['groups'][1:2]: print(g)"
