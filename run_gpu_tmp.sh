mkdir -p gpurun_out
timeout 200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -15
timeout 200 python bench.py --batch 32 --ctx 2688 --steps 16 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'],'tok/s', d['ms_per_step'],'ms', d['step'])
for g in d['roofline']['groups']: print(g)"
