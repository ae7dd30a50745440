mkdir -p gpurun_out
python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -3
python bench.py --steps 64 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'],'tok/s', d['ms_per_step'],'ms', d['step']['achieved_GBs'],'GB/s')
for g in d['roofline']['groups']: print(g)"
