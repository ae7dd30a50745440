mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | grep -E "passed|failed|^E  |^FAILED" > gpurun_out/red.log
timeout 200 python bench.py --no-cpu-baseline --no-batch32 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('B1', d['value'], d['ms_per_step']); [print(g['kernel'], g['avg_us']) for g in d['roofline']['groups']]" >> gpurun_out/red.log 2>&1
cat gpurun_out/red.log
