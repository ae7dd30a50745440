mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | grep -E "passed|failed|^E  |^FAILED" > gpurun_out/fix.log
timeout 200 python bench.py --steps 64 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B1', d['value'], d['ms_per_step'], 'B32', d['batch32']['value'])" >> gpurun_out/fix.log 2>&1
cat gpurun_out/fix.log
