mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x 2>&1 | grep -E "passed|failed|^E  |^FAILED" > gpurun_out/wide5.log
for t in "4=1" "4=2"; do echo "TUNE $t"; MI355_TUNE=$t timeout 200 python bench.py --batch 32 --steps 16 --warmup 4 --no-cpu-baseline --no-batch32 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B32', d['value'], d['ms_per_step'])"; done >> gpurun_out/wide5.log 2>&1
timeout 100 python tests/bench_kernels.py --batch 32 2>&1 | grep -E "^wq|^down q4k|gate/up|lm_head|qkv" >> gpurun_out/wide5.log 2>&1
cat gpurun_out/wide5.log
