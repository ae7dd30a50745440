mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | grep -E "passed|failed|^E  |^FAILED|Error" | head -20 > gpurun_out/pg.log
timeout 300 python tests/bench_prefill.py >> gpurun_out/pg.log 2>&1
cat gpurun_out/pg.log
