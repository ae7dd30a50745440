mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -k greedy 2>&1 | grep -E "passed|failed" > gpurun_out/full.log
timeout 800 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^E  .*Assert|^FAILED" >> gpurun_out/full.log
cat gpurun_out/full.log
