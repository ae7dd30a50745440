mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_linear.py tests/test_gpu_dense_model.py -m gpu -q 2>&1 | grep -E "passed|failed|^E  |^FAILED|Error" | head -20 > gpurun_out/dg.log
cat gpurun_out/dg.log
