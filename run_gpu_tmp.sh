mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_model.py -m gpu -q 2>&1 | grep -E "passed|failed|^E  |^FAILED|Error" | head -20 > gpurun_out/fp8g.log
cat gpurun_out/fp8g.log
