mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/moe.log
cat gpurun_out/moe.log
