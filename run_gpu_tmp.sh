mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_dense_model.py -m gpu -q 2>&1 | tail -12 > gpurun_out/fp8m.log
cat gpurun_out/fp8m.log
