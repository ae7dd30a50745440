mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_model.py -m gpu -q 2>&1 | tail -15 > gpurun_out/moem.log
cat gpurun_out/moem.log
