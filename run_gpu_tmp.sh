mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_dense_model.py -m gpu -q 2>&1 | tail -15 > gpurun_out/stablelm.log
cat gpurun_out/stablelm.log
