mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -k "gguf" 2>&1 | tail -15 > gpurun_out/gguf.log
cat gpurun_out/gguf.log
