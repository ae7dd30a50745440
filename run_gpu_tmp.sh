mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_dense_model.py -m gpu -q 2>&1 | tail -2 > gpurun_out/dense_bench.log
timeout 400 python tests/bench_dense_model.py >> gpurun_out/dense_bench.log 2>&1
cat gpurun_out/dense_bench.log
