mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_linear.py -m gpu -q > gpurun_out/lin_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/lin_tests.log
tail -15 gpurun_out/lin_tests.log
