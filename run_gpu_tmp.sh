mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_linear.py tests/test_gpu_prefill.py -m gpu -q > gpurun_out/pf_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/pf_tests.log
tail -25 gpurun_out/pf_tests.log
