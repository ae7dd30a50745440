mkdir -p gpurun_out
python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -3
python bench.py --steps 64 --warmup 8 > gpurun_out/bench1.json 2> gpurun_out/bench1.err; tail -3 gpurun_out/bench1.err; cat gpurun_out/bench1.json
