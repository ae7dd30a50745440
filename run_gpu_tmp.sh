mkdir -p gpurun_out
for t in "4=1" "4=2" "4=1" "4=2"; do echo "TUNE $t"; MI355_TUNE=$t timeout 200 python bench.py --batch 32 --steps 16 --warmup 4 --no-cpu-baseline --no-batch32 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B32', d['value'], d['ms_per_step'])"; done > gpurun_out/ab.log 2>&1
cat gpurun_out/ab.log
