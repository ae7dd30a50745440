mkdir -p gpurun_out/c13
for t in "11=1" "11=2" "11=3" "10=512" "10=2048" "10=-256" "17=4" "14=0"; do
  MI355_TUNE=$t timeout 300 python bench.py --parity off --legs none --no-cpu-baseline --steps 16 --warmup 4 > gpurun_out/c13/b_$t.json 2>> gpurun_out/c13/err.log
  python -c "
import json,sys; d=json.load(open('gpurun_out/c13/b_$t.json')); print('$t', 'b32', d['batch32']['value'], 'prefill', d['prefill']['value'])"
done
