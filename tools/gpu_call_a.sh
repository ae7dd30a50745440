#!/bin/bash
# scratch: fullsize parity + bench(with parity) + attention sweep at batch 1
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/call_a
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s > $OUT/fullsize.log 2>&1
grep -o "{'[a-z_0-9]*': [^}]*}" $OUT/fullsize.log | cut -c1-420
grep -E "passed|failed" $OUT/fullsize.log | tail -1
TUNES="3=1;3=2;3=2,8=8;3=2,8=16;3=0,8=8;3=0,8=16;3=2,8=1" PF_BATCHES="1" NO_PROBE=1 timeout 600 python tools/exp_decode_sweep.py > $OUT/sweep_attn.log 2>&1
grep -E "tok/s|Error|error" $OUT/sweep_attn.log | tail -12
timeout 900 python bench.py --no-batch32 > $OUT/bench.json 2> $OUT/bench.err
python - <<'P'
import json
d=json.load(open("gpurun_out/call_a/bench.json"))
print(d["value"], d["ms_per_step"]); print(json.dumps(d.get("parity"))[:1500]); print(d["roofline"].get("traffic_source"))
P
