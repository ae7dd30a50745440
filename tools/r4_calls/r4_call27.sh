#!/bin/bash
# final state of the prompt attention (query-block size chosen at launch, heaviest block first in both prompt-attention kernels): full GPU suite + prompt legs
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4c27
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1
grep "passed\|failed" $OUT/pytest_gpu.log | tail -1
PF_T=512,1024,2048,4096 PF_MODES=1 timeout 200 python tests/bench_prefill.py 2>&1 | grep "prefill" | tee $OUT/pf.log
for leg in bf16_prompt mixtral_prompt_16k; do
  timeout 300 python bench_legs.py $leg --no-parity 2>/dev/null | grep '^{' | tail -1 | cut -c1-400 | tee -a $OUT/legs.log
done
