#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for i in 1 2; do
B32_STEPS=16 timeout 100 python tools/exp_b32.py 2>&1 | grep "value" | cut -c1-100
MI355_TMP_Q4_LONGK=1 B32_STEPS=16 timeout 100 python tools/exp_b32.py 2>&1 | grep "value" | cut -c1-100
done
