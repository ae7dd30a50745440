#!/bin/bash
# usage: tools/kernel_resources.sh candle_vllm_amd/csrc/qmatmul.hip [filter]
cd /tmp && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fopenmp -Wno-unused-value -Rpass-analysis=kernel-resource-usage -c "$OLDPWD/$1" -o /tmp/_kr.o 2>&1 \
 | grep -E "Function Name|VGPRs:|Occupancy|VGPRs Spill|LDS Size" | sed 's/.*remark: [^ ]* *//; s/ *\[-Rpass.*//' | paste - - - - - | grep "${2:-.}"
