// Device-side restatement of `prepare_decode` for the greedy loop of both host layers (host_model.cpp, dense_model.cpp):
// positions / slots / context lengths of the NEXT step from the block table, so that a captured graph runs step after step
// without host input.   reference: src/openai/pipelines/inputs.rs:389-423 (`prepare_decode`), backend/graph.rs:592-608 (static inputs)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

static __global__ void advance_kernel(uint32_t* tokens, const uint32_t* next_tokens, int64_t* positions, int64_t* slots,
                                      uint32_t* ctx, const uint32_t* bt, int max_blocks, int block_size, int batch) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    tokens[b] = next_tokens[b];
    const uint32_t n = ctx[b] + 1;              // sequence length after appending the sampled token
    ctx[b] = n;
    const int64_t pos = (int64_t)n - 1;
    positions[b] = pos;
    // the step after the last reserved block has no slot (-1 = "do not write", as a padded slot): the host refuses to
    // run that step, and this kernel never reads past the block-table row
    if (pos / block_size >= max_blocks) { slots[b] = -1; return; }
    const int64_t blk = bt[(size_t)b * max_blocks + pos / block_size];
    slots[b] = blk * block_size + pos % block_size;
}
