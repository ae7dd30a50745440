// Share boundaries of the balanced LDS-DMA attention stream (paged_attention.hip paged_attn_stream_kernel): workgroup w of the W per kv
// head takes the flat (sequence, 64-token stage) indices [S w / W, S (w + 1) / W).  Shared with the merge kernels, which recompute the cuts
// (paged_attn_stream_reduce_kernel; qmatmul.hip's merge that also stages the next mat-mul's activation image) and with the host-side
// layout test.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
__host__ __device__ inline int64_t pas_cut(int64_t S, int W, int w) { return S * w / W; }
