// KV-cache index/copy kernels for gfx950: reshape_and_cache (K1), copy_blocks (K5), swap_blocks (K6).
// All three are pure byte movers -> results are bit-exact by construction.
//
// Reference boundary:
//   copy_blocks_{bf16,f16,f32}   src/backend/cache.rs:127-162 (attention_rs::kernels::ffi)
//   swap_blocks                  src/scheduler/cache_engine.rs:527-535 (attention_rs::cache::swap_blocks)
//   reshape_and_cache            inside PagedAttention::forward, src/openai/models/layers/attention.rs:983-995
//   cache layouts                src/scheduler/cache_engine.rs:298-341
#include "common.h"
#include "../../include/mi355_vllm.h"

// ------------------------------------------------------------------------------------------------
// reshape_and_cache.  One workgroup per token; k/v rows are [Hkv*D] elements of ES bytes.
// FLASH layout: cache[(slot*Hkv + h)*D + d]           -> the row is one contiguous 16-B-vector copy.
// PAGED layout: K[(((blk*Hkv+h)*(D/x) + d/x)*bs + off)*x + d%x]  (x = 16/ES, 16-B granules)
//               V[((blk*Hkv+h)*D + d)*bs + off]                  (element scatter)
template <int ES>
__global__ void __launch_bounds__(256) reshape_and_cache_flash_kernel(
    const uint8_t* __restrict__ k, const uint8_t* __restrict__ v, uint8_t* __restrict__ kc,
    uint8_t* __restrict__ vc, const int64_t* __restrict__ slot_mapping, int row_elems,
    int64_t k_tok_stride_b, int64_t v_tok_stride_b) {
    const int t = blockIdx.x;
    const int64_t slot = slot_mapping[t];
    if (slot < 0) return;                                 // _PAD_SLOT_ID: skip (llm_engine.rs:94)
    const int row_bytes = row_elems * ES;
    const uint8_t* ks = k + (int64_t)t * k_tok_stride_b;
    const uint8_t* vs = v + (int64_t)t * v_tok_stride_b;
    uint8_t* kd = kc + slot * (int64_t)row_bytes;
    uint8_t* vd = vc + slot * (int64_t)row_bytes;
    if ((row_bytes & 15) == 0 && (((uintptr_t)ks | (uintptr_t)vs | (uintptr_t)kd | (uintptr_t)vd) & 15) == 0) {
        const int nv = row_bytes >> 4;
        for (int i = threadIdx.x; i < nv; i += blockDim.x) {
            reinterpret_cast<uint4*>(kd)[i] = reinterpret_cast<const uint4*>(ks)[i];
            reinterpret_cast<uint4*>(vd)[i] = reinterpret_cast<const uint4*>(vs)[i];
        }
    } else {
        for (int i = threadIdx.x; i < row_bytes; i += blockDim.x) { kd[i] = ks[i]; vd[i] = vs[i]; }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) reshape_and_cache_paged_kernel(
    const T* __restrict__ k, const T* __restrict__ v, T* __restrict__ kc, T* __restrict__ vc,
    const int64_t* __restrict__ slot_mapping, int num_kv_heads, int head_dim, int block_size,
    int64_t k_tok_stride, int64_t v_tok_stride) {
    constexpr int X = 16 / (int)sizeof(T);
    const int t = blockIdx.x;
    const int64_t slot = slot_mapping[t];
    if (slot < 0) return;
    const int64_t blk = slot / block_size;
    const int off = (int)(slot % block_size);
    const int n = num_kv_heads * head_dim;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int h = i / head_dim, d = i % head_dim;
        const int64_t ki = ((((blk * num_kv_heads + h) * (head_dim / X) + d / X) * block_size + off) * X) + d % X;
        const int64_t vi = ((blk * num_kv_heads + h) * head_dim + d) * (int64_t)block_size + off;
        kc[ki] = k[(int64_t)t * k_tok_stride + i];
        vc[vi] = v[(int64_t)t * v_tok_stride + i];
    }
}

// fp8 KV cache (`--kvcache-dtype fp8`, src/main.rs:263-267; K layout x = 16, cache_engine.rs:304-311): bf16 k, v are
// stored as OCP e4m3fn bytes = e4m3(value / scale), round-to-nearest-even, saturating at +-448 (scale 1.0 is what the
// reference passes today [EXT: conversion lives in attention-rs]).
__global__ void __launch_bounds__(256) reshape_and_cache_fp8_kernel(const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
                                                                    uint8_t* __restrict__ kc, uint8_t* __restrict__ vc,
                                                                    const int64_t* __restrict__ slot_mapping, int Hkv, int D, int bs,
                                                                    int layout, float inv_k_scale, float inv_v_scale) {
    const int t = blockIdx.x;
    const int64_t slot = slot_mapping[t];
    if (slot < 0) return;
    const int64_t blk = slot / bs, off = slot % bs;
    const int n = Hkv * D;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int h = i / D, d = i % D;
        const float kf = bf16_to_f32(k[(int64_t)t * n + i]) * inv_k_scale, vf = bf16_to_f32(v[(int64_t)t * n + i]) * inv_v_scale;
        int64_t ko, vo;
        if (layout == MI355_KV_FLASH) { ko = vo = (slot * Hkv + h) * D + d; }
        else {
            ko = (((blk * Hkv + h) * (D / 16) + d / 16) * bs + off) * 16 + d % 16;
            vo = ((blk * Hkv + h) * D + d) * (int64_t)bs + off;
        }
        kc[ko] = to_e4m3(kf);
        vc[vo] = to_e4m3(vf);
    }
}
extern "C" int mi355_reshape_and_cache_fp8(const void* k, const void* v, void* key_cache, void* value_cache,
                                           const int64_t* slot_mapping, int32_t num_tokens, int32_t num_kv_heads,
                                           int32_t head_dim, int32_t block_size, int32_t layout, float k_scale, float v_scale,
                                           int64_t stream) {
    if (num_tokens <= 0) return 0;
    if (layout == MI355_KV_PAGED && (head_dim % 16)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(reshape_and_cache_fp8_kernel, dim3(num_tokens), dim3(256), 0, to_stream(stream), (const uint16_t*)k,
                       (const uint16_t*)v, (uint8_t*)key_cache, (uint8_t*)value_cache, slot_mapping, num_kv_heads, head_dim,
                       block_size, layout, 1.f / k_scale, 1.f / v_scale);
    return (int)hipGetLastError();
}

extern "C" int mi355_reshape_and_cache(const void* k, const void* v, void* key_cache, void* value_cache,
                                       const int64_t* slot_mapping, int32_t num_tokens,
                                       int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                                       int32_t elem_size, int32_t layout, int64_t stream) {
    if (num_tokens <= 0) return 0;
    if (elem_size != 1 && elem_size != 2 && elem_size != 4) return (int)hipErrorInvalidValue;
    hipStream_t st = to_stream(stream);
    const int row = num_kv_heads * head_dim;
    if (layout == MI355_KV_FLASH) {
        const int64_t ts = (int64_t)row * elem_size;
#define LAUNCH(ES)                                                                                     \
    hipLaunchKernelGGL(reshape_and_cache_flash_kernel<ES>, dim3(num_tokens), dim3(256), 0, st,         \
                       (const uint8_t*)k, (const uint8_t*)v, (uint8_t*)key_cache, (uint8_t*)value_cache, \
                       slot_mapping, row, ts, ts)
        if (elem_size == 1) LAUNCH(1); else if (elem_size == 2) LAUNCH(2); else LAUNCH(4);
#undef LAUNCH
    } else if (layout == MI355_KV_PAGED) {
        if (head_dim % (16 / elem_size)) return (int)hipErrorInvalidValue;
#define LAUNCH(T)                                                                                      \
    hipLaunchKernelGGL(reshape_and_cache_paged_kernel<T>, dim3(num_tokens), dim3(256), 0, st,          \
                       (const T*)k, (const T*)v, (T*)key_cache, (T*)value_cache, slot_mapping,         \
                       num_kv_heads, head_dim, block_size, (int64_t)row, (int64_t)row)
        if (elem_size == 1) LAUNCH(uint8_t); else if (elem_size == 2) LAUNCH(uint16_t); else LAUNCH(uint32_t);
#undef LAUNCH
    } else {
        return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// copy_blocks.  The reference ABI hands HOST arrays (pointer tables + pair list) that the caller
// frees right after the call (Rust Vecs, cache.rs:44-113), so they are passed BY VALUE in the kernel
// argument block (copied at launch time: asynchronous, capturable, no staging buffer, no sync).
// grid = (pairs_in_chunk, layers_in_chunk, 2 {K,V}); each workgroup streams one block with 16-B vectors.
#define CB_MAX_LAYERS 64
#define CB_MAX_PAIRS 96
struct CopyBlocksArgs {
    uint64_t kptr[CB_MAX_LAYERS];
    uint64_t vptr[CB_MAX_LAYERS];
    int64_t pairs[2 * CB_MAX_PAIRS];
};

__global__ void __launch_bounds__(256) copy_blocks_kernel(const CopyBlocksArgs a, int64_t bytes_per_block) {
    const int pair = blockIdx.x, layer = blockIdx.y;
    uint8_t* base = reinterpret_cast<uint8_t*>(blockIdx.z == 0 ? a.kptr[layer] : a.vptr[layer]);
    const int64_t src = a.pairs[2 * pair], dst = a.pairs[2 * pair + 1];
    const uint8_t* s = base + src * bytes_per_block;
    uint8_t* d = base + dst * bytes_per_block;
    if (((bytes_per_block | (int64_t)(uintptr_t)s | (int64_t)(uintptr_t)d) & 15) == 0) {
        const int64_t nv = bytes_per_block >> 4;
        for (int64_t i = threadIdx.x; i < nv; i += blockDim.x)
            reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
    } else {
        for (int64_t i = threadIdx.x; i < bytes_per_block; i += blockDim.x) d[i] = s[i];
    }
}

static void copy_blocks_impl(void* key_cache_ptrs, void* value_cache_ptrs, const void* block_mapping,
                             int32_t num_layers, int32_t num_pairs, int32_t numel_per_block, int elem_size,
                             int64_t stream) {
    if (num_layers <= 0 || num_pairs <= 0) return;
    const uint64_t* kp = static_cast<const uint64_t*>(key_cache_ptrs);
    const uint64_t* vp = static_cast<const uint64_t*>(value_cache_ptrs);
    const int64_t* bm = static_cast<const int64_t*>(block_mapping);
    const int64_t bytes = (int64_t)numel_per_block * elem_size;
    // NOTE: pairs of one call may chain (a->b, b->c); chunks are launched in order on one stream and the
    // reference kernel gives no ordering guarantee inside a call either (grid of independent blocks).
    for (int l0 = 0; l0 < num_layers; l0 += CB_MAX_LAYERS) {
        const int nl = (num_layers - l0 < CB_MAX_LAYERS) ? num_layers - l0 : CB_MAX_LAYERS;
        for (int p0 = 0; p0 < num_pairs; p0 += CB_MAX_PAIRS) {
            const int np = (num_pairs - p0 < CB_MAX_PAIRS) ? num_pairs - p0 : CB_MAX_PAIRS;
            CopyBlocksArgs a;
            for (int i = 0; i < nl; ++i) { a.kptr[i] = kp[l0 + i]; a.vptr[i] = vp[l0 + i]; }
            for (int i = 0; i < 2 * np; ++i) a.pairs[i] = bm[2 * p0 + i];
            hipLaunchKernelGGL(copy_blocks_kernel, dim3(np, nl, 2), dim3(256), 0, to_stream(stream), a, bytes);
        }
    }
}

extern "C" void copy_blocks_bf16(void* k, void* v, const void* bm, int32_t nl, int32_t np, int32_t numel, int64_t s) {
    copy_blocks_impl(k, v, bm, nl, np, numel, 2, s);
}
extern "C" void copy_blocks_f16(void* k, void* v, const void* bm, int32_t nl, int32_t np, int32_t numel, int64_t s) {
    copy_blocks_impl(k, v, bm, nl, np, numel, 2, s);
}
extern "C" void copy_blocks_f32(void* k, void* v, const void* bm, int32_t nl, int32_t np, int32_t numel, int64_t s) {
    copy_blocks_impl(k, v, bm, nl, np, numel, 4, s);
}
extern "C" void copy_blocks_u8(void* k, void* v, const void* bm, int32_t nl, int32_t np, int32_t numel, int64_t s) {
    copy_blocks_impl(k, v, bm, nl, np, numel, 1, s);   // fp8 KV cache is stored as u8 (main.rs:263-267)
}

// ------------------------------------------------------------------------------------------------
// swap_blocks: block-granular copies between two tensors whose dim 0 is the block index; either side
// may be host (pinned or pageable) or device memory.  Runs of consecutive (src+1 -> dst+1) pairs are
// merged into one hipMemcpyAsync (the reference issues one memcpy per block).
extern "C" int mi355_swap_blocks(const void* src, void* dst, const int64_t* mapping_pairs, int32_t num_pairs,
                                 int64_t bytes_per_block, int32_t kind, int64_t stream) {
    if (num_pairs <= 0 || bytes_per_block <= 0) return 0;
    hipMemcpyKind mk;
    switch (kind) {
        case MI355_SWAP_H2D: mk = hipMemcpyHostToDevice; break;
        case MI355_SWAP_D2H: mk = hipMemcpyDeviceToHost; break;
        case MI355_SWAP_D2D: mk = hipMemcpyDeviceToDevice; break;
        default: return (int)hipErrorInvalidValue;
    }
    const uint8_t* s = static_cast<const uint8_t*>(src);
    uint8_t* d = static_cast<uint8_t*>(dst);
    int i = 0;
    while (i < num_pairs) {
        int j = i + 1;
        while (j < num_pairs && mapping_pairs[2 * j] == mapping_pairs[2 * (j - 1)] + 1 &&
               mapping_pairs[2 * j + 1] == mapping_pairs[2 * (j - 1) + 1] + 1)
            ++j;
        hipError_t e = hipMemcpyAsync(d + mapping_pairs[2 * i + 1] * bytes_per_block,
                                      s + mapping_pairs[2 * i] * bytes_per_block,
                                      (size_t)(j - i) * bytes_per_block, mk, to_stream(stream));
        if (e != hipSuccess) return (int)e;
        i = j;
    }
    return 0;
}
