// Small memory-bound operators of the decode step for gfx950: RMSNorm (K8), RoPE (K7), SiLU*mul (K9),
// residual add, embedding gather (K16), argmax (K18), dtype casts.  All vectorised to 16 B per lane
// where the layout allows; fp32 math throughout.
//
// Reference semantics:
//   rms_norm   candle_nn::ops::rms_norm       src/openai/models/layers/qrmsnorm.rs:28-31
//   rope       FusedRope::apply_inplace[_partial] / candle rope, rope_i   layers/rotary_emb.rs:52-101
//   silu_mul   candle_nn::ops::silu(w1) * w3   src/openai/models/quantized_llama.rs:33-37
//   argmax     logits.argmax(-1) (first max)   src/openai/logits_processor.rs:92-95
#include "common.h"
#include "scratch.h"
#include "../../include/mi355_vllm.h"

// ------------------------------------------------------------------------------------------------ RMSNorm
template <typename T> __device__ __forceinline__ float ld_as_f32(const T* p, int64_t i);
template <> __device__ __forceinline__ float ld_as_f32<float>(const float* p, int64_t i) { return p[i]; }
template <> __device__ __forceinline__ float ld_as_f32<uint16_t>(const uint16_t* p, int64_t i) { return bf16_to_f32(p[i]); }
template <typename T> __device__ __forceinline__ void st_from_f32(T* p, int64_t i, float v);
template <> __device__ __forceinline__ void st_from_f32<float>(float* p, int64_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void st_from_f32<uint16_t>(uint16_t* p, int64_t i, float v) { p[i] = f32_to_bf16(v); }

// one workgroup per row; x is re-read from L1/L2 in the second pass (row <= 64 KB).
// WT = weight element type (GGUF path: f32 weights with f32 x; safetensors path: bf16/bf16).
template <typename T, typename WT>
__global__ void __launch_bounds__(256) rms_norm_kernel(T* __restrict__ out, const T* __restrict__ x,
                                                       const WT* __restrict__ w, int hidden, float eps) {
    __shared__ float red[16];
    const int64_t row = blockIdx.x;
    const T* xr = x + row * hidden;
    T* orow = out + row * hidden;
    float ss = 0.f;
    if constexpr (sizeof(T) == 4) {
        if ((hidden & 3) == 0) {
            const float4* x4 = reinterpret_cast<const float4*>(xr);
            for (int i = threadIdx.x; i < hidden / 4; i += blockDim.x) {
                float4 v = x4[i];
                ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            }
        } else {
            for (int i = threadIdx.x; i < hidden; i += blockDim.x) { float v = xr[i]; ss += v * v; }
        }
    } else {
        for (int i = threadIdx.x; i < hidden; i += blockDim.x) { float v = ld_as_f32<T>(xr, i); ss += v * v; }
    }
    ss = block_sum(ss, red);
    const float inv = rsqrtf(ss / (float)hidden + eps);
    for (int i = threadIdx.x; i < hidden; i += blockDim.x)
        st_from_f32<T>(orow, i, ld_as_f32<T>(xr, i) * inv * ld_as_f32<WT>(w, i));
}

// bf16 rows up to 8192 wide (the decode step of the 16-bit host layers: one row = one workgroup = a chain of memory round
// trips): x AND the weight are requested up front as 16-byte pieces and stay in registers, so the launch is one round trip
// instead of three (measured on the Qwen2-7B GPTQ leg: 11.3 us per call with the generic kernel, two calls per layer).
__global__ void __launch_bounds__(256) rms_norm_bf16_kernel(uint16_t* __restrict__ out, const uint16_t* __restrict__ x,
                                                            const uint16_t* __restrict__ w, int hidden, float eps) {
    __shared__ float red[16];
    const int64_t row = blockIdx.x;
    const uint4* x4 = reinterpret_cast<const uint4*>(x + row * hidden);
    const uint4* w4 = reinterpret_cast<const uint4*>(w);
    uint4* o4 = reinterpret_cast<uint4*>(out + row * hidden);
    const int n8 = hidden >> 3;
    uint4 xv[4], wv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int i = threadIdx.x + c * 256;
        xv[c] = i < n8 ? x4[i] : make_uint4(0, 0, 0, 0);
        wv[c] = i < n8 ? w4[i] : make_uint4(0, 0, 0, 0);
    }
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t u[4] = {xv[c].x, xv[c].y, xv[c].z, xv[c].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = bf16lo_to_f32(u[e]), b = bf16hi_to_f32(u[e]);
            ss += a * a + b * b;
        }
    }
    ss = block_sum(ss, red);
    const float inv = rsqrtf(ss / (float)hidden + eps);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int i = threadIdx.x + c * 256;
        if (i >= n8) continue;
        const uint32_t u[4] = {xv[c].x, xv[c].y, xv[c].z, xv[c].w}, g[4] = {wv[c].x, wv[c].y, wv[c].z, wv[c].w};
        uint32_t r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            r[e] = pack_bf16x2(bf16lo_to_f32(u[e]) * inv * bf16lo_to_f32(g[e]), bf16hi_to_f32(u[e]) * inv * bf16hi_to_f32(g[e]));
        o4[i] = make_uint4(r[0], r[1], r[2], r[3]);
    }
}

extern "C" int mi355_rms_norm(void* out, const void* x, const void* w, int32_t num_tokens, int32_t hidden,
                              float eps, int32_t dtype, int32_t w_dtype, int64_t stream) {
    if (num_tokens <= 0) return 0;
    hipStream_t st = to_stream(stream);
    if (dtype == MI355_DTYPE_BF16 && w_dtype == MI355_DTYPE_BF16 && (hidden & 7) == 0 && hidden <= 8192 &&
        ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) == 0) {
        hipLaunchKernelGGL(rms_norm_bf16_kernel, dim3(num_tokens), dim3(256), 0, st, (uint16_t*)out, (const uint16_t*)x,
                           (const uint16_t*)w, hidden, eps);
        return (int)hipGetLastError();
    }
    if (dtype == MI355_DTYPE_F32 && w_dtype == MI355_DTYPE_F32)
        hipLaunchKernelGGL((rms_norm_kernel<float, float>), dim3(num_tokens), dim3(256), 0, st, (float*)out,
                           (const float*)x, (const float*)w, hidden, eps);
    else if (dtype == MI355_DTYPE_BF16 && w_dtype == MI355_DTYPE_BF16)
        hipLaunchKernelGGL((rms_norm_kernel<uint16_t, uint16_t>), dim3(num_tokens), dim3(256), 0, st,
                           (uint16_t*)out, (const uint16_t*)x, (const uint16_t*)w, hidden, eps);
    else if (dtype == MI355_DTYPE_BF16 && w_dtype == MI355_DTYPE_F32)
        hipLaunchKernelGGL((rms_norm_kernel<uint16_t, float>), dim3(num_tokens), dim3(256), 0, st,
                           (uint16_t*)out, (const uint16_t*)x, (const float*)w, hidden, eps);
    else
        return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ LayerNorm
// candle_nn::LayerNorm (StableLM: src/openai/models/stable_lm.rs:61-72) on 16-bit data: statistics and the affine
// in f32, one rounding at the end (the fused candle_nn::ops::layer_norm path for contiguous inputs [EXT]).
template <typename T>
__global__ void __launch_bounds__(256) layer_norm_kernel(T* __restrict__ out, const T* __restrict__ x, const T* __restrict__ w,
                                                         const T* __restrict__ b, int hidden, float eps) {
    __shared__ float red[16];
    const int64_t row = blockIdx.x;
    const T* xr = x + row * hidden;
    T* orow = out + row * hidden;
    float s = 0.f;
    for (int i = threadIdx.x; i < hidden; i += blockDim.x) s += ld_as_f32<T>(xr, i);
    const float mean = block_sum(s, red) / (float)hidden;
    float ss = 0.f;
    for (int i = threadIdx.x; i < hidden; i += blockDim.x) { const float d = ld_as_f32<T>(xr, i) - mean; ss += d * d; }
    const float inv = rsqrtf(block_sum(ss, red) / (float)hidden + eps);
    for (int i = threadIdx.x; i < hidden; i += blockDim.x) {
        float v = (ld_as_f32<T>(xr, i) - mean) * inv * ld_as_f32<T>(w, i);
        if (b) v += ld_as_f32<T>(b, i);
        st_from_f32<T>(orow, i, v);
    }
}
extern "C" int mi355_layer_norm(void* out, const void* x, const void* w, const void* b, int32_t num_tokens, int32_t hidden,
                                float eps, int32_t dtype, int64_t stream) {
    if (num_tokens <= 0) return 0;
    hipStream_t st = to_stream(stream);
    if (dtype == MI355_DTYPE_BF16)
        hipLaunchKernelGGL((layer_norm_kernel<uint16_t>), dim3(num_tokens), dim3(256), 0, st, (uint16_t*)out, (const uint16_t*)x,
                           (const uint16_t*)w, (const uint16_t*)b, hidden, eps);
    else if (dtype == MI355_DTYPE_F32)
        hipLaunchKernelGGL((layer_norm_kernel<float>), dim3(num_tokens), dim3(256), 0, st, (float*)out, (const float*)x,
                           (const float*)w, (const float*)b, hidden, eps);
    else
        return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ RoPE (in place)
// q [T, H, D], k [T, Hkv, D]; cos/sin f32 [max_seq, rot/2]; positions i64 [T].
// is_rope_i != 0: interleaved pairs (x[2i], x[2i+1])  (GGUF llama);  else half-split (x[i], x[i+rot/2]).
template <typename T>
__global__ void __launch_bounds__(256) rope_kernel(T* __restrict__ q, T* __restrict__ k,
                                                   const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                   const int64_t* __restrict__ positions, int H, int Hkv, int D,
                                                   int rot, int is_rope_i) {
    const int t = blockIdx.x;
    const int64_t pos = positions[t];
    const int half = rot >> 1;
    const float* c = cosT + pos * half;
    const float* s = sinT + pos * half;
    const int npairs = (H + Hkv) * half;
    for (int i = threadIdx.x; i < npairs; i += blockDim.x) {
        const int h = i / half, j = i % half;
        T* base = (h < H) ? q + ((int64_t)t * H + h) * D : k + ((int64_t)t * Hkv + (h - H)) * D;
        const int i0 = is_rope_i ? 2 * j : j;
        const int i1 = is_rope_i ? 2 * j + 1 : j + half;
        const float x0 = ld_as_f32<T>(base, i0), x1 = ld_as_f32<T>(base, i1);
        const float cc = c[j], sn = s[j];
        float f0, f1;
        rope_rotate(x0, x1, cc, sn, f0, f1);
        st_from_f32<T>(base, i0, f0);
        st_from_f32<T>(base, i1, f1);
    }
}

extern "C" int mi355_rope_inplace(void* q, void* k, const float* cos_t, const float* sin_t,
                                  const int64_t* positions, int32_t num_tokens, int32_t num_heads,
                                  int32_t num_kv_heads, int32_t head_dim, int32_t rotary_dim,
                                  int32_t is_rope_i, int32_t dtype, int64_t stream) {
    if (num_tokens <= 0) return 0;
    if (rotary_dim <= 0 || rotary_dim > head_dim || (rotary_dim & 1)) return (int)hipErrorInvalidValue;
    hipStream_t st = to_stream(stream);
    if (dtype == MI355_DTYPE_F32)
        hipLaunchKernelGGL(rope_kernel<float>, dim3(num_tokens), dim3(256), 0, st, (float*)q, (float*)k, cos_t,
                           sin_t, positions, num_heads, num_kv_heads, head_dim, rotary_dim, is_rope_i);
    else if (dtype == MI355_DTYPE_BF16)
        hipLaunchKernelGGL(rope_kernel<uint16_t>, dim3(num_tokens), dim3(256), 0, st, (uint16_t*)q,
                           (uint16_t*)k, cos_t, sin_t, positions, num_heads, num_kv_heads, head_dim, rotary_dim,
                           is_rope_i);
    else
        return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}

// RoPE of q and k in place AND the cache write of the rotated k and of v, one launch (16-bit data, bf16 cache, both cache
// layouts): in a decode step the two separate launches are 5 us each for a few KB.  Same arithmetic and rounding points as
// rope_kernel followed by reshape_and_cache (q, k stay rotated in place; padded slots skip the cache).
__global__ void __launch_bounds__(256) rope_cache_bf16_kernel(uint16_t* __restrict__ q, uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
                                                              uint16_t* __restrict__ kc, uint16_t* __restrict__ vc,
                                                              const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                              const int64_t* __restrict__ positions, const int64_t* __restrict__ slot_mapping,
                                                              int H, int Hkv, int D, int rot, int is_rope_i, int bs, int flash) {
    const int t = blockIdx.x;
    const int64_t pos = positions[t], slot = slot_mapping[t];
    const int half = rot >> 1;
    const float* c = cosT + pos * half;
    const float* s = sinT + pos * half;
    const int64_t blk = slot >= 0 ? slot / bs : 0;
    const int off = slot >= 0 ? (int)(slot % bs) : 0;
    auto kidx = [&](int h, int d) -> int64_t {
        return flash ? (slot * Hkv + h) * D + d : ((((blk * Hkv + h) * (D / 8) + d / 8) * bs + off) * 8) + d % 8;
    };
    auto vidx = [&](int h, int d) -> int64_t {
        return flash ? (slot * Hkv + h) * D + d : ((blk * Hkv + h) * D + d) * (int64_t)bs + off;
    };
    const int npairs = (H + Hkv) * half;
    for (int i = threadIdx.x; i < npairs; i += blockDim.x) {
        const int h = i / half, j = i % half;
        const bool isq = h < H;
        uint16_t* base = isq ? q + ((int64_t)t * H + h) * D : k + ((int64_t)t * Hkv + (h - H)) * D;
        const int i0 = is_rope_i ? 2 * j : j;
        const int i1 = is_rope_i ? 2 * j + 1 : j + half;
        const float x0 = bf16_to_f32(base[i0]), x1 = bf16_to_f32(base[i1]);
        const float cc = c[j], sn = s[j];
        float f0, f1;
        rope_rotate(x0, x1, cc, sn, f0, f1);
        const uint16_t r0 = f32_to_bf16(f0), r1 = f32_to_bf16(f1);
        base[i0] = r0; base[i1] = r1;
        if (!isq && slot >= 0) { kc[kidx(h - H, i0)] = r0; kc[kidx(h - H, i1)] = r1; }
    }
    if (slot < 0) return;
    // the dimensions RoPE leaves alone (partial rotary), and v
    const int nrest = Hkv * (D - rot);
    for (int i = threadIdx.x; i < nrest; i += blockDim.x) {
        const int h = i / (D - rot), d = rot + i % (D - rot);
        kc[kidx(h, d)] = k[((int64_t)t * Hkv + h) * D + d];
    }
    const int nv = Hkv * D;
    for (int i = threadIdx.x; i < nv; i += blockDim.x) {
        const int h = i / D, d = i % D;
        vc[vidx(h, d)] = v[(int64_t)t * nv + i];
    }
}
/* internal to the 16-bit host layer (dense_model.cpp); -4 = shapes this kernel does not take (caller uses the two launches) */
extern "C" int mi355_internal_rope_cache(void* q, void* k, const void* v, void* key_cache, void* value_cache, const float* cos_t,
                                         const float* sin_t, const int64_t* positions, const int64_t* slot_mapping,
                                         int32_t num_tokens, int32_t num_heads, int32_t num_kv_heads, int32_t head_dim,
                                         int32_t rotary_dim, int32_t is_rope_i, int32_t block_size, int32_t layout, int32_t dtype,
                                         int64_t stream) {
    if (num_tokens <= 0) return 0;
    if (dtype != MI355_DTYPE_BF16 || rotary_dim <= 0 || rotary_dim > head_dim || (rotary_dim & 1) || (head_dim & 7) ||
        (layout != MI355_KV_FLASH && layout != MI355_KV_PAGED))
        return -4;
    hipLaunchKernelGGL(rope_cache_bf16_kernel, dim3(num_tokens), dim3(256), 0, to_stream(stream), (uint16_t*)q, (uint16_t*)k,
                       (const uint16_t*)v, (uint16_t*)key_cache, (uint16_t*)value_cache, cos_t, sin_t, positions, slot_mapping,
                       num_heads, num_kv_heads, head_dim, rotary_dim, is_rope_i, block_size, layout == MI355_KV_FLASH ? 1 : 0);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ SiLU * mul, add, casts
template <typename T>
__global__ void __launch_bounds__(256) silu_mul_kernel(T* __restrict__ out, const T* __restrict__ g,
                                                       const T* __restrict__ u, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float a = ld_as_f32<T>(g, i), b = ld_as_f32<T>(u, i);
        st_from_f32<T>(out, i, a / (1.f + __expf(-a)) * b);
    }
}
extern "C" int mi355_silu_mul(void* out, const void* gate, const void* up, int64_t n, int32_t dtype, int64_t stream) {
    if (n <= 0) return 0;
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    if (dtype == MI355_DTYPE_F32)
        hipLaunchKernelGGL(silu_mul_kernel<float>, dim3(grid), dim3(256), 0, to_stream(stream), (float*)out,
                           (const float*)gate, (const float*)up, n);
    else if (dtype == MI355_DTYPE_BF16)
        hipLaunchKernelGGL(silu_mul_kernel<uint16_t>, dim3(grid), dim3(256), 0, to_stream(stream), (uint16_t*)out,
                           (const uint16_t*)gate, (const uint16_t*)up, n);
    else
        return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}

__global__ void __launch_bounds__(256) add_f32_kernel(float* __restrict__ out, const float* __restrict__ a,
                                                      const float* __restrict__ b, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = a[i] + b[i];
}
extern "C" int mi355_add_f32(float* out, const float* a, const float* b, int64_t n, int64_t stream) {
    if (n <= 0) return 0;
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(add_f32_kernel, dim3(grid), dim3(256), 0, to_stream(stream), out, a, b, n);
    return (int)hipGetLastError();
}

__global__ void __launch_bounds__(256) cast_f32_bf16_kernel(uint16_t* __restrict__ out, const float* __restrict__ in, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = f32_to_bf16(in[i]);
}
__global__ void __launch_bounds__(256) cast_bf16_f32_kernel(float* __restrict__ out, const uint16_t* __restrict__ in, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = bf16_to_f32(in[i]);
}
extern "C" int mi355_cast(void* out, const void* in, int64_t n, int32_t src_dtype, int32_t dst_dtype, int64_t stream) {
    if (n <= 0) return 0;
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    if (src_dtype == MI355_DTYPE_F32 && dst_dtype == MI355_DTYPE_BF16)
        hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid), dim3(256), 0, to_stream(stream), (uint16_t*)out, (const float*)in, n);
    else if (src_dtype == MI355_DTYPE_BF16 && dst_dtype == MI355_DTYPE_F32)
        hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid), dim3(256), 0, to_stream(stream), (float*)out, (const uint16_t*)in, n);
    else
        return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ embedding gather
// table f32 [V, hidden] (GGUF: dequantised once at load, quantized_llama.rs:262-264) -> out f32 [T, hidden]
__global__ void __launch_bounds__(256) embedding_f32_kernel(float* __restrict__ out, const float* __restrict__ table,
                                                            const uint32_t* __restrict__ ids, int hidden) {
    const int64_t t = blockIdx.x;
    const float* src = table + (int64_t)ids[t] * hidden;
    float* dst = out + t * hidden;
    if ((hidden & 3) == 0) {
        for (int i = threadIdx.x; i < hidden / 4; i += blockDim.x)
            reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
    } else {
        for (int i = threadIdx.x; i < hidden; i += blockDim.x) dst[i] = src[i];
    }
}
extern "C" int mi355_embedding_f32(float* out, const float* table, const uint32_t* ids, int32_t num_tokens,
                                   int32_t hidden, int64_t stream) {
    if (num_tokens <= 0) return 0;
    hipLaunchKernelGGL(embedding_f32_kernel, dim3(num_tokens), dim3(256), 0, to_stream(stream), out, table, ids, hidden);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ argmax (greedy)
// logits f32 [B, V] -> u32 [B]; ties -> lowest index (candle argmax keeps the first maximum [EXT]).
// NaNs never win (they map to the lowest key), matching a sequential `if v > best` scan.
// Stage 1: grid (ARGMAX_SPLIT, B) workgroups reduce a slice each and atomicMax a packed 64-bit key
//          (order-preserving value bits << 32 | ~index) into a per-row slot; stage 2 decodes and re-zeroes the
//          slot, so the scratch is self-cleaning and the pair is safe to replay from a hipGraph.
#define ARGMAX_SPLIT 64
#define ARGMAX_MAX_ROWS 4096

__device__ __forceinline__ unsigned long long argmax_key(float v, uint32_t idx) {
    uint32_t u = __float_as_uint(v);
    uint32_t k = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    if (v != v) k = 0;
    return ((unsigned long long)k << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
}

__global__ void __launch_bounds__(256) argmax_stage1_kernel(unsigned long long* __restrict__ slots,
                                                            const float* __restrict__ logits, int V) {
    __shared__ unsigned long long sk[4];
    const float* row = logits + (int64_t)blockIdx.y * V;
    const int per = (V + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = min(V, lo + per);
    unsigned long long best = 0;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const unsigned long long k = argmax_key(row[i], (uint32_t)i);
        best = k > best ? k : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long ok = __shfl_xor(best, o, 64);
        best = ok > best ? ok : best;
    }
    if ((threadIdx.x & 63) == 0) sk[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) best = sk[w] > best ? sk[w] : best;
        atomicMax(slots + blockIdx.y, best);
    }
}
__global__ void argmax_stage2_kernel(uint32_t* __restrict__ out, unsigned long long* __restrict__ slots, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const unsigned long long k = slots[b];
    slots[b] = 0;
    out[b] = (k >> 32) == 0 ? 0u : (0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFu));
}
extern "C" int mi355_argmax_f32(uint32_t* out, const float* logits, int32_t batch, int32_t vocab, int64_t stream) {
    if (batch <= 0) return 0;
    if (batch > ARGMAX_MAX_ROWS || vocab <= 0) return (int)hipErrorInvalidValue;
    hipStream_t st = to_stream(stream);
    void* sl = nullptr;                                     // slots belong to (device, stream): scratch.cpp
    const int src = mi355_scratch_get(&sl, MI355_SCR_ARGMAX, ARGMAX_MAX_ROWS * 8, st, true);
    if (src) return src;
    unsigned long long* g_argmax_slots = static_cast<unsigned long long*>(sl);
    hipLaunchKernelGGL(argmax_stage1_kernel, dim3(ARGMAX_SPLIT, batch), dim3(256), 0, st, g_argmax_slots, logits, vocab);
    hipLaunchKernelGGL(argmax_stage2_kernel, dim3((batch + 255) / 256), dim3(256), 0, st, out, g_argmax_slots, batch);
    return (int)hipGetLastError();
}
