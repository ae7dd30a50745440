// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.  wave = 64 lanes, always.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MI355_WAVE 64

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

#include "../../include/mi355_vllm.h"   // dtype / layout / epilogue codes of the C ABI

static inline hipStream_t to_stream(int64_t s) { return reinterpret_cast<hipStream_t>(s); }

// ---- bf16 <-> f32 (bit-exact round-to-nearest-even, NaN -> quiet NaN) -------------------------
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ float bf16lo_to_f32(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi_to_f32(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return 0x7FC0;
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}
// two f32 -> packed bf16x2 (lo in bits 0-15), round-to-nearest-even in hardware (gfx950 v_cvt_pk_bf16_f32).
// Written as a vector conversion the compiler selects the instruction for, NOT as inline asm: hipcc's hazard pass does not see
// through an asm string, so (a) an MFMA reading the asm-written VGPR right after got no wait states (wrong sums on the first
// m-tile of a wide launch, round 3) and (b) when the register allocator put the asm's result in a VGPR that an MFMA still in
// flight writes as a dead part of its 4-register result (a T=1 launch keeps only rows 0-1 of the tile), the MFMA's late write
// landed on top of the packed value: the NaN defect of the `<1,*,Q4_K,f32,no norm>` build, rounds 5-6 (profiles/r06_nan_hunt*).
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// RoPE rotation of one pair in f32, every product and sum rounded on its own (candle's rope on f32 values does not fuse a multiply
// into the add; with fused multiply-adds the compiler picks a different association in every kernel it inlines this into, and the
// same step run through two launch shapes stops being bit-identical)
__device__ __forceinline__ void rope_rotate(const float x0, const float x1, const float c, const float s, float& r0, float& r1) {
#pragma clang fp contract(off)
    const float a = x0 * c, b = x1 * s, d = x0 * s, e = x1 * c;
    r0 = a - b;
    r1 = d + e;
}
__device__ __forceinline__ float f16_bits_to_f32(uint16_t h) {
    _Float16 v = __builtin_bit_cast(_Float16, h);
    return (float)v;
}
__device__ __forceinline__ uint16_t f32_to_f16_bits(float f) {
    _Float16 v = (_Float16)f;
    return __builtin_bit_cast(uint16_t, v);
}

// f32 -> OCP e4m3fn byte: round-to-nearest-even in hardware (gfx950 v_cvt_pk_fp8_f32), saturating at +-448
__device__ __forceinline__ uint8_t to_e4m3(float f) {
    f = fminf(fmaxf(f, -448.f), 448.f);
    const int p = __builtin_amdgcn_cvt_pk_fp8_f32(f, 0.f, 0, false);
    return (uint8_t)(p & 0xFF);
}

// ---- wavefront reductions (ds_bpermute / DPP through __shfl_xor, 64 lanes) ---------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// The same two reductions without the LDS crossbar (every __shfl_xor above is a ds_bpermute: six dependent LDS round trips per
// call): four DPP row shifts leave each 16-lane row's result in its lane 15, four v_readlane pick them up.  ALL 64 lanes must be
// active.  The result is wave-uniform.  (Summation order differs from wave_sum's butterfly.)
__device__ __forceinline__ float wave_sum_dpp(float v) {
    int x = __float_as_int(v);
    x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true)));   // row_shr:1
    x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true)));   // row_shr:2
    x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true)));   // row_shr:4
    x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true)));   // row_shr:8
    return (__int_as_float(__builtin_amdgcn_readlane(x, 15)) + __int_as_float(__builtin_amdgcn_readlane(x, 31))) +
           (__int_as_float(__builtin_amdgcn_readlane(x, 47)) + __int_as_float(__builtin_amdgcn_readlane(x, 63)));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
    int x = __float_as_int(v);                                     // lanes without a source keep their own value (old = x, no bound_ctrl)
    x = __float_as_int(fmaxf(__int_as_float(x), __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x111, 0xf, 0xf, false))));
    x = __float_as_int(fmaxf(__int_as_float(x), __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x112, 0xf, 0xf, false))));
    x = __float_as_int(fmaxf(__int_as_float(x), __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x114, 0xf, 0xf, false))));
    x = __float_as_int(fmaxf(__int_as_float(x), __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x118, 0xf, 0xf, false))));
    return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(x, 15)), __int_as_float(__builtin_amdgcn_readlane(x, 31))),
                 fmaxf(__int_as_float(__builtin_amdgcn_readlane(x, 47)), __int_as_float(__builtin_amdgcn_readlane(x, 63))));
}
// reduce over lane groups of width W (power of two <= 64); result valid in every lane of the group
template <int W>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum for blocks of up to 1024 threads; `red` must hold >= 16 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = (lane < nw) ? red[lane] : 0.f;
    t = wave_sum(t);
    __syncthreads();
    return t;
}

// ---- per-DEVICE "done once" flags and the CU count (host side).  A process that drives more than one device has to set kernel attributes
// (hipFuncSetAttribute is per device) and size its grids on EACH of them: a process-wide `static bool` set the attribute on the first
// device only and the second device's launches failed or inherited the wrong CU count.
#include <atomic>
struct Mi355DevOnce {
    std::atomic<unsigned long long> mask{0};
    static int dev() { int d = 0; (void)hipGetDevice(&d); return d & 63; }
    bool done() const { return (mask.load(std::memory_order_acquire) >> dev()) & 1ull; }
    void set() { mask.fetch_or(1ull << dev(), std::memory_order_release); }
};
static inline int mi355_num_cus() {
    static std::atomic<int> cus[64];
    const int d = Mi355DevOnce::dev();
    int n = cus[d].load(std::memory_order_relaxed);
    if (n == 0) {
        hipDeviceProp_t prop;
        n = (hipGetDeviceProperties(&prop, d) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        cus[d].store(n, std::memory_order_relaxed);
    }
    return n;
}
