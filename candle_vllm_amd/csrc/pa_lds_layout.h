// LDS stage layout and DMA helper shared by the LDS-DMA attention kernels (paged_attention.hip: paged_attn_lds_kernel,
// paged_attn_stream_kernel; prefill_attention.hip: prefill_attn_lds_kernel) and by the host-side check of the permutations
// (mi355_internal_pal_layout, tests/test_cpu_pa_lds_layout.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// layout of one 64-token stage in LDS: K 16 KiB [16 channel groups][64 slots of 16 B], V 16 KiB [128 channels][8 slots of 16 B].
// The same functions serve the kernel and the host-side check of the permutations (mi355_internal_pal_layout,
// tests/test_cpu_c_abi.py): what the DMA writes where, and what every lane's fragment read finds there.
__host__ __device__ inline int pal_k_slot_token(int slot) {           // token (0..63 of the stage) held by K slot `slot` of every row
    return 32 * (slot >> 5) + 8 * ((slot & 15) >> 2) + (slot & 3) + 4 * ((slot >> 4) & 1);
}
__host__ __device__ inline int pal_v_swz(int ch, int slot) { return slot ^ ((ch >> 1) & 7); }   // LDS slot <-> row slot (an involution)
__host__ __device__ inline int pal_k_read_off(int j, int kg, int ip, int it, int r) { return (4 * j + kg) * 1024 + (32 * ip + 16 * it + r) * 16; }
__host__ __device__ inline int pal_v_read_off(int ch, int ip, int kg) { return 16384 + ch * 128 + pal_v_swz(ch, 4 * ip + kg) * 16; }
// DMA piece q (0..31, 1 KiB each, lane-linear destination q * 1024 + lane * 16): the 16 bytes lane `lane` copies --
// K (q < 16): channel group q, token pal_k_slot_token(lane); V: channel 8 (q - 16) + (lane >> 3), tokens 8 g .. 8 g + 7
__host__ __device__ inline int pal_dma_row(int q, int lane) { return q < 16 ? q : 8 * (q - 16) + (lane >> 3); }
__host__ __device__ inline int pal_dma_token(int q, int lane) {
    return q < 16 ? pal_k_slot_token(lane) : 8 * pal_v_swz(8 * (q - 16) + (lane >> 3), lane & 7);
}

// ---- the 64-token stage of the e4m3fn cache (paged_attn_stream_kernel<.., KV8>): 16 KiB = K [8 channel groups][64 slots of 16 B: one
// token's 16 channels, slot order pal_k_slot_token] | V [128 channels][4 slots of 16 B: 16 tokens], slot s of channel ch holds row slot
// s ^ ((ch >> 2) & 3).  DMA piece q (0..15, 1 KiB, lane-linear): K (q < 8) channel group q, token pal_k_slot_token(lane); V channel
// 16 (q - 8) + (lane >> 2), tokens 16 g .. 16 g + 15 with g = pal8_v_swz(channel, lane & 3).
__host__ __device__ inline int pal8_v_swz(int ch, int slot) { return slot ^ ((ch >> 2) & 3); }
// 8 bytes = channels 32 j + 8 kg .. + 7 of the token in K slot 32 ip + 16 it + r
__host__ __device__ inline int pal8_k_read_off(int j, int kg, int ip, int it, int r) { return (2 * j + (kg >> 1)) * 1024 + (32 * ip + 16 * it + r) * 16 + 8 * (kg & 1); }
// 4 bytes = tokens 32 ip + 8 kg + 4 it .. + 3 of channel ch
__host__ __device__ inline int pal8_v_read_off(int ch, int ip, int kg, int it) {
    return 8192 + ch * 64 + pal8_v_swz(ch, 2 * ip + (kg >> 1)) * 16 + 8 * (kg & 1) + 4 * it;
}
__host__ __device__ inline int pal8_dma_row(int q, int lane) { return q < 8 ? q : 16 * (q - 8) + (lane >> 2); }
__host__ __device__ inline int pal8_dma_token(int q, int lane) {
    return q < 8 ? pal_k_slot_token(lane) : 16 * pal8_v_swz(16 * (q - 8) + (lane >> 2), lane & 3);
}

__device__ __forceinline__ void pa_dma16(const uint8_t* gsrc_lane, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc_lane), "s"(lds_dst) : "memory");
}

