/* mi355_vllm.h -- C ABI of the MI355X (gfx950) paged-attention + quantised-matmul decode path.
 *
 * Drop-in boundary for candle-vllm's `src/backend` (reference @ /root/reference, crate v0.8.9).
 * The reference binds its device code through `attention_rs::kernels::ffi` (an `extern "C"` block:
 * void return, raw device pointers, i32 dims, the stream handle as i64 -- src/backend/cache.rs:127-162,
 * src/backend/gptq.rs:99-199) plus the Rust-level API of attention-rs / candle that has no C spelling
 * (PagedAttention::forward, FusedRope, cache::swap_blocks, QMatMul::forward, rms_norm ...).
 *
 * Conventions (identical to the reference FFI unless stated):
 *   - every entry point enqueues asynchronously on `stream` (a hipStream_t passed as int64_t; 0 = null stream);
 *   - pointers are raw DEVICE pointers unless the comment says HOST; the caller owns all buffers;
 *   - symbols spelled exactly like the reference FFI (`copy_blocks_bf16` ...) keep its `void` signature
 *     verbatim so the Rust `extern "C"` block binds unchanged;
 *   - `mi355_*` entry points return an int status (0 = ok, otherwise a hipError_t value) because the
 *     Python/C++ harness has no other error channel; a Rust binding may ignore it.
 *   - no torch / candle types anywhere in this file;
 *   - threading: one caller thread per process drives the device, as in the reference (one engine thread per rank bound
 *     to its GPU, llm_engine.rs:1406-1414; all kernels on that device's single non-default stream, lib.rs:109).  The
 *     library keeps grow-only device scratch (split-K partial sums, staged activation images, arrival counters) per
 *     PROCESS: calls are not re-entrant across host threads, and launches that use the scratch must be stream-ordered.
 */
#ifndef MI355_VLLM_H
#define MI355_VLLM_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element dtype codes */
#define MI355_DTYPE_F32 0
#define MI355_DTYPE_F16 1
#define MI355_DTYPE_BF16 2
#define MI355_DTYPE_U8 3
/* KV cache layouts -- src/scheduler/cache_engine.rs:298-341 */
#define MI355_KV_FLASH 0 /* K,V [num_blocks, block_size, num_kv_heads, head_dim]            (:326-341) */
#define MI355_KV_PAGED 1 /* K [nb, Hkv, D/x, bs, x], V [nb, Hkv, D, bs], x = 16/elem_size   (:298-324) */
#define MI355_KV_PAGED_FP8 2 /* the same with 1-byte OCP e4m3fn elements (x = 16): `--kvcache-dtype fp8` (host layers) */
/* ggml tensor type ids (GGUF) */
#define MI355_GGML_Q4_K 12
#define MI355_GGML_Q6_K 14
#define MI355_GGML_Q8_0 8      /* TP shards the reference re-quantises (quantized_var_builder.rs:234-269); native blocks, no repack */
/* swap directions */
#define MI355_SWAP_H2D 0
#define MI355_SWAP_D2H 1
#define MI355_SWAP_D2D 2
/* fused epilogues of mi355_qmatmul_fused */
#define MI355_EPI_STORE 0          /* out[t][row] = y (+bias)                                         */
#define MI355_EPI_RESID 1          /* out[t][row] = residual[t][row] + y (+bias)   (out may alias)    */
#define MI355_EPI_SILU_MUL 2       /* 2 segments (gate, up): out[t][row] = silu(gate)*up             */
#define MI355_EPI_QKV_ROPE_CACHE 3 /* 3 segments (q,k,v): interleaved RoPE on q,k; bf16 cast; q -> q_out,
                                      k,v scattered into the paged cache at slot_mapping[t]          */

/* ---------------------------------------------------------------------------------------------
 * 1. Reference FFI symbols, verbatim (attention_rs::kernels::ffi).
 * ------------------------------------------------------------------------------------------- */
/* replaces attention_rs::kernels::ffi::copy_blocks_{bf16,f16,f32} -- src/backend/cache.rs:127-162.
 * key_cache_ptrs / value_cache_ptrs: HOST arrays u64[num_layers] of device addresses;
 * block_mapping: HOST i64[2*num_pairs] = (src,dst) pairs; numel_per_block = elements of key_cache[0]. */
void copy_blocks_bf16(void* key_cache_ptrs, void* value_cache_ptrs, const void* block_mapping,
                      int32_t num_layers, int32_t num_pairs, int32_t numel_per_block, int64_t stream);
void copy_blocks_f16(void* key_cache_ptrs, void* value_cache_ptrs, const void* block_mapping,
                     int32_t num_layers, int32_t num_pairs, int32_t numel_per_block, int64_t stream);
void copy_blocks_f32(void* key_cache_ptrs, void* value_cache_ptrs, const void* block_mapping,
                     int32_t num_layers, int32_t num_pairs, int32_t numel_per_block, int64_t stream);
/* same, for the fp8 KV cache the reference stores as u8 (src/main.rs:263-267); not in the reference FFI */
void copy_blocks_u8(void* key_cache_ptrs, void* value_cache_ptrs, const void* block_mapping,
                    int32_t num_layers, int32_t num_pairs, int32_t numel_per_block, int64_t stream);

/* ---------------------------------------------------------------------------------------------
 * 2. attention-rs Rust-level API re-expressed as C.
 * ------------------------------------------------------------------------------------------- */
/* replaces attention_rs::cache::swap_blocks(src, dst, &HashMap) -- src/scheduler/cache_engine.rs:527-535.
 * mapping_pairs: HOST i64[2*num_pairs] (src_block, dst_block); bytes_per_block = elem_count/dim0*elem_size. */
int mi355_swap_blocks(const void* src, void* dst, const int64_t* mapping_pairs, int32_t num_pairs,
                      int64_t bytes_per_block, int32_t kind, int64_t stream);

/* the cache-write half of PagedAttention::forward -- src/openai/models/layers/attention.rs:983-995.
 * k, v: [num_tokens, num_kv_heads, head_dim] (elem_size bytes/elem, copied bit-exactly);
 * slot_mapping: i64 [num_tokens], slot = block*block_size + offset, negative = skip (llm_engine.rs:94). */
int mi355_reshape_and_cache(const void* k, const void* v, void* key_cache, void* value_cache,
                            const int64_t* slot_mapping, int32_t num_tokens, int32_t num_kv_heads,
                            int32_t head_dim, int32_t block_size, int32_t elem_size, int32_t layout,
                            int64_t stream);

/* the decode half of PagedAttention::forward (same call site; metadata src/openai/pipelines/inputs.rs:552-568).
 * q, out: [num_seqs, num_heads, head_dim] 16-bit (dtype = MI355_DTYPE_BF16 / F16, also the cache dtype);
 * block_tables: u32 [num_seqs, max_blocks_per_seq] (0-padded); context_lens: u32 [num_seqs];
 * softcap <= 0 disables soft-capping.  v1 = one partition per sequence; v2 = context split into
 * partition_size-token partitions merged by a log-sum-exp reduce
 * (tmp_out f32 [num_seqs,num_heads,P,head_dim], exp_sums/max_logits f32 [num_seqs,num_heads,P],
 *  P = ceil(max_context_len/partition_size)).  On the PAGED layout (bf16, head_dim 64 / 128, <= 16 query heads per kv head,
 *  block_size % 16 == 0) the MFMA kernel serves partition_size 32 / 64 / 128 (one wave per partition) and 256 / 512 (one wave
 *  walks its chunk 32 tokens at a time; needs partition_size % block_size == 0); every other size runs the generic kernel,
 *  which is far slower at long contexts -- the step drivers use 32. */
int mi355_paged_attention_v1(void* out, const void* q, const void* key_cache, const void* value_cache,
                             const uint32_t* block_tables, const uint32_t* context_lens, int32_t num_seqs,
                             int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                             int32_t max_blocks_per_seq, int32_t max_context_len, float scale, float softcap,
                             int32_t layout, int32_t dtype, int64_t stream);
int mi355_paged_attention_v2(void* out, float* exp_sums, float* max_logits, float* tmp_out, const void* q,
                             const void* key_cache, const void* value_cache, const uint32_t* block_tables,
                             const uint32_t* context_lens, int32_t num_seqs, int32_t num_heads,
                             int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                             int32_t max_blocks_per_seq, int32_t max_context_len, int32_t partition_size,
                             float scale, float softcap, int32_t layout, int32_t dtype, int64_t stream);

/* `sliding_window` of PagedAttention::new (attention.rs:566-575,888-897; the reference hands it to attention-rs' paged kernel and, on
 * prompt steps, to its causal mask, layers/mask.rs:22-27): the query at position i sees keys max(0, i - w + 1) .. i.  Decode: position
 * i = context_len - 1, i.e. the last w cached keys (what vLLM gets by truncating the block table to the window; exact here for windows
 * that do not start on a block boundary).  sliding_window <= 0 delegates to the unwindowed entry point.  No BASELINE model carries a
 * window: these are correctness paths (generic kernels, any head size <= 256, both layouts, bf16 / f16; no fp8 cache). */
int mi355_paged_attention_window(void* out, const void* q, const void* key_cache, const void* value_cache,
                                 const uint32_t* block_tables, const uint32_t* context_lens, int32_t num_seqs,
                                 int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                                 int32_t max_blocks_per_seq, int32_t max_context_len, float scale, float softcap,
                                 int32_t layout, int32_t dtype, int32_t sliding_window, int64_t stream);
int mi355_prefill_attention_window(void* out, const void* q, const void* k, const void* v, const void* key_cache,
                                   const void* value_cache, const uint32_t* block_tables, const uint32_t* context_lens,
                                   const uint32_t* cu_seqlens_q, int32_t num_seqs, int32_t max_seqlen_q,
                                   int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                                   int32_t max_blocks_per_seq, float scale, float softcap, int32_t layout,
                                   int32_t dtype, int32_t sliding_window, int64_t stream);

/* the prefill half of PagedAttention::forward (K4; attention.rs:707-719, metadata inputs.rs:90-230,351-367):
 * causal variable-length attention.  q, out [num_tokens, num_heads, head_dim], 16-bit `dtype`;
 * cu_seqlens_q u32 [num_seqs+1] delimits each sequence's chunk inside the flattened token dim.
 * key_cache == NULL: no cached prefix, K/V come from k, v [num_tokens, num_kv_heads, head_dim].
 * key_cache != NULL (prefix cache / chunked prefill, `use_cached_kv` inputs.rs:133-143): every key of a sequence
 * (cached prefix AND the current chunk, which reshape_and_cache has already written) is read from the paged cache
 * through block_tables; context_lens[i] = cached_i + chunk_i; query t of the chunk sees keys 0 .. cached_i + t. */
int mi355_prefill_attention(void* out, const void* q, const void* k, const void* v, const void* key_cache,
                            const void* value_cache, const uint32_t* block_tables, const uint32_t* context_lens,
                            const uint32_t* cu_seqlens_q, int32_t num_seqs, int32_t max_seqlen_q,
                            int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                            int32_t max_blocks_per_seq, float scale, float softcap, int32_t layout,
                            int32_t dtype, int64_t stream);

/* fp8 KV cache (`--kvcache-dtype fp8`: U8 cache tensors src/main.rs:263-267, K layout x = 16 cache_engine.rs:304-311,
 * PagedAttention built with is_fp8_keys attention.rs:574,896).  Values are stored as OCP e4m3fn(value / scale),
 * round-to-nearest-even, saturating; k, v, q, out are bf16.  PAGED layout for the attention entry points. */
int mi355_reshape_and_cache_fp8(const void* k, const void* v, void* key_cache, void* value_cache,
                                const int64_t* slot_mapping, int32_t num_tokens, int32_t num_kv_heads,
                                int32_t head_dim, int32_t block_size, int32_t layout, float k_scale, float v_scale,
                                int64_t stream);
/* PARITY MODE (tests): decode attention with the reference CPU path's rounding points -- `NaiveAttention::forward` on bf16
 * tensors, models/mod.rs:1288-1306: bf16 scores, `* scale` in bf16, bf16 probabilities, f32-accumulated P.V returned as bf16 --
 * and the oracle's summation orders.  bf16 q / cache / out, PAGED or FLASH layout; one workgroup per (head, sequence), slow by
 * design.  The product kernels above keep scores and probabilities in f32.  mi355_llama_set_attention_numerics(model, 1) routes
 * the GGUF host layer's decode steps through it. */
int mi355_paged_attention_reference_numerics(void* out, const void* q, const void* key_cache, const void* value_cache,
                                             const uint32_t* block_tables, const uint32_t* context_lens, int32_t num_seqs,
                                             int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                                             int32_t max_blocks_per_seq, int32_t max_context_len, float scale, int32_t layout,
                                             int64_t stream);
/* partition_size 0 = one pass (v1; exp_sums / max_logits / tmp_out may be NULL), else v2 temporaries as above */
int mi355_paged_attention_fp8(void* out, float* exp_sums, float* max_logits, float* tmp_out, const void* q,
                              const void* key_cache, const void* value_cache, const uint32_t* block_tables,
                              const uint32_t* context_lens, int32_t num_seqs, int32_t num_heads,
                              int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                              int32_t max_blocks_per_seq, int32_t max_context_len, int32_t partition_size,
                              float scale, float softcap, float k_scale, float v_scale, int64_t stream);
int mi355_prefill_attention_fp8(void* out, const void* q, const void* key_cache, const void* value_cache,
                                const uint32_t* block_tables, const uint32_t* context_lens,
                                const uint32_t* cu_seqlens_q, int32_t num_seqs, int32_t max_seqlen_q,
                                int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                                int32_t max_blocks_per_seq, float scale, float softcap, float k_scale,
                                float v_scale, int32_t dtype, int64_t stream);

/* replaces attention_rs::fused_rope::FusedRope::apply_inplace[_partial] -- layers/rotary_emb.rs:58-70.
 * q [T,H,D], k [T,Hkv,D] rotated in place; cos/sin f32 [max_seq, rotary_dim/2]; positions i64 [T];
 * is_rope_i != 0 -> interleaved pairs (GGUF llama), else half-split ("neox"). dtype F32 or BF16. */
int mi355_rope_inplace(void* q, void* k, const float* cos_table, const float* sin_table,
                       const int64_t* positions, int32_t num_tokens, int32_t num_heads,
                       int32_t num_kv_heads, int32_t head_dim, int32_t rotary_dim, int32_t is_rope_i,
                       int32_t dtype, int64_t stream);

/* ---------------------------------------------------------------------------------------------
 * 3. candle ops on the path.
 * ------------------------------------------------------------------------------------------- */
/* candle_nn::ops::rms_norm(x, w, eps) -- layers/qrmsnorm.rs:28-31.  x,out [T,hidden]; dtype F32 or BF16. */
int mi355_rms_norm(void* out, const void* x, const void* weight, int32_t num_tokens, int32_t hidden,
                   float eps, int32_t dtype, int32_t weight_dtype, int64_t stream);
/* candle_nn::ops::silu(gate) * up -- quantized_llama.rs:33-37 */
int mi355_silu_mul(void* out, const void* gate, const void* up, int64_t n, int32_t dtype, int64_t stream);
/* candle_nn::LayerNorm with bias (StableLM, stable_lm.rs:61-72): f32 statistics, one rounding; dtype BF16 / F32 */
int mi355_layer_norm(void* out, const void* x, const void* w, const void* b, int32_t num_tokens, int32_t hidden,
                     float eps, int32_t dtype, int64_t stream);
/* residual add (quantized_llama.rs:464,470) */
int mi355_add_f32(float* out, const float* a, const float* b, int64_t n, int64_t stream);
/* Tensor::to_dtype between F32 and BF16 (attention.rs:977-981, 1004) */
int mi355_cast(void* out, const void* in, int64_t n, int32_t src_dtype, int32_t dst_dtype, int64_t stream);
/* Embedding::forward on the dequantised table (quantized_llama.rs:262-264, 450) */
int mi355_embedding_f32(float* out, const float* table, const uint32_t* ids, int32_t num_tokens,
                        int32_t hidden, int64_t stream);
/* logits.argmax(-1) -- src/openai/logits_processor.rs:92-95 (first maximum wins) */
int mi355_argmax_f32(uint32_t* out, const float* logits, int32_t batch, int32_t vocab, int64_t stream);

/* candle QTensor::dequantize on NATIVE GGUF blocks (quantized_llama.rs:262-264) -> f32 */
int mi355_dequantize(float* out, const void* w_native, int32_t ggml_type, int64_t n_elems, int64_t stream);
/* candle QMatMul::forward reference path on NATIVE blocks (simple kernel; cross-check / small matrices) */
int mi355_qmatmul_ref(float* out, const float* x, const void* w_native, int32_t ggml_type, int32_t num_tokens,
                      int32_t n, int32_t k, int64_t stream);

/* One-time (load-time) re-tiling of a GGUF matrix [n_rows, k] into the MI355X tile order consumed by
 * mi355_qmatmul / mi355_qmatmul_fused.  HOST -> HOST; same byte count per weight; rows padded to 16. */
int64_t mi355_qweight_repacked_size(int32_t ggml_type, int64_t n_rows, int64_t k);
int mi355_qweight_repack(void* dst_host, const void* src_native_host, int32_t ggml_type, int64_t n_rows, int64_t k);

/* candle QMatMul::forward (attention.rs:920-922,1004; quantized_llama.rs:33-37):
 * out f32 [num_tokens, n] = x f32 [num_tokens, k] . dequant(W)^T (+ bias f32 [n] or NULL) */
int mi355_qmatmul(float* out, const float* x, const void* w_tiles, int32_t ggml_type, int32_t num_tokens,
                  int32_t n, int32_t k, const float* bias, int64_t stream);

/* Fused decode building block: [RMSNorm ->] up to 3 quantised matrices sharing x -> epilogue. */
typedef struct mi355_qmm_desc {
    int32_t nseg;               /* 1..3 weight matrices that share the input x                         */
    const void* w_tiles[3];     /* repacked weights                                                    */
    int32_t ggml_type[3];
    int32_t n_rows[3];          /* output rows of each matrix                                          */
    const void* x;              /* [num_tokens, ldx]; f32, or bf16 (x_dtype) e.g. the attention output   */
    int32_t x_dtype;            /* MI355_DTYPE_F32 / MI355_DTYPE_BF16                                  */
    int32_t ldx, k, num_tokens;
    const float* norm_weight;   /* non-NULL: x <- rms_norm(x, norm_weight, norm_eps) before the matmul */
    float norm_eps;
    int32_t epilogue;           /* MI355_EPI_*                                                         */
    float* out;                 /* f32 [num_tokens, ldo]                                               */
    int32_t ldo;
    const float* residual;      /* EPI_RESID                                                           */
    const float* bias;          /* f32 [sum n_rows] or NULL                                            */
    /* EPI_QKV_ROPE_CACHE only */
    const float* cos_table;
    const float* sin_table;
    const int64_t* positions;
    const int64_t* slot_mapping;
    void* q_out;                /* bf16 [num_tokens, num_heads*head_dim]                               */
    void* key_cache;            /* bf16 paged cache                                                    */
    void* value_cache;
    int32_t num_heads, num_kv_heads, head_dim, rotary_dim, block_size, kv_layout;
    /* mixture of experts (NULL / 0 = off): one launch covers moe_pairs (token, slot) pairs; pair p multiplies row
     * p / moe_x_div of x with expert moe_expert_ids[p] (DEVICE i32 [moe_pairs]) whose tiles start
     * moe_expert_stride[s] bytes after the previous expert's in w_tiles[s]; out row = p.  num_tokens is ignored;
     * epilogues STORE and SILU_MUL only. */
    const int32_t* moe_expert_ids;
    int32_t moe_pairs, moe_x_div;
    int64_t moe_expert_stride[3];
    /* chain hint (0 / NULL = off; 9..32 tokens only, ignored elsewhere): the NEXT mi355_qmatmul_fused call on this
     * stream reads this call's `out` as its x (k = chain_next_k = ldo, norm_weight = chain_next_norm or NULL).  The
     * epilogue then also stages that call's activation image, and the next call skips its own staging pass when its
     * arguments match (it silently stages itself when they do not). */
    int32_t chain_next, chain_next_k;
    const float* chain_next_norm;
    /* device-side gate (NULL = off; 9..32-token launches only): the launch is a no-op unless *rows_dev > rows_min -- a
     * graph-captured MoE step launches one fixed-shape chunk per (expert, 32 rows) and the chunks no pair landed in fall through */
    const int32_t* rows_dev;
    int32_t rows_min;
    /* grouped call (0 / 1 = off): group_count mat-muls of exactly these shapes in one call -- the experts of a mixture-of-experts layer
     * over their blocks of gathered rows (mi355_moe_group).  Group e reads x + e * group_x_stride elements, the weights
     * w_tiles[s] + e * moe_expert_stride[s] bytes, writes out + e * group_out_stride elements, and is gated by rows_dev[e] (rows_dev,
     * when given, then holds group_count counts).  Epilogues STORE and SILU_MUL, no bias, not together with moe_expert_ids.  With 9..32
     * tokens the groups are the z extent of ONE launch each of the staging, GEMM and epilogue kernels (and a chain hint stages all the
     * groups' next images); any other token count runs the groups one after the other. */
    int32_t group_count;
    int64_t group_x_stride, group_out_stride;
    /* grouped PROMPT-step call over a device block table (round 6; NULL = off; not together with moe_expert_ids / group_count): x holds the
     * rows of ALL groups, every group in whole 64-row blocks (mi355_moe_group_blocks); block b of the table = {group, one past the group's
     * last row} as two i32 ({0, 0}: no rows), the group's weights start group * moe_expert_stride[s] bytes into w_tiles[s].  ONE launch per
     * kernel walks every block; rows past a group's end are computed and never stored.  num_tokens = 64 * blocks (>= 96); the table needs
     * num_tokens / 64 + 2 entries.  Q4_K segments with epilogues STORE / RESID / SILU_MUL, or one Q6_K matrix with STORE / RESID (what the prompt
     * GEMMs' store loops fuse); any other
     * call is refused (hipErrorNotSupported) and the caller runs the groups one by one. */
    const int32_t* group_block_table;
} mi355_qmm_desc;
int mi355_qmatmul_fused(const mi355_qmm_desc* desc, int64_t stream);
/* MoE routing on the device (MlpOrMoe::forward, quantized_llama.rs:56-123, without the host round trip):
 * logits = gate_inp . rms_norm(x) ; softmax ; top-k (descending, ties -> lower expert id, as the stable sort there);
 * weights renormalised over the selected experts.  x f32 [T, hidden] (un-normalised residual stream when norm_weight
 * != NULL), gate_inp f32 [n_expert, hidden]; expert_ids i32 [T, k], weights f32 [T, k]. */
int mi355_moe_route(int32_t* expert_ids, float* weights, const float* x, const float* norm_weight, float norm_eps,
                    const float* gate_inp, int32_t num_tokens, int32_t hidden, int32_t n_expert, int32_t top_k,
                    int64_t stream);
/* ys[t] (+)= sum_j weights[t][j] * y_pairs[t*k + j]   (index_add of the weighted expert outputs); accumulate != 0
 * adds into ys (the residual stream), else overwrites */
int mi355_moe_combine(float* ys, const float* y_pairs, const float* weights, int32_t num_tokens, int32_t hidden,
                      int32_t top_k, int32_t accumulate, int64_t stream);
/* grouped experts for prompt steps / large batches (the reference's per-expert index_select -> expert -> index_add loop,
 * quantized_llama.rs:93-119; layers/moe.rs:746-810): `perm` = the (token * top_k + slot) pair indices sorted by expert,
 * `inv` = its inverse.  gather: dst[p] = src[perm[p] / top_k] (f32 rows); scatter_combine: ys[t] += sum_j w[t,j] * y_sorted[inv[t*top_k+j]] */
/* The same grouping decided on the device, for decode steps inside a captured graph (no host round trip, fixed launch shapes):
 * every expert owns `cap` rows (cap >= num_pairs) of the gathered buffers; mi355_moe_group writes pos[p] = expert_ids[p] * cap +
 * (earlier pairs of the same expert) and, when counts != NULL, the pairs per expert (the rows_dev gate of mi355_qmm_desc); mi355_moe_gather_pos copies token p / top_k into row pos[p]; the expert GEMMs then run over
 * all cap rows of every expert (rows are independent; unwritten rows are never read back) and mi355_moe_scatter_combine(inv = pos)
 * adds the weighted rows to the residual -- quantized_llama.rs:93-119 without `to_vec2`.  An expert id outside [0, n_expert) is sent to
 * the DUMP row n_expert * cap (size the row buffers n_expert * cap + 1): it joins no expert's block and is not counted. */
int mi355_moe_group(int32_t* pos, int32_t* counts, const int32_t* expert_ids, int32_t num_pairs, int32_t n_expert, int32_t cap, int64_t stream);
int mi355_moe_gather_pos(float* dst, const float* src, const int32_t* pos, int32_t num_pairs, int32_t top_k, int32_t hidden, int64_t stream);
/* prompt steps (round 6): the same grouping without a cap per expert and without the host -- every expert owns whole 64-row blocks, expert
 * e's rows start at 64 * (blocks of the experts before it); pos[p] = the row of pair p (stable), block_table[b] = {expert, row end} for the
 * n_blocks >= ceil(num_pairs / 64) + n_expert blocks ({0, 0} past the last expert's): the group_block_table of mi355_qmm_desc.  The
 * reference sorts on the host (quantized_llama.rs:70-91). */
int mi355_moe_group_blocks(int32_t* pos, int32_t* block_table, const int32_t* expert_ids, int32_t num_pairs, int32_t n_expert,
                           int32_t n_blocks, int64_t stream);
int mi355_moe_gather(float* dst, const float* src, const int32_t* perm, int32_t num_pairs, int32_t top_k, int32_t hidden, int64_t stream);
int mi355_moe_scatter_combine(float* ys, const float* y_sorted, const float* weights, const int32_t* inv, int32_t num_tokens,
                              int32_t hidden, int32_t top_k, int64_t stream);
/* A/B switches (process-global; for measurements and for tests that drive an alternative kernel through the same entry point).
 * No product path depends on a non-default value.  The PRODUCT library honours exactly these ten keys:
 *    3  decode-attention partition merge: low 4 bits 0 = separate reduce launch, 1 = in the last arriver where it pays (default),
 *       2 = always; bits 4.. = partitions per workgroup (0 = chosen per launch, else 1 | 4 | 8 | 16)
 *    5  decode-attention partition size of the step drivers (0 = their own choice: 32, or 64 for the balanced stream)
 *    6  prompt steps (>= 96 tokens) on the hand-written quantised GEMM (1, default) or streamed like decode batches (0)
 *    9  chained wide launches: an epilogue stages the next mat-mul's activation image (1, default)
 *   24  "exact" activations on the 9..32-token and prompt paths: f16 hi + lo planes instead of one f16 plane (0, default)
 *   30  16-bit / GPTQ linears, bit mask of folded launches switched OFF: 1 the 1..4-token 4-bit kernel, 2 no RMSNorm on the way in,
 *       4 RoPE + cache write in their own launch, 8 the LDS-shared-activation 16-bit kernel, 16 the one-pass 4-bit prompt GEMM;
 *       32 switches ON the 256-token tile of the 16-bit prompt GEMM where it fills the chip twice (64 as well: wherever it runs);
 *       128 the 5..48-token LDS-shared-activation 4-bit kernel, 256 its gate/up pairs split over two waves
 *   41  MoE decode steps group their (token, slot) pairs by expert on the device: 1 (default) = all experts in the z extent of one
 *       launch per kernel (mi355_qmm_desc.group_count); 2 = grouped, one launch group per expert; 0 = one mat-vec per pair
 *   44  decode-attention kernel per partition size on the PAGED bf16 cache: 1 (default) = 256 / 512 as looped chunks, 64 as the
 *       balanced LDS-DMA stream at >= 64 (sequence, kv head) pairs; 0 = neither; 5 = chunks only; 3 = the stream for every launch
 *       with partition size 64; 2 = additionally 1024 / 2048 / 4096 through the chunked LDS-DMA kernel
 *   47  prompt attention: bit 0 = K / V through the LDS ring (bf16 cache, head_dim 128; default 1), bit 1 = fp8 caches on the generic kernel
 *   48  Q4_K prompt-step launches apply store / residual / SiLU * up in the GEMM's store loop (1, default)
 * mi355_tuning_supported(key) = 1 for these; mi355_get_tuning returns their live value.  Every other key (0-2, 8, 10-23, 33-38, 42, 49:
 * wave / tile / split geometry, ablation modes, the experiments that lost their A/B) exists in probe builds only
 * (-DMI355_QMM_PROBES, tools/build_probe_lib.sh, listed next to their variables in csrc/); the product library ignores them
 * (mi355_get_tuning: INT32_MIN). */
void mi355_set_tuning(int32_t key, int32_t value);
int32_t mi355_get_tuning(int32_t key);
int32_t mi355_tuning_supported(int32_t key);
/* experiments only: device buffer of uint64 [workgroup][16 waves][4] that the 1..8-token mat-vec fills with wall-clock
 * stamps (entry, main loop done, past the barrier, exit) while probe mode 7 is set (probe builds only); NULL switches it off */
int mi355_debug_set_timestamps(void* dev_ptr);
/* EXPERIMENTS, compiled into probe builds only (-DMI355_QMM_PROBES, tools/build_probe_lib.sh); in the product library the
 * three entry points below exist and refuse (hipErrorNotSupported / error word 0).
 * Single-token launches on an LDS-DMA loader / consumer engine (csrc/qmv_engine.inc; tuning keys 20 on / off, 21 consumer
 * waves, 22 ring KiB cap).  Its in-workgroup waits are bounded: a wait that gave up leaves a sticky device word (1 consumer
 * waited for data, 2 loader for ring space, 3 / 4 consumer rendezvous, 5 a chain edge) and the launch's outputs are garbage.
 * Reads the word into *out_host (synchronising copy) and clears it when reset != 0. */
int mi355_qmv_error(int32_t* out_host, int32_t reset);
/* A chain of n (<= 4) single-token launches in ONE persistent launch (csrc/qmv_chain.inc): descs[p + 1].x must be
 * descs[p].out (f32, one token); only the last may carry MI355_EPI_QKV_ROPE_CACHE.  This is the decode step's
 * wo + residual -> ffn_norm + gate/up + silu*mul -> down + residual -> next layer's attn_norm + q|k|v + RoPE + cache write
 * (quantized_llama.rs:424-506), bit-identical to the four mi355_qmatmul_fused calls.  `sync` = mi355_qmv_chain_sync_bytes()
 * bytes of device memory zeroed once (arrival counters, monotonic across launches and hipGraph replays).  The launch needs
 * every CU of the device to itself (one resident workgroup per CU; the phase boundaries are grid-wide dependencies with
 * bounded waits, mi355_qmv_error).  Returns hipErrorNotSupported (801) when the chain does not fit -- issue the calls one by
 * one then.  Tuning key 23 = 1 switches chaining on (probe builds; measured slower than launch by launch). */
int mi355_qmv_chain_sync_bytes(void);
int mi355_qmatmul_chain(const mi355_qmm_desc* descs, int32_t n, void* sync, int64_t stream);

/* ---------------------------------------------------------------------------------------------
 * 3b. Safetensors path: dense 16-bit and GPTQ/AWQ/Marlin 4-bit linears (decode-shaped, any num_tokens).
 * ------------------------------------------------------------------------------------------- */
/* Reference FFI symbols, verbatim (attention_rs::kernels::ffi, src/backend/gptq.rs:99-199,313-332).
 * in [m,k] 16-bit; qweight = output of gptq_repack / awq_repack (opaque to the caller, shape [k/16, 2n] u32);
 * scales [k/g, n] 16-bit, Marlin-permuted by the caller (linear.rs:341-379); zeros: NULL (sym, z = 8) or the
 * Marlin-permuted packed zero points [k/g, n/8] u32 (AWQ, examples/convert_awq_marlin.py:72-115);
 * g_idx, workspace: accepted and unused (no act-order on this path; no global reduction locks needed);
 * out [m,n] 16-bit; group_size 32/64/128/multiple of 256/-1. */
void marlin_4bit_f16(const void* in, const int32_t* qweight, const void* scales, const void* zeros, const void* g_idx,
                     void* out, int32_t m, int32_t k, int32_t n, const void* workspace, int32_t group_size, int64_t stream);
void marlin_4bit_bf16(const void* in, const int32_t* qweight, const void* scales, const void* zeros, const void* g_idx,
                      void* out, int32_t m, int32_t k, int32_t n, const void* workspace, int32_t group_size, int64_t stream);
void marlin_awq_4bit_f16(const void* in, const int32_t* qweight, const void* scales, const void* zeros, const void* g_idx,
                         void* out, int32_t m, int32_t k, int32_t n, const void* workspace, int32_t group_size, int64_t stream);
void marlin_awq_4bit_bf16(const void* in, const int32_t* qweight, const void* scales, const void* zeros, const void* g_idx,
                          void* out, int32_t m, int32_t k, int32_t n, const void* workspace, int32_t group_size, int64_t stream);
/* exllama-style GPTQ (act-order), f16 only -- gptq.rs:181-197.  bit = 4 | 8 (linear.rs:215-217), pack = 32 / bit:
 * a [m,k] f16; b_q_weight [k/pack,n] u32 (checkpoint layout); qzeros [k/g, n/pack] u32 (stored zero - 1); scales [k/g,n]
 * f16; g_idx [k] (required); c [m,n] f16.
 * These symbols return nothing (the reference's signatures).  A call that cannot be served -- other bit widths, n % 16,
 * k % 32 (marlin_*: k % 256), null pointers -- records its error code (mi355_last_error) and fills the 16-bit output
 * with 0xFFFF (NaN), so a wrong configuration never reads as data. */
void gemm_half_q_half_alt(const void* a, const uint32_t* b_q_weight, const uint32_t* b_gptq_qzeros, const void* b_gptq_scales,
                          const int32_t* b_g_idx, void* c, int32_t m, int32_t n, int32_t k, int32_t bit, int64_t stream);
/* load-time repack -- gptq.rs:313-332.  gptq: in [k_packed = k/8, n] u32; awq: in [k, n_packed = n/8] u32
 * (AutoAWQ nibble order [0,2,4,6,1,3,5,7]); out: k*n/8 u32 in the layout the marlin_* entry points consume. */
void gptq_repack(const void* in, void* out, int32_t k_packed, int32_t n, int64_t stream);
void awq_repack(const void* in, void* out, int32_t k, int32_t n_packed, int32_t bits, int64_t stream);
/* first error recorded by a `void` entry point (copy_blocks_*, marlin_*, gemm_half_q_half_alt, *_repack) since the last
 * clear; 0 = none.  The reference bails in Rust around these calls (gptq.rs:196, linear.rs:215-217); a host that binds
 * this library checks here instead. */
int mi355_last_error(void);
void mi355_clear_error(void);
/* `checkpoint_format == "marlin"` (linear.rs:222-239,279-290; dims gptq.rs:46-49): the file's B [k/16, 2n] u32 is already in
 * the Marlin tile order and the reference feeds it to marlin_4bit_* without a repack.  This library's marlin_* entry
 * points stream the layout gptq_repack produces, so the integration converts B once at load time: out = k*n/8 u32,
 * k % 16 == 0, n % 64 == 0.  `s` (scales) needs nothing: it is Marlin-permuted in both cases. */
int mi355_marlin_format_repack(const void* in_B, void* out, int32_t k, int32_t n, int64_t stream);
/* the weight permutation of one 1024-value chunk of a Marlin tile row (host helper; -1 outside [0, 1024)) */
int32_t mi355_marlin_weight_perm(int32_t j);

/* position of natural column n inside a Marlin-permuted scale row (grouped: group_size < k; else "single"),
 * and inside a Marlin zero-point row (nibble index) -- host helpers, used by the kernels' index arithmetic */
int32_t mi355_marlin_scale_pos(int32_t n, int32_t grouped);
int32_t mi355_marlin_zero_pos(int32_t n);

#define MI355_ZERO_SYM8 0        /* z = 8                                                    */
#define MI355_ZERO_GPTQ_PLUS1 1  /* qzeros in checkpoint packing, z = stored + 1 (AutoGPTQ)  */
#define MI355_ZERO_AWQ_MARLIN 2  /* Marlin-permuted packed zero points, z = stored           */
/* `Linear::forward` (linear.rs:124-172): out = x . w^T (+bias), all tensors `dtype` (BF16/F16), w [n,k] row-major
 * as stored in the checkpoint.  epilogue: STORE; RESID (out = residual + y, out may alias residual);
 * SILU_MUL (w is the packed [gate;up] matrix of mlp.rs:324-352 with n = 2*intermediate rows: out [T, n/2] =
 * silu(gate)*up with candle's per-op rounding).  Every op result is rounded to `dtype` as candle does. */
int mi355_linear(void* out, const void* x, const void* w, const void* bias, const void* residual, int32_t num_tokens,
                 int32_t n, int32_t k, int32_t dtype, int32_t epilogue, int64_t stream);
/* The same op over the TILED weight image: w [n][k] re-ordered once at load time into 16-row x 256-k tiles, element (n, k) at
 *   ((((n/16) * (k_total/256) + k/256) * 8 + (k%256)/32) * 64 + 16 * ((k%32)/8) + n%16) * 8 + k%8
 * -- the MFMA fragment order: fragment j of a (tile, k-block) is 1 KiB contiguous (a row-major tile is sixteen 512-byte pieces
 * 2 k bytes apart; measured bound of that pattern: 3.9 TB/s).  n % 16 == 0, k % 256 == 0.  The host layer (mi355_dense_*) keeps
 * its 16-bit projections in this form.  mi355_dense_tile_repack: device to device, in != out, row stride ld_in elements. */
int mi355_linear_tiled(void* out, const void* x, const void* w_tiled, const void* bias, const void* residual, int32_t num_tokens,
                       int32_t n, int32_t k, int32_t dtype, int32_t epilogue, int64_t stream);
int mi355_dense_tile_repack(const void* in, void* out, int32_t n, int32_t k, int64_t ld_in, int64_t stream);
int64_t mi355_dense_tile_index(int32_t n_row, int32_t k_col, int32_t n, int32_t k);
/* `QLinear::forward` GPTQ arm (linear.rs:854-906) with the same fused epilogues; qweight [k/8, n] u32. */
int mi355_gptq_linear(void* out, const void* x, const void* qweight, const void* scales, const void* qzeros,
                      int32_t zero_mode, int32_t scales_permuted, const void* bias, const void* residual,
                      int32_t num_tokens, int32_t n, int32_t k, int32_t group_size, int32_t dtype, int32_t epilogue,
                      int64_t stream);
/* The same op over the TILED weight image: qweight re-ordered once at load time into 16-column x 256-k tiles, word (kr = k/8, n)
 * at  ((((n/16) * (k/256) + kr/32) * 2 + (kr%32)/16) * 64 + 16 * (kr%4) + n%16) * 4 + ((kr%32)/4) % 4  -- one wave's share of a
 * k-block is two contiguous 1 KiB runs.  This is the image gptq_repack / awq_repack / mi355_marlin_format_repack write and the
 * marlin_* entry points read (the slot the reference fills with gptq_marlin_repack's output, linear.rs:845-853).
 * mi355_gptq_tile_repack: columns [0, n_in) of a checkpoint-layout tensor [k/8][n_in] -> tiles tile0 .. of `out` (k % 256 == 0,
 * n_in % 16 == 0, in != out); mi355_gptq_tile_unpack: the inverse over a whole [k/8][n] tensor. */
int mi355_gptq_linear_tiled(void* out, const void* x, const void* qweight_tiled, const void* scales, const void* qzeros,
                            int32_t zero_mode, int32_t scales_permuted, const void* bias, const void* residual,
                            int32_t num_tokens, int32_t n, int32_t k, int32_t group_size, int32_t dtype, int32_t epilogue,
                            int64_t stream);
int mi355_gptq_tile_repack(const void* in, void* out, int32_t k, int32_t n_in, int32_t tile0, int64_t stream);
int mi355_gptq_tile_unpack(const void* in, void* out, int32_t k, int32_t n, int64_t stream);
/* host helper: word index of qweight[kr][n] inside the tiled image of a [k/8][n_total] tensor (-1 when the shape is not tiled) */
int64_t mi355_gptq_tile_index(int32_t kr, int32_t n, int32_t k, int32_t n_total);

/* RoPE cos/sin table builders (HOST): DefaultRotaryEmbedding::new / ScalingRotaryEmbedding::new
 * (layers/rotary_emb.rs:14-48,107-341,358-457) in the reference's f32 arithmetic.  Tables are f32
 * [n_positions, rotary_dim/2].  0 / negative fields of the scaling struct mean "absent" (the reference's defaults). */
#define MI355_ROPE_DEFAULT 0
#define MI355_ROPE_LINEAR 1
#define MI355_ROPE_LLAMA3 2
#define MI355_ROPE_DYNAMIC 3
#define MI355_ROPE_YARN 4
typedef struct mi355_rope_scaling {
    int32_t type;                                   /* MI355_ROPE_* = rope_scaling["rope_type"]                     */
    double factor, low_freq_factor, high_freq_factor, original_max_position_embeddings, alpha;
    double beta_fast, beta_slow, attn_factor, extrapolation_factor;   /* yarn (defaults 32, 1, 1, 1)               */
} mi355_rope_scaling;
/* number of positions the reference tabulates for this configuration */
int32_t mi355_rope_table_len(const mi355_rope_scaling* sc, int32_t max_seq_len, int32_t max_position_embeddings);
int mi355_rope_tables(float* cos_out_host, float* sin_out_host, int32_t rotary_dim, int32_t n_positions, double rope_theta,
                      const mi355_rope_scaling* sc /* NULL = default */, int32_t max_seq_len, int32_t max_position_embeddings);
/* `Config::effective_max_seq_len` (src/openai/models/mod.rs:663-702): max(base, round(original * factor)) for yarn with
 * factor > 1, the base otherwise -- the context length the KV cache and the tables above are sized for */
int64_t mi355_effective_max_seq_len(const mi355_rope_scaling* sc, int64_t base_max_position_embeddings);

/* ---------------------------------------------------------------------------------------------
 * 4. Host layer: the GGUF llama decode step (GGUFLLaMa::forward + CacheEngine + decode graph), C handles.
 *    Mirrors src/openai/models/quantized_llama.rs:424-506, src/scheduler/cache_engine.rs:122-341,
 *    src/backend/graph.rs:471-661,685.  Everything a Rust `DefaultPipeline::forward` arm would call.
 * ------------------------------------------------------------------------------------------- */
typedef struct mi355_llama_config {
    int32_t hidden, n_layers, n_heads, n_kv_heads, head_dim, intermediate, vocab;
    int32_t max_seq, block_size, kv_layout, max_batch, max_blocks_per_seq;
    float rms_eps, rope_theta;
    int32_t tp_rank, tp_world;   /* heads / kv heads / rows are the LOCAL shard when tp_world > 1 */
    int32_t n_expert, n_expert_used;   /* > 1: Mixtral-style MoE MLP (llama.expert_count / expert_used_count) */
} mi355_llama_config;
/* weight slots */
#define MI355_W_WQ 0
#define MI355_W_WK 1
#define MI355_W_WV 2
#define MI355_W_WO 3
#define MI355_W_W1 4 /* ffn_gate */
#define MI355_W_W2 5 /* ffn_down */
#define MI355_W_W3 6 /* ffn_up */
#define MI355_W_ATTN_NORM 7
#define MI355_W_FFN_NORM 8
#define MI355_W_TOK_EMBD 9     /* layer = -1 */
#define MI355_W_OUTPUT_NORM 10 /* layer = -1 */
#define MI355_W_OUTPUT 11      /* layer = -1 */

void* mi355_llama_create(const mi355_llama_config* cfg);
void mi355_llama_destroy(void* model);
/* native GGUF blocks on the HOST -> repacked + uploaded (owned by the model) */
int mi355_llama_set_qweight(void* model, int32_t layer, int32_t which, int32_t ggml_type, const void* native_host,
                            int32_t n_rows, int32_t k);
/* already-repacked tiles on the DEVICE, borrowed (caller keeps them alive) */
int mi355_llama_set_qweight_tiles(void* model, int32_t layer, int32_t which, int32_t ggml_type,
                                  const void* tiles_dev, int32_t n_rows, int32_t k);
int mi355_llama_set_f32(void* model, int32_t layer, int32_t which, const float* host, int64_t n);
/* MoE layers (quantized_llama.rs:347-365): router `ffn_gate_inp` f32 [n_expert, hidden] through mi355_llama_set_f32
 * with which = MI355_W_GATE_INP; expert `e`'s ffn_gate / ffn_down / ffn_up (which = W1 / W2 / W3) as native GGUF
 * blocks -- all experts of one (layer, which) share the ggml type */
#define MI355_W_GATE_INP 12
int mi355_llama_set_moe_expert(void* model, int32_t layer, int32_t which, int32_t expert, int32_t ggml_type,
                               const void* native_host, int32_t n_rows, int32_t k);
int mi355_llama_alloc_kv_cache(void* model, int32_t num_blocks);
void* mi355_llama_kv_ptr(void* model, int32_t layer, int32_t which /* 0 = K, 1 = V */);
int64_t mi355_llama_kv_bytes_per_tensor(void* model);
/* copy a layer's K or V tensor from (to_model != 0) / to a host or device buffer */
int mi355_llama_kv_copy(void* model, int32_t layer, int32_t which, void* buf, int64_t bytes, int32_t to_model);
/* one eager decode step over DEVICE inputs (InputMetadata fields as raw pointers); logits f32 [batch, vocab] */
int mi355_llama_forward_decode(void* model, const uint32_t* tokens, const int64_t* positions,
                               const int64_t* slot_mapping, const uint32_t* block_tables,
                               const uint32_t* context_lens, int32_t batch, int32_t max_blocks,
                               int32_t max_context_len, float* logits, int64_t stream);
/* one PROMPT step (is_prefill; prepare_prompt fields as raw DEVICE pointers, inputs.rs:90-230): tokens of all
 * sequences flattened [num_tokens]; cu_seqlens_q u32 [num_seqs+1]; context_lens[i] = cached_i + chunk_i;
 * K/V of the chunk are written to the paged cache and attention (K4) reads prefix + chunk from it;
 * logits f32 [num_seqs, vocab] for the LAST chunk token of every sequence (quantized_llama.rs:495-505). */
int mi355_llama_forward_prefill(void* model, const uint32_t* tokens, const int64_t* positions,
                                const int64_t* slot_mapping, const uint32_t* block_tables,
                                const uint32_t* context_lens, const uint32_t* cu_seqlens_q, int32_t num_seqs,
                                int32_t num_tokens, int32_t max_seqlen_q, int32_t max_blocks, float* logits,
                                int64_t stream);
/* greedy decode loop on static device buffers; the step is replayed from a hipGraph when stream != 0 */
int mi355_llama_decode_begin(void* model, const uint32_t* tokens_host, const uint32_t* seq_lens_host,
                             const uint32_t* block_tables_host, int32_t batch, int32_t max_blocks, int32_t ctx_cap,
                             int64_t stream);
/* enable: 0 = eager steps, 1 (or 2) = hipGraph replay (default).  Tensor-parallel steps are captured too when the
 * communicator is device-native (RCCL on its side stream joins the capture as a fork / join, the one-shot peer kernel is a
 * plain node); with host-supplied collectives (mi355_comm_create_external) they stay eager. */
int mi355_llama_set_graph(void* model, int32_t enable);
/* parity mode of the decode steps (tests only): 0 = product kernels, 1 = decode attention through
 * mi355_paged_attention_reference_numerics, 2 = that AND every quantised mat-vec of the step exact to f32 rounding (weights dequantised
 * as the oracle's O1, f64 sums, the product's own fused epilogues; csrc/qmm_exact.inc -- the switch behind it is process-wide:
 * mi355_internal_qmm_set_exact).  Mode 2 exists to show that the end-to-end distance to the oracle is amplified per-product rounding
 * noise: with it the distance falls to the oracle's own f64-vs-f32-sum self-spread. */
int mi355_llama_set_attention_numerics(void* model, int32_t mode);
void mi355_internal_qmm_set_exact(int32_t on);
int32_t mi355_internal_qmm_get_exact(void);
int mi355_llama_decode_step(void* model, int64_t stream);
/* bookkeeping of the greedy loop for benchmarks: step graphs captured + instantiated so far, and steps that ran eagerly (the first
 * step of every new (batch, max_blocks, ctx_cap) shape, every step with graphs off).  A timed region bracketed by these shows that no
 * capture and no eager step fell inside it (graph.rs:471-661 captures at load; here the second step of a shape captures). */
int64_t mi355_llama_graph_captures(void* model);
int64_t mi355_llama_eager_steps(void* model);
int mi355_llama_decode_read_tokens(void* model, uint32_t* host_out, int64_t stream);
float* mi355_llama_logits_ptr(void* model);
/* tensor parallel: RCCL communicator (one process per GPU).  rank 0: mi355_comm_unique_id -> 128 bytes that the
 * launcher ships to every rank -> mi355_llama_init_comm on every rank (cudarc Comm::from_rank, pipeline.rs:805-812) */
int mi355_comm_unique_id(void* out128);
int mi355_llama_init_comm(void* model, const void* id128);
int mi355_llama_set_comm(void* model, void* comm);          /* borrowed mi355_comm_* handle instead of init_comm */
void* mi355_llama_comm_handle(void* model);                 /* the model's communicator (pipeline.rs:805-812), e.g. for mi355_comm_set_options */
/* replace the default RoPE tables (built at create from rope_theta) by scaled ones: HOST f32 [n_positions >= max_seq, head_dim/2] */
int mi355_llama_set_rope_tables(void* model, const float* cos_host, const float* sin_host, int32_t n_positions);
/* measurement hook: one launch group of the step on the static inputs (part 0 qkv, 1 attention, 2 wo,
 * 3 gate/up, 4 down, 5 lm_head, 6 embedding) */
int mi355_llama_run_part(void* model, int32_t layer, int32_t part, int64_t stream);
/* parity hook: device pointer of one activation buffer of the decode step -- 0 = residual stream xs f32 [max_batch, hidden]
 * (`layer_in` of quantized_llama.rs:438-475), 1 = q bf16, 2 = attention output bf16, 3 = MLP intermediate f32.  With
 * mi355_llama_run_part a test can feed the oracle's layer input and compare one layer at a time at full geometry. */
void* mi355_llama_act_ptr(void* model, int32_t which);

/* generic communicator (the reference's per-process nccl `Comm`, pipeline.rs:805-812; collectives of
 * distributed.rs:547-654,1335-1446): RCCL bound by dlopen; id128 from mi355_comm_unique_id on rank 0 */
void* mi355_comm_create(const void* id128, int32_t rank, int32_t world);
/* ... or collectives the HOST supplies (it already owns a communicator, as the reference's Rust side does: both run in
 * stream order on `stream`; all_reduce is sum, in place; dtype = MI355_DTYPE_*; return 0 on success) */
typedef int (*mi355_allreduce_fn)(void* user, void* buf, int64_t count, int32_t dtype, int64_t stream);
typedef int (*mi355_allgather_fn)(void* user, const void* send, void* recv, int64_t count_per_rank, int32_t dtype, int64_t stream);
void* mi355_comm_create_external(mi355_allreduce_fn all_reduce, mi355_allgather_fn all_gather, void* user);
void mi355_comm_destroy(void* comm);
int mi355_comm_all_reduce(void* comm, void* buf, int64_t count, int32_t dtype, int64_t stream);   /* sum, in place */
/* side_stream: 1 (default) = RCCL calls run on the communicator's own stream, fenced by events against `stream`
 * (north_star; the reference's nccl ops run on the nccl stream, distributed.rs:547-654); 0 = in `stream`.
 * wire_bf16: 0 (default) = the f32 residual stream on the wire, rank 0's partial carries the residual (one rounding fewer
 * than the reference); 1 = the reference's numerics: every rank rounds its o_proj / down_proj partial to bf16, the sum is
 * taken in bf16 and added to the f32 residual (attention.rs:1003-1008, quantized_llama.rs:38-42). */
int mi355_comm_set_options(void* comm, int32_t side_stream, int32_t wire_bf16);
int mi355_comm_wire_bf16(void* comm);
/* resid[i] += sum over ranks of y[i] (f32, y is clobbered) with the communicator's wire numerics -- the call the GGUF host
 * layer makes after o_proj / down_proj in wire mode 1 (attention.rs:1003-1008) */
int mi355_comm_all_reduce_residual(void* comm, float* y, float* resid, int64_t count, int64_t stream);
/* One-shot peer-to-peer all-reduce for decode-sized f32 messages (<= 256 KiB; SURVEY 2.4: C1/C2 are 8-256 KiB and
 * latency-bound -- distributed.rs:547-654 runs them as ring all-reduces): every rank publishes its partial in a
 * fine-grained region, every rank sums all partials in rank order (bit-identical on all ranks), one kernel, replayable in
 * a hipGraph.  mi355_comm_p2p_export: allocate this rank's region, return its 64-byte IPC handle; the launcher gathers
 * the world's handles (as it ships the RCCL unique id, pipeline.rs:805-812); mi355_comm_p2p_attach: open the peers'
 * regions (handles = world x 64 bytes in rank order).  Works on RCCL and on host-supplied communicators; larger messages
 * and all-gathers keep their transport.  mi355_comm_p2p_error: 1 after a wait that ran into its spin bound. */
int mi355_comm_p2p_export(void* comm, void* handle_out64);
int mi355_comm_p2p_attach(void* comm, const void* handles, int32_t rank, int32_t world);
int mi355_comm_p2p_error(void* comm);
/* after mi355_comm_p2p_attach: 0 = route all-reduces through the communicator's own transport again (the regions stay open),
 * 1 = the peer kernel again.  For a launcher that self-tests the peer path across its ranks before the first step and falls
 * back on ALL ranks together when any of them failed (candle_vllm_amd/model.py::init_comm, bench.py --all-reduce auto). */
int mi355_comm_p2p_enable(void* comm, int32_t on);
/* 0 when EVERY collective flavour of a TP step can be captured in a hipGraph on this stack: captures a small and a
 * hidden-sized f32 all-reduce, a bf16 all-reduce and an all-gather on `stream`, instantiates and destroys the graph without
 * launching it (nothing runs on the wire), so every rank can test locally and the ranks agree on captured / eager steps
 * before the first real step.  A communicator that holds any host-supplied callback is refused (hipErrorNotSupported): a
 * host call made during capture is not replayed. */
int mi355_comm_capture_probe(void* comm, int64_t stream);
int mi355_comm_all_gather(void* comm, const void* send, void* recv, int64_t count, int32_t dtype, int64_t stream);

/* ---------------------------------------------------------------------------------------------
 * 4b. Host layer for 16-bit safetensors llama-family models (Llama / Qwen2 shapes): src/openai/models/llama.rs:
 *     39-201, layers/attention.rs:585-734, layers/mlp.rs:440-458.  Residual stream and every op result in the
 *     model dtype, rounded where candle rounds.  mi355_dense_forward is one eager step; mi355_dense_decode_* is the greedy loop on
 *     static device buffers replayed from a hipGraph (backend/graph.rs:471-661,685-807, pipelines/pipeline.rs:2091-2135).
 * ------------------------------------------------------------------------------------------- */
typedef struct mi355_dense_config {
    int32_t hidden, n_layers, n_heads, n_kv_heads, head_dim, intermediate, vocab;
    int32_t max_seq, block_size, kv_layout, max_batch, max_blocks_per_seq;
    float rms_eps, rope_theta;
    int32_t dtype;             /* MI355_DTYPE_BF16 */
    int32_t rope_interleaved;  /* 0 = half-split ("neox", HF llama / qwen / stablelm), 1 = interleaved */
    int32_t norm_type;         /* 0 = RMSNorm, 1 = LayerNorm with bias (StableLM, stable_lm.rs:61-72) */
    int32_t rotary_dim;        /* <= head_dim; StableLM: partial_rotary_factor 0.25 (stable_lm.rs:28); 0 = head_dim */
    int32_t kv_fp8;            /* 1 = `--kvcache-dtype fp8`: U8 e4m3fn cache (PAGED layout, x = 16), scale 1.0 */
    int32_t tp_rank, tp_world; /* tensor parallel: n_heads / n_kv_heads / intermediate / vocab are the LOCAL shard */
    int32_t vocab_total;       /* tensor parallel: the REAL vocabulary (VocabParallelLinear narrows the gathered row to it before sampling,
                                  distributed.rs:1657-1663; the lm_head shards of a padded vocabulary end in zero rows); logits rows, the
                                  greedy loop's argmax and the replicated embedding table have this many entries.  0 = vocab x tp_world */
} mi355_dense_config;
#define MI355_W_BQ 12 /* q_proj.bias (Qwen2, StableLM use_qkv_bias) */
#define MI355_W_BK 13
#define MI355_W_BV 14
#define MI355_W_ATTN_NORM_B 15   /* LayerNorm biases */
#define MI355_W_FFN_NORM_B 16
#define MI355_W_OUTPUT_NORM_B 17 /* layer = -1 */
void* mi355_dense_create(const mi355_dense_config* cfg);
void mi355_dense_destroy(void* model);
/* 16-bit tensor from the HOST in checkpoint layout [out, in]; slots MI355_W_* (W1 = gate_proj, W3 = up_proj are
 * packed into one gate_up matrix as mlp.rs:324-352 does) */
int mi355_dense_set_weight(void* model, int32_t layer, int32_t which, const void* host, int64_t n_elems);
int mi355_dense_set_weight_dev(void* model, int32_t layer, int32_t which, const void* dev, int64_t n_elems);
/* GPTQ 4-bit projection (QLinear GPTQ arm, linear.rs:854-906) instead of a dense one: qweight u32 [k/8, n], scales
 * 16-bit [k/g, n] from the HOST in checkpoint order; sym, no act-order (the Marlin-eligible case, linear.rs:319-325) */
int mi355_dense_set_gptq(void* model, int32_t layer, int32_t which, const void* qweight_host, const void* scales_host,
                         int32_t n, int32_t k, int32_t group_size);
/* tensor parallel (distributed.rs:243-249,492-534,696-711,1632-1667): comm from mi355_comm_create (borrowed);
 * all-reduce of the 16-bit stream after o_proj / down_proj, vocab-parallel lm_head + all-gather */
int mi355_dense_set_comm(void* model, void* comm);
int mi355_dense_set_rope_tables(void* model, const float* cos_host, const float* sin_host, int32_t n_positions);   /* [n, rotary_dim/2] */
int mi355_dense_alloc_kv_cache(void* model, int32_t num_blocks);
void* mi355_dense_kv_ptr(void* model, int32_t layer, int32_t which);
/* test hook (full-size parity legs): the following mi355_dense_forward calls run layers first..last only, from the 16-bit
 * residual stream xs_in_dev [num_tokens, hidden], and copy the stream after layer `last` to xs_out_dev (no embedding, no
 * head, `logits` untouched); first < 0 restores the whole forward */
int mi355_dense_set_layer_window(void* model, int32_t first, int32_t last, const void* xs_in_dev, void* xs_out_dev);
/* one step: prompt when cu_seqlens_q != NULL (flattened tokens), else decode (num_tokens == num_seqs);
 * DEVICE inputs as prepare_prompt / prepare_decode build them; logits f32 [num_seqs, vocab] */
int mi355_dense_forward(void* model, const uint32_t* tokens, const int64_t* positions, const int64_t* slot_mapping,
                        const uint32_t* block_tables, const uint32_t* context_lens, const uint32_t* cu_seqlens_q,
                        int32_t num_seqs, int32_t num_tokens, int32_t max_seqlen_q, int32_t max_blocks,
                        int32_t max_context_len, float* logits, int64_t stream);
/* Freeze the weights: the one-off re-ordering of the 16-bit projections into 16-row x 256-k tiles, which allocates and
 * synchronises (the first forward does it otherwise, and refuses to while its stream is capturing).  Call it from the loader
 * after the last setter; from then on mi355_dense_set_weight* / _set_gptq refuse every slot. */
int mi355_dense_finalize(void* model);
/* Greedy decode loop, same contract as mi355_llama_decode_begin / _step / _read_tokens: tokens_host[b] = last token of
 * sequence b, seq_lens_host[b] = its length including that token, block tables [batch, max_blocks] with every block of the
 * steps to come reserved, ctx_cap = upper bound of the context length over those steps.  A step = forward -> argmax (first
 * maximum, logits_processor.rs:92-95) -> next positions / slots / context lengths on the device; captured once per
 * (batch, max_blocks, ctx_cap) and replayed.  Tensor-parallel steps are captured when the communicator is RCCL's own, and
 * run eagerly with host-supplied collectives.  Whether a capture works is a property of the local stack: tensor-parallel callers
 * run mi355_comm_capture_probe on every rank, agree on the answer (min over the ranks) and switch graphs off EVERYWHERE
 * (mi355_dense_set_graph(model, 0)) unless every rank can capture -- the library cannot make that decision per rank without
 * leaving the other ranks inside a collective.  A step that fails leaves the loop where it was (the context length is advanced only
 * by a step that was enqueued).  Needs cfg.max_blocks_per_seq >= max_blocks. */
int mi355_dense_set_graph(void* model, int32_t enable);
int mi355_dense_decode_begin(void* model, const uint32_t* tokens_host, const uint32_t* seq_lens_host,
                             const uint32_t* block_tables_host, int32_t batch, int32_t max_blocks, int32_t ctx_cap, int64_t stream);
int mi355_dense_decode_step(void* model, int64_t stream);
int64_t mi355_dense_graph_captures(void* model);   /* as mi355_llama_graph_captures / _eager_steps */
int64_t mi355_dense_eager_steps(void* model);
int mi355_dense_decode_read_tokens(void* model, uint32_t* host_out, int64_t stream);
float* mi355_dense_logits_ptr(void* model);   /* f32 [batch, vocab (x tp_world)] of the loop's last step */

/* ---------------------------------------------------------------------------------------------
 * 5. Host block manager: BlockEngine + PrefixCache + Sequence bookkeeping + input preparation.
 *    Mirrors src/scheduler/block_engine.rs:190-1474, prefix_cache.rs:36-384, sequence.rs:90-300 and
 *    src/openai/pipelines/inputs.rs:90-230,376-454.  Pure host code (no device work).  A block is an int32
 *    code: id >= 0 on the GPU, -1-id on the CPU.  A "group" is passed as an array of sequence ids.
 * ------------------------------------------------------------------------------------------- */
void* mi355_be_create(int32_t block_size, int32_t num_gpu_blocks, int32_t num_cpu_blocks, int32_t prefix_cache_enabled,
                      int32_t max_cached_blocks);
void mi355_be_destroy(void* be);
int32_t mi355_be_num_free_blocks(void* be);
int32_t mi355_be_num_free_cpu_blocks(void* be);
int32_t mi355_be_num_blocks(void* be);
int32_t mi355_be_prefix_cache_blocks(void* be);
int32_t mi355_be_free_block_ids(void* be, int32_t* out, int32_t cap);
/* Sequence (_Sequence::new / add_token / prefill_chunk_tokens ...) */
int32_t mi355_be_seq_create(void* be, int64_t seq_id, const uint32_t* prompt, int32_t n);
int32_t mi355_be_seq_remove(void* be, int64_t seq_id);
int32_t mi355_be_seq_add_token(void* be, int64_t seq_id, uint32_t token);
int32_t mi355_be_seq_len(void* be, int64_t seq_id);
int32_t mi355_be_seq_logical_blocks(void* be, int64_t seq_id);
int32_t mi355_be_seq_get_cached_tokens(void* be, int64_t seq_id);
int32_t mi355_be_seq_set_cached_tokens(void* be, int64_t seq_id, int32_t n);
int32_t mi355_be_seq_set_warmup_tokens(void* be, int64_t seq_id, int32_t n);
int32_t mi355_be_seq_prefill_chunk_tokens(void* be, int64_t seq_id, int32_t chunk);
int32_t mi355_be_seq_has_prefix_hash(void* be, int64_t seq_id);
/* block tables */
int32_t mi355_be_block_table(void* be, int64_t seq_id, int32_t* out, int32_t cap);
int32_t mi355_be_block_refcount(void* be, int32_t block_code);
int32_t mi355_be_pop_back_block(void* be, int64_t seq_id);
/* can_allocate[_for_prefill]: 0 Ok / 1 Later / 2 Impossible (block_engine.rs:292-373); allocate[_for_prefill] */
int32_t mi355_be_can_allocate(void* be, const int64_t* seq_ids, int32_t n, int32_t chunk);
int32_t mi355_be_allocate(void* be, const int64_t* seq_ids, int32_t n, int32_t chunk);
int32_t mi355_be_can_append_token(void* be, const int64_t* seq_ids, int32_t n);
/* append_token_slot_to_seq: returns 1 and (src,dst) when the shared last block was copied-on-write */
int32_t mi355_be_append_token_slot(void* be, int64_t seq_id, int32_t* cow_src, int32_t* cow_dst);
int32_t mi355_be_prefill_chunk_blocks_required(void* be, const int64_t* seq_ids, int32_t n, int32_t chunk);
int32_t mi355_be_can_append_prefill_chunk(void* be, const int64_t* seq_ids, int32_t n, int32_t chunk);
int32_t mi355_be_append_prefill_chunk_slots(void* be, const int64_t* seq_ids, int32_t n, int32_t chunk);
int32_t mi355_be_free_sequence(void* be, int64_t seq_id);
int32_t mi355_be_cache_sequence(void* be, int64_t seq_id);
int32_t mi355_be_evict_prefix_cache_blocks(void* be, int32_t num_blocks);
int32_t mi355_be_evict_prefix_cache_until_free(void* be, int32_t min_free);
int32_t mi355_be_query_prefix_match_tokens(void* be, const uint32_t* tokens, int32_t n);
int32_t mi355_be_fallback_to_full_prefill(void* be, int64_t seq_id);
int32_t mi355_be_rebuild_with_cached_prefix(void* be, int64_t seq_id, int32_t cached_tokens);
/* swap: pairs out = (src_block, dst_block) to hand to mi355_swap_blocks; returns the pair count, or < 0 with NOTHING
 * changed: -1 unknown sequence, -2 not enough free blocks on the destination side, -3 `cap` pairs are not enough
 * (cap = the group's total block count always is) */
int32_t mi355_be_can_swap_out(void* be, const int64_t* seq_ids, int32_t n);
int32_t mi355_be_swap_in_required_blocks(void* be, const int64_t* seq_ids, int32_t n);
int32_t mi355_be_can_swap_in(void* be, const int64_t* seq_ids, int32_t n);
int32_t mi355_be_swap_out(void* be, int64_t group_id, const int64_t* seq_ids, int32_t n, int64_t* pairs, int32_t cap);
int32_t mi355_be_swap_in(void* be, int64_t group_id, const int64_t* seq_ids, int32_t n, int64_t* pairs, int32_t cap);
/* test hook: the next n_out swap_out calls / n_in swap_in calls are refused (-4) before anything is touched */
void mi355_be_test_refuse_swaps(void* be, int32_t n_out, int32_t n_in);
void mi355_be_finalize_swap_out(void* be, int64_t group_id);
void mi355_be_rollback_swap_out(void* be, int64_t group_id);
void mi355_be_finalize_swap_in(void* be, int64_t group_id);
void mi355_be_rollback_swap_in(void* be, int64_t group_id);
/* bare PrefixCache (the unit under test in prefix_cache.rs:386-599) */
void* mi355_pc_create(int32_t block_size, int32_t enabled, int32_t max_cached_blocks, int32_t num_block_ids);
int32_t mi355_pc_insert(void* pc, const uint32_t* tokens, int32_t n, const int32_t* blocks, int32_t nblocks, int32_t* evicted, int32_t cap);
int32_t mi355_pc_match(void* pc, const uint32_t* tokens, int32_t n, int32_t* blocks, int32_t cap);
int32_t mi355_pc_evict(void* pc, int32_t num, const uint32_t* protect_tokens, int32_t protect_n, int32_t* evicted, int32_t cap);
int32_t mi355_pc_cached_blocks(void* pc);
int32_t mi355_pc_lru_len(void* pc);
uint64_t mi355_pc_hash_for_blocks(void* pc, const uint32_t* tokens, int32_t n, int32_t full_blocks, int32_t has_seed, uint64_t seed, int32_t seed_block);
/* a1 / a2: InputMetadata arrays on the HOST from the engine state (inputs.rs:376-454 / :90-230) */
int32_t mi355_be_prepare_decode(void* be, const int64_t* seq_ids, int32_t n, uint32_t* tokens, int64_t* positions,
                                int64_t* slot_mapping, uint32_t* context_lens, uint32_t* block_tables, int32_t bt_cap_cols);
int32_t mi355_be_prepare_prompt(void* be, const int64_t* seq_ids, int32_t n, int32_t chunk, uint32_t* tokens,
                                int64_t* positions, int64_t* slot_mapping, uint32_t* context_lens, uint32_t* cu_q,
                                uint32_t* cu_k, uint32_t* block_tables, int32_t tok_cap, int32_t bt_cap_cols,
                                int32_t* max_blocks_out);

/* ---------------------------------------------------------------------------------------------
 * 6. Continuous-batching scheduler (src/scheduler/mod.rs:66-839) on top of the block manager.
 *    Wall-clock inputs (arrival order, the 300 ms swap cooling period) are explicit arguments.
 * ------------------------------------------------------------------------------------------- */
#define MI355_SCHED_SCHEDULED 0       /* group ids of this step                                */
#define MI355_SCHED_IGNORED 1         /* groups whose prompt can never fit (FinishedIgnored)    */
#define MI355_SCHED_SWAP_IN_PAIRS 2   /* (cpu_block, gpu_block) flattened                       */
#define MI355_SCHED_SWAP_OUT_PAIRS 3  /* (gpu_block, cpu_block) flattened                       */
#define MI355_SCHED_COPY_PAIRS 4      /* copy-on-write (src, dst) flattened                     */
#define MI355_SCHED_SWAP_IN_GROUPS 5
#define MI355_SCHED_SWAP_OUT_GROUPS 6
#define MI355_SCHED_RUNNER_RELEASES 7 /* take_pending_runner_releases: sequence ids             */
/* group status codes returned by mi355_sched_group_status */
#define MI355_GROUP_WAITING 0
#define MI355_GROUP_PENDING 1
#define MI355_GROUP_RUNNING 2
#define MI355_GROUP_SWAPPED 3
#define MI355_GROUP_FINISHED 4
#define MI355_GROUP_ABORTED 5
#define MI355_GROUP_IGNORED 6
void* mi355_sched_create(int32_t block_size, int32_t num_gpu_blocks, int32_t num_cpu_blocks, int32_t prefix_cache_enabled,
                         int32_t max_cached_blocks, int32_t max_num_parallel_reqs, int32_t max_num_batched_tokens,
                         int32_t prefill_chunk_size);
void mi355_sched_destroy(void* sched);
void* mi355_sched_block_engine(void* sched);   /* the mi355_be_* handle owned by the scheduler */
int32_t mi355_sched_add_group(void* sched, int64_t group_id, const int64_t* seq_ids, int32_t n, uint64_t arrival);
int32_t mi355_sched_group_status(void* sched, int64_t group_id);
int32_t mi355_sched_set_group_finished(void* sched, int64_t group_id);
int32_t mi355_sched_queue_len(void* sched, int32_t which /* 0 waiting, 1 running, 2 swapped */);
int32_t mi355_sched_has_unfinished(void* sched);
int32_t mi355_sched_is_last_prefill(void* sched);
/* 1 = prompt step, 0 = decode step; read the step's lists with mi355_sched_result */
int32_t mi355_sched_schedule(void* sched, uint64_t now_ms);
int32_t mi355_sched_result(void* sched, int32_t which, int64_t* out, int32_t cap);
int32_t mi355_sched_filter_prefill_finished(void* sched, const int64_t* scheduled, int32_t n, int64_t* finished_out, int32_t cap);
int32_t mi355_sched_free_finished(void* sched, int64_t* released_out, int32_t cap);
int32_t mi355_sched_abort_sequences(void* sched, const int64_t* seq_ids, int32_t n);
void mi355_sched_rollback_swap_in(void* sched, int64_t group_id);
void mi355_sched_rollback_swap_out(void* sched, int64_t group_id);

/* Cache budget: `compute_kvcache_budget_bytes` (src/lib.rs:515-523) and `get_cache_config`
 * (src/lib.rs:128-284, the non-MLA / non-TurboQuant arm).  Host arithmetic only.
 *   budget: round(free_bytes * fraction); -1 for a fraction outside (0, 1].
 *   config: per_block = dsize * block_size * kv_heads_per_shard * head_dim * max(layers, 1) * 2;
 *           num_gpu_blocks = mem_gpu_mb * 2^20 / per_block; num_cpu_blocks = cpu_swap ? (mem_cpu_mb == 0 ? gpu / 2 :
 *           mem_cpu_mb * 2^20 / per_block) : 0; kv_heads_per_shard = heads < shards ? 1 : heads / shards. */
int64_t mi355_kvcache_budget_bytes(int64_t free_bytes, float fraction);
int mi355_get_cache_config(int64_t mem_gpu_mb, int64_t mem_cpu_mb, int32_t block_size, int32_t num_kv_heads, int32_t head_dim,
                           int32_t num_layers, int32_t dsize, int32_t num_shards, int32_t cpu_swap, int64_t* num_gpu_blocks,
                           int64_t* num_cpu_blocks);

/* ---------------------------------------------------------------------------------------------
 * 7. GGUF file reader: src/backend/gguf.rs:48-102,623-712, quantized_var_builder.rs:27-58;
 *    keys and tensor names as GGUFLLaMa reads them (quantized_llama.rs:225-371).  mmap, zero copy.
 * ------------------------------------------------------------------------------------------- */
void* mi355_gguf_open(const char* path);
void mi355_gguf_close(void* gguf);
int32_t mi355_gguf_version(void* gguf);
int32_t mi355_gguf_n_tensors(void* gguf);
int32_t mi355_gguf_get_u64(void* gguf, const char* key, uint64_t* out);
int32_t mi355_gguf_get_f64(void* gguf, const char* key, double* out);
int32_t mi355_gguf_get_str(void* gguf, const char* key, char* out, int32_t cap);
int32_t mi355_gguf_find(void* gguf, const char* name);
int32_t mi355_gguf_tensor_info(void* gguf, int32_t i, char* name, int32_t name_cap, int64_t* dims4, int32_t* n_dims,
                               int32_t* ggml_type, uint64_t* nbytes);
const void* mi355_gguf_tensor_data(void* gguf, int32_t i);
/* Tensor-parallel shard of tensor i as a raw byte range -- `QVarBuilder::get_sharded` over candle's
 * `Content::tensor_shard` (quantized_var_builder.rs:135-183): dim 0 = rows [rank*R/W, (rank+1)*R/W) (contiguous),
 * dim 1 = the same block range of every row.  Host only, no dequantisation.  Returns the shard's bytes (out == NULL
 * sizes the buffer); -1 bad argument / dimension does not divide; -2 a dim-1 shard would cut a quantisation block (the
 * reference's dequantise -> narrow -> re-quantise fallback, :234-269, is mi355_gguf_tensor_shard_q8_0 below); -3 out_cap too small. */
int64_t mi355_gguf_tensor_shard(void* gguf, int32_t i, int32_t dim, int32_t rank, int32_t world, void* out, int64_t out_cap);
/* the fallback of `get_sharded_no_shape` (quantized_var_builder.rs:234-269) for a dim-1 shard that cuts a quantisation block:
 * dequantise to f16, narrow to this rank's columns, re-quantise to Q8_0 (f16 d, 32 x i8 per 34-byte block); the mat-mul of such
 * a tensor goes through the Q8_0 arm of mi355_qmatmul_fused (MI355_GGML_Q8_0, native blocks, no repack).  Q4_K / Q6_K sources */
int64_t mi355_gguf_tensor_shard_q8_0(void* gguf, int32_t tensor, int32_t rank, int32_t world, void* out, int64_t out_cap);
/* rows [row0, row0 + n_rows) of a 2-D tensor in the file's block format; rows beyond the tensor are ZERO blocks (rows of 0.0): the
 * vocab-parallel lm_head shard of a padded vocabulary (distributed.rs:1596-1616).  Byte count (out == NULL: size query), -1 / -3 as above. */
int64_t mi355_gguf_tensor_rows_padded(void* gguf, int32_t tensor, int64_t row0, int64_t n_rows, void* out, int64_t out_cap);
/* GGUFLLaMa::from_gguf (quantized_llama.rs:203-420): config from the metadata, every tensor handed to the model
 * (matrices Q4_K / Q6_K re-tiled, token_embd dequantised on the device, norms F32).  *model_out = mi355_llama handle. */
int mi355_llama_load_gguf(const char* path, int32_t max_batch, int32_t max_blocks_per_seq, int32_t block_size,
                          int32_t kv_layout, int32_t max_seq, void** model_out, mi355_llama_config* cfg_out);
/* The same under tensor parallelism (quantized_llama.rs:316-371, attention.rs:790-848, distributed.rs:1574-1629): rank
 * `tp_rank` of `tp_world` reads the file and keeps its shard -- attn_q / ffn_gate / ffn_up / output rows, attn_k / attn_v
 * rows of `kv_head_shard` (replicated groups when Hkv < W), attn_output / ffn_down k-blocks; token_embd, norms and a
 * Mixtral layer's router + experts whole.  cfg_out holds the GLOBAL dimensions plus tp_rank / tp_world; attach a
 * communicator (mi355_llama_init_comm / mi355_llama_set_comm) before the first step.  An attn_output / ffn_down shard that
 * cuts a k-quant block is re-quantised to Q8_0 (quantized_var_builder.rs:234-269); a vocabulary `pad_vocab_size` pads
 * (distributed.rs:1448-1454) gets zero rows: rank r holds rows [r * local, (r + 1) * local) of the padded matrix and the
 * logits come back narrowed to the real vocabulary (= the unsharded model's). */
int mi355_llama_load_gguf_tp(const char* path, int32_t max_batch, int32_t max_blocks_per_seq, int32_t block_size,
                             int32_t kv_layout, int32_t max_seq, int32_t tp_rank, int32_t tp_world, void** model_out,
                             mi355_llama_config* cfg_out);
/* Host-only dry run of the loader (no device call): reads the metadata, checks every tensor the model needs against it
 * (present, Q4_K / Q6_K or F32 where required, shapes [rows, cols] consistent with head counts / embedding length / feed
 * forward length / vocabulary, tile- and block-aligned, shardable `tp_world` ways) and fills cfg_out with the GLOBAL
 * dimensions.  0 = mi355_llama_load_gguf[_tp] will accept the file; hipErrorInvalidValue (1) = malformed / inconsistent;
 * hipErrorNotSupported (801) = a format or shard this build does not serve (other ggml types, re-quantising shards).
 * The loaders run the same checks before their first allocation. */
int mi355_llama_check_gguf(const char* path, int32_t tp_rank, int32_t tp_world, mi355_llama_config* cfg_out);

/* ABI guard for bindings that mirror the structs by hand (ctypes, a Rust #[repr(C)]): sizeof of
 * 0 = mi355_qmm_desc, 1 = mi355_llama_config, 2 = mi355_dense_config, 3 = mi355_rope_scaling; -1 for an unknown id */
int64_t mi355_abi_struct_size(int32_t which);

#ifdef __cplusplus
}
#endif
#endif /* MI355_VLLM_H */
