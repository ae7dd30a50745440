// Host side of the decode hot path above the kernel C ABI (C++ because the reference's host side is compiled
// Rust and no Rust toolchain exists here).  Mirrors, for the GGUF llama path:
//   GGUFLLaMa::forward_inner            src/openai/models/quantized_llama.rs:424-506   (layer loop, residuals)
//   QuantizedAttention::forward         src/openai/models/layers/attention.rs:910-1011 (qkv, rope, paged attn, wo)
//   Mlp::forward                        src/openai/models/quantized_llama.rs:30-45
//   CacheEngine::allocate_kv_cache      src/scheduler/cache_engine.rs:122-294,298-341
//   LLMEngine::prepare_decode           src/openai/pipelines/inputs.rs:376-454 (slot = table[pos/bs]*bs + pos%bs)
//   decode graph capture / replay       src/backend/graph.rs:471-661,685 ; src/openai/pipelines/pipeline.rs:2091-2135
//   LogitsProcessor::sample_argmax      src/openai/logits_processor.rs:92-95
// Per layer the step is 5 launches (+1 when the context is partitioned):
//   [RMSNorm + wq|wk|wv + RoPE + bf16 + cache scatter] -> [paged attention (+reduce)] -> [wo + residual]
//   -> [RMSNorm + w1|w3 + SiLU*mul] -> [w2 + residual]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <array>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/mi355_vllm.h"
#include "comm.h"
#include "step_inputs.h"

#define HCHECK(expr)                                   \
    do {                                               \
        hipError_t e_ = (expr);                        \
        if (e_ != hipSuccess) return (int)e_;          \
    } while (0)
#define RCHECK(expr)                                   \
    do {                                               \
        int r_ = (expr);                               \
        if (r_ != 0) return r_;                        \
    } while (0)

namespace {

struct QW {
    void* tiles = nullptr;
    int type = 0, n_rows = 0, k = 0;
    bool owned = false;
};

struct Layer {
    float* attn_norm = nullptr;
    float* ffn_norm = nullptr;
    QW w[7];   // MI355_W_WQ .. MI355_W_W3
    // MoE (n_expert > 1): router + one slab of repacked tiles per projection, experts back to back
    float* gate_inp = nullptr;          // f32 [n_expert, hidden]
    uint8_t* eslab[3] = {nullptr, nullptr, nullptr};   // W1, W2, W3
    int etype[3] = {0, 0, 0}, erows[3] = {0, 0, 0}, ek[3] = {0, 0, 0};
    int64_t estride[3] = {0, 0, 0};
};

struct Model {
    mi355_llama_config cfg{};
    std::vector<Layer> layers;
    float* tok_embd = nullptr;      // f32 [vocab, hidden]  (dequantised at load, quantized_llama.rs:262-264)
    float* output_norm = nullptr;
    QW output;
    float *cos_t = nullptr, *sin_t = nullptr;
    // activations
    float* xs = nullptr;            // residual stream f32 [B, hidden]
    uint16_t* q = nullptr;          // bf16 [B, H*D]
    uint16_t* attn = nullptr;       // bf16 [B, H*D]
    float* h = nullptr;             // f32 [B, I]
    float* logits = nullptr;        // f32 [B, vocab]
    float *pa_tmp = nullptr, *pa_max = nullptr, *pa_sum = nullptr;
    int pa_cap_partitions = 0;
    // CacheEngine
    void* kv_slab = nullptr;
    std::vector<void*> kcache, vcache;
    int num_blocks = 0;
    // static step inputs (graph mode)
    uint32_t* d_tokens = nullptr;
    int64_t* d_positions = nullptr;
    int64_t* d_slots = nullptr;
    uint32_t* d_ctx = nullptr;
    uint32_t* d_bt = nullptr;
    int cur_batch = 0, cur_max_blocks = 0, cur_ctx_cap = 0;
    int cur_ctx_max = 0;            // host mirror of max(d_ctx): the device-side loop must stay inside ctx_cap / the block table / max_seq
    // hipGraphs of the greedy step, one per (batch, table width, ctx bucket) -- graph.rs:471-661 keeps one per batch size; a batch
    // that shrinks and grows as requests finish and arrive (continuous batching) finds its graph again instead of re-capturing
    struct StepGraph { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; uint64_t used = 0; };
    std::map<std::array<int, 3>, StepGraph> graphs;
    std::set<std::array<int, 3>> warmed;                  // shapes whose first (EAGER) step ran: kernel attributes / scratch exist
    uint64_t graph_clock = 0;
    int64_t graph_captures = 0, eager_steps = 0;          // mi355_llama_graph_captures: a re-capture inside a timed region must be visible
    bool use_graph = true;
    bool graph_tp = false;          // set_graph(2): capture tensor-parallel steps too (RCCL calls inside the graph; opt-in)
    // tensor parallel
    Comm* comm = nullptr;           // mi355_comm_* handle (owned when created by mi355_llama_init_comm)
    bool comm_owned = false;
    bool use_comm = false;          // tp_world > 1, or forced (single-rank plumbing test: MI355_FORCE_COMM=1)
    float* tp_y = nullptr;          // [B, hidden] this rank's o_proj / down_proj partial in the reference's wire mode (bf16 all-reduce)
    float* p_tp_y = nullptr;        // same for prompt steps (grow-only with the prefill workspace)
    float* logits_local = nullptr;  // [B, vocab/W]
    void* chain_sync = nullptr;     // arrival counters of the chained single-token launches (qmv_chain.inc), zeroed once
    float* logits_gather = nullptr; // [W, B, vocab/W]
    // prefill workspace (grow-only, sized by the largest chunk seen): same roles as xs / q / attn / h
    float* p_xs = nullptr; uint16_t* p_q = nullptr; uint16_t* p_attn = nullptr; float* p_h = nullptr;
    int p_cap = 0;
    // MoE routing scratch (sized for the decode batch; prompt steps use their own, below)
    int32_t* moe_ids = nullptr; float* moe_w = nullptr; float* moe_y = nullptr;
    int32_t* p_moe_ids = nullptr; float* p_moe_w = nullptr; float* p_moe_y = nullptr;
    // grouped experts on prompt steps: gathered rows in expert order + the permutation and its inverse
    float* p_moe_xg = nullptr; int32_t* p_moe_perm = nullptr; int32_t* p_moe_inv = nullptr;
    int32_t* p_moe_tab = nullptr;    // block table of the device-grouped prompt path: {expert, row end} per 64-row block
    // decode steps with many (token, slot) pairs: experts grouped on the device, `g_cap` rows per expert (host_model.cpp run_part)
    float* g_moe_xg = nullptr; float* g_moe_h = nullptr; float* g_moe_yg = nullptr; int32_t* g_moe_pos = nullptr; int32_t* g_moe_cnt = nullptr; int g_cap = 0;
    int attn_numerics = 0;          // parity mode (mi355_llama_set_attention_numerics): 1 = the reference CPU path's bf16 rounding points in decode attention
    bool moe_grouped_done = false;  // this layer's MlpOrMoe ran grouped inside the gate/up part: the down part has nothing left to do
};

// per-pair expert mat-vecs read an expert once per pair; grouped, every expert is read once per 32-row chunk of its `pairs` rows
int g_moe_group = 1;            // tuning key 41: 0 = decode steps never group, 2 = grouped with one launch group per expert (A/B)
// (round 5: an expert receives at most ONE row per token -- a token's n_expert_used experts are distinct, layers/moe.rs top-k -- so its block
// has `tokens` live rows at most and is walked in ceil(tokens / 32) chunks, not ceil(pairs / 32): at batch 32 the second chunk of every
// expert was a launch group that could never be live -- 16 of the 34 GEMM launches of a Mixtral layer, each a no-op with its boundary)
inline bool moe_group_pays(int pairs, int n_expert, int tokens) { return pairs > n_expert * ((tokens + 31) / 32); }
extern "C" int mi355_internal_paged_attention_v2_partials(void* out, float* exp_sums, float* max_logits, float* tmp_out, const void* q,
                                                          const void* key_cache, const void* value_cache, const uint32_t* block_tables,
                                                          const uint32_t* context_lens, int32_t num_seqs, int32_t num_heads,
                                                          int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                                                          int32_t max_blocks_per_seq, int32_t max_context_len, int32_t partition_size,
                                                          float scale, float softcap, int32_t layout, int32_t dtype, int64_t stream,
                                                          int32_t* w_out);                                                   // paged_attention.hip
extern "C" int mi355_internal_pa_stream_reduce(void* out, const float* tmp_out, const float* max_logits, const float* exp_sums,
                                               const uint32_t* context_lens, int32_t B, int32_t H, int32_t W, int32_t slots, int64_t stream);
extern "C" int mi355_internal_paged_attention_fp8_partials(void* out, float* exp_sums, float* max_logits, float* tmp_out, const void* q,
                                                           const void* key_cache, const void* value_cache, const uint32_t* block_tables,
                                                           const uint32_t* context_lens, int32_t num_seqs, int32_t num_heads,
                                                           int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                                                           int32_t max_blocks_per_seq, int32_t max_context_len, int32_t partition_size,
                                                           float scale, float softcap, float k_scale, float v_scale, int64_t stream, int32_t* w_out);
extern "C" int mi355_internal_moe_stage_grouped(const float* xs, const int32_t* ids, int32_t pairs, int32_t top_k, int32_t n_expert, int32_t cap,
                                                int32_t r0, int32_t rows, int32_t hidden, const float* norm_w, int32_t* pos_out,
                                                int32_t* counts_out, const float* x_key, int32_t row_limit, int64_t stream);
extern "C" int mi355_internal_moe_group_limited(int32_t* pos, int32_t* counts, const int32_t* expert_ids, int32_t num_pairs, int32_t n_expert,
                                                int32_t cap, int32_t row_limit, int64_t stream);                       // moe.hip             // qmatmul.hip
extern "C" int mi355_internal_moe_scatter_combine_to_image(float* ys, const float* y_rows, const float* weights, const int32_t* inv, int32_t num_tokens,
                                                           int32_t hidden, int32_t top_k, const float* next_norm_w, int64_t stream);   // qmatmul.hip
extern "C" int mi355_internal_pa_stream_reduce_to_image(void* out, const float* tmp_out, const float* max_logits, const float* exp_sums,
                                                        const uint32_t* context_lens, int32_t B, int32_t H, int32_t W, int32_t slots,
                                                        int64_t stream);                                                     // qmatmul.hip
extern "C" int mi355_pa_stream_auto(int32_t num_seqs, int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, int32_t block_size);   // paged_attention.hip
int g_host_ps_override = 0;     // experiments: mi355_set_tuning(5, partition_size)
// MI355_MOE_PROMPT_DEVICE=0: prompt steps of a mixture-of-experts model sort their (token, slot) pairs on the host again (rounds 2-5; A/B runs)
static const int g_moe_prompt_device = []() { const char* e = getenv("MI355_MOE_PROMPT_DEVICE"); return e && e[0] == '0' ? 0 : 1; }();

int local_heads(const Model* m) { return m->cfg.n_heads / (m->cfg.tp_world > 0 ? m->cfg.tp_world : 1); }
int local_kv_heads(const Model* m) {
    const int w = m->cfg.tp_world > 0 ? m->cfg.tp_world : 1;
    const int l = m->cfg.n_kv_heads / w;
    return l > 0 ? l : 1;   // kv_head_shard: replicated when Hkv < W (distributed.rs:725-765)
}

// v2 partition = what ONE wave streams (32 tokens per wave-chunk for head_dim 128): aim at ~8 waves per CU
int choose_partition(int batch, int kv_heads, int ctx_cap) {
    if (ctx_cap <= 256) return 0;
    int per_seq = (2048 + batch * kv_heads - 1) / (batch * kv_heads);
    if (per_seq < 1) per_seq = 1;
    int ps = (ctx_cap + per_seq - 1) / per_seq;
    ps = ((ps + 31) / 32) * 32;
    if (ps < 32) ps = 32;
    return ps < ctx_cap ? ps : 0;
}

struct StepIn {
    const uint32_t* tokens; const int64_t* positions; const int64_t* slots; const uint32_t* bt; const uint32_t* ctx;
    int B, max_blocks, ctx_cap;
    // activation buffers of this step (decode: the model's static ones; prefill: the T-row workspace)
    float* xs; uint16_t* q; uint16_t* attn; float* h;
    int32_t* moe_ids; float* moe_w; float* moe_y;     // MoE scratch matching the buffers above
    float* tp_y;                                      // [B, hidden] partial of o_proj / down_proj (reference wire mode)
    // prefill only (is_prefill): cu_seqlens_q [num_seqs+1], B = total tokens
    bool is_prefill; const uint32_t* cu_q; int num_seqs, max_seqlen_q;
};

// xs rows of the last token of every sequence -> [num_seqs, hidden]   (quantized_llama.rs:495-499)
__global__ void select_last_rows_kernel(float* dst, const float* src, const uint32_t* cu_q, int hidden) {
    const size_t row = (size_t)cu_q[blockIdx.x + 1] - 1;
    for (int i = threadIdx.x; i < hidden; i += blockDim.x) dst[(size_t)blockIdx.x * hidden + i] = src[row * hidden + i];
}

// [W, B, Vl] -> [B, V]   (VocabParallelLinear: all-gather, un-interleave, narrow to the real vocabulary V <= W * Vl -- the columns
// beyond it are the zero rows of a padded vocabulary, distributed.rs:1637-1663)
__global__ void gather_transpose_kernel(float* out, const float* in, int W, int B, int Vl, int V) {
    const int64_t n = (int64_t)W * B * Vl;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % Vl), b = (int)((i / Vl) % B), w = (int)(i / ((int64_t)Vl * B));
        const int64_t col = (int64_t)w * Vl + v;
        if (col < V) out[(int64_t)b * V + col] = in[i];
    }
}
// distributed.rs:1446-1452
inline int64_t pad_vocab(int64_t vocab, int world) {
    const int64_t p64 = (vocab + 63) / 64 * 64;
    return ((p64 + world - 1) / world * world + 63) / 64 * 64;
}

// C1/C2: all-reduce(sum) of [B, hidden] after o_proj / down_proj (distributed.rs:696-711).
//   wire 0 (default): the f32 stream goes on the wire; rank 0's mat-mul epilogue already added the residual, so the sum is
//                     the new stream (one rounding fewer than the reference; decode messages are latency-bound: 16 KiB at B=1)
//   wire 1          : the reference's numerics -- every rank's partial `y` is rounded to bf16, summed in bf16, and the
//                     result is added to the f32 residual (attention.rs:1003-1008, quantized_llama.rs:38-42)
bool wire_bf16(const Model* m) { return m->use_comm && m->comm && m->comm->wire_bf16; }
int all_reduce_xs(Model* m, float* xs, float* y, int B, int64_t st) {
    if (!m->use_comm) return 0;
    if (!m->comm) return (int)hipErrorNotInitialized;
    const int64_t n = (int64_t)B * m->cfg.hidden;
    if (m->comm->wire_bf16) return y ? comm_all_reduce_f32(m->comm, y, xs, n, st) : (int)hipErrorInvalidValue;
    return comm_all_reduce_f32(m->comm, xs, nullptr, n, st);
}

enum { PART_QKV = 0, PART_ATTN = 1, PART_WO = 2, PART_GATEUP = 3, PART_DOWN = 4, PART_HEAD = 5, PART_EMBED = 6 };

// The mat-mul descriptor of one dense (non-MoE) launch group, exactly as run_part issues it -- shared with the chained
// single-token step (run_chain), which hands four of them to ONE persistent launch.
void dense_desc(Model* m, int l, int part, const StepIn& in, float* logits, mi355_qmm_desc& d) {
    const mi355_llama_config& c = m->cfg;
    const int H = local_heads(m), Hkv = local_kv_heads(m), D = c.head_dim, hid = c.hidden, B = in.B;
    const bool lead = c.tp_rank == 0;
    memset(&d, 0, sizeof(d));
    if (part == PART_HEAD) {
        d.nseg = 1;
        d.w_tiles[0] = m->output.tiles; d.ggml_type[0] = m->output.type; d.n_rows[0] = m->output.n_rows;
        d.x = in.xs; d.x_dtype = MI355_DTYPE_F32; d.ldx = hid; d.k = hid; d.num_tokens = B;
        d.norm_weight = m->output_norm; d.norm_eps = c.rms_eps;
        d.epilogue = MI355_EPI_STORE; d.ldo = m->output.n_rows;
        d.out = m->use_comm ? m->logits_local : logits;
        return;
    }
    Layer& L = m->layers[l];
    const int I = L.w[MI355_W_W1].n_rows;
    if (part == PART_QKV) {
        d.nseg = 3;
        const int qkv[3] = {MI355_W_WQ, MI355_W_WK, MI355_W_WV};
        for (int s = 0; s < 3; ++s) {
            d.w_tiles[s] = L.w[qkv[s]].tiles; d.ggml_type[s] = L.w[qkv[s]].type; d.n_rows[s] = L.w[qkv[s]].n_rows;
        }
        d.x = in.xs; d.x_dtype = MI355_DTYPE_F32; d.ldx = hid; d.k = hid; d.num_tokens = B;
        d.norm_weight = L.attn_norm; d.norm_eps = c.rms_eps;
        d.epilogue = MI355_EPI_QKV_ROPE_CACHE;
        d.cos_table = m->cos_t; d.sin_table = m->sin_t; d.positions = in.positions; d.slot_mapping = in.slots;
        d.q_out = in.q; d.key_cache = m->kcache[l]; d.value_cache = m->vcache[l];
        d.num_heads = H; d.num_kv_heads = Hkv; d.head_dim = D; d.rotary_dim = D;
        d.block_size = c.block_size; d.kv_layout = c.kv_layout;
    } else if (part == PART_WO) {
        d.nseg = 1;
        d.w_tiles[0] = L.w[MI355_W_WO].tiles; d.ggml_type[0] = L.w[MI355_W_WO].type; d.n_rows[0] = L.w[MI355_W_WO].n_rows;
        d.x = in.attn; d.x_dtype = MI355_DTYPE_BF16; d.ldx = H * D; d.k = H * D; d.num_tokens = B;
        d.epilogue = lead ? MI355_EPI_RESID : MI355_EPI_STORE; d.out = in.xs; d.ldo = hid; d.residual = in.xs;
        if (wire_bf16(m)) { d.epilogue = MI355_EPI_STORE; d.out = in.tp_y; d.residual = nullptr; }   // partial only; + residual after the sum
        // 9..32 tokens: the epilogue stages the gate/up launch's activation image (xs is final here unless TP reduces it)
        if (!m->use_comm && c.n_expert <= 1) { d.chain_next = 1; d.chain_next_k = hid; d.chain_next_norm = L.ffn_norm; }
    } else if (part == PART_GATEUP) {
        d.nseg = 2;
        d.w_tiles[0] = L.w[MI355_W_W1].tiles; d.ggml_type[0] = L.w[MI355_W_W1].type; d.n_rows[0] = L.w[MI355_W_W1].n_rows;
        d.w_tiles[1] = L.w[MI355_W_W3].tiles; d.ggml_type[1] = L.w[MI355_W_W3].type; d.n_rows[1] = L.w[MI355_W_W3].n_rows;
        d.x = in.xs; d.x_dtype = MI355_DTYPE_F32; d.ldx = hid; d.k = hid; d.num_tokens = B;
        d.norm_weight = L.ffn_norm; d.norm_eps = c.rms_eps;
        d.epilogue = MI355_EPI_SILU_MUL; d.out = in.h; d.ldo = I;
        d.chain_next = 1; d.chain_next_k = I; d.chain_next_norm = nullptr;          // -> w2
    } else if (part == PART_DOWN) {
        d.nseg = 1;
        d.w_tiles[0] = L.w[MI355_W_W2].tiles; d.ggml_type[0] = L.w[MI355_W_W2].type; d.n_rows[0] = L.w[MI355_W_W2].n_rows;
        d.x = in.h; d.x_dtype = MI355_DTYPE_F32; d.ldx = I; d.k = I; d.num_tokens = B;
        d.epilogue = lead ? MI355_EPI_RESID : MI355_EPI_STORE; d.out = in.xs; d.ldo = hid; d.residual = in.xs;
        if (wire_bf16(m)) { d.epilogue = MI355_EPI_STORE; d.out = in.tp_y; d.residual = nullptr; }
        if (!m->use_comm) {                                         // -> next layer's QKV, or the lm_head
            d.chain_next = 1; d.chain_next_k = hid;
            d.chain_next_norm = (l + 1 < c.n_layers) ? m->layers[l + 1].attn_norm : m->output_norm;
        }
    }
}

// one launch group of the step; `l` = layer (ignored for EMBED / HEAD)
int run_part(Model* m, int l, int part, const StepIn& in, float* logits, int64_t st) {
    const mi355_llama_config& c = m->cfg;
    const int H = local_heads(m), Hkv = local_kv_heads(m), D = c.head_dim, hid = c.hidden, B = in.B;
    const bool lead = c.tp_rank == 0;                              // the rank that carries the residual into the sum
    mi355_qmm_desc d;
    memset(&d, 0, sizeof(d));
    if (part == PART_EMBED) return mi355_embedding_f32(in.xs, m->tok_embd, in.tokens, B, hid, st);
    if (part == PART_HEAD) {
        // --- output_norm + lm_head -> logits f32       (quantized_llama.rs:500-505; vocab-parallel under TP)
        dense_desc(m, 0, PART_HEAD, in, logits, d);
        RCHECK(mi355_qmatmul_fused(&d, st));
        if (!m->use_comm) return 0;
        if (!m->comm) return (int)hipErrorNotInitialized;
        // every rank holds pad_vocab / world rows (zero rows beyond the real vocabulary): the gathered row is narrowed to c.vocab
        if ((int64_t)m->output.n_rows * c.tp_world < c.vocab || (int64_t)m->output.n_rows * c.tp_world > pad_vocab(c.vocab, c.tp_world))
            return (int)hipErrorInvalidValue;
        const size_t cnt = (size_t)B * m->output.n_rows;
        RCHECK(comm_all_gather(m->comm, m->logits_local, m->logits_gather, (int64_t)cnt, MI355_DTYPE_F32, st));
        hipLaunchKernelGGL(gather_transpose_kernel, dim3(512), dim3(256), 0, reinterpret_cast<hipStream_t>(st), logits,
                           m->logits_gather, c.tp_world, B, m->output.n_rows, c.vocab);
        return (int)hipGetLastError();
    }
    Layer& L = m->layers[l];
    const bool moe = c.n_expert > 1;
    const int I = moe ? L.erows[0] : L.w[MI355_W_W1].n_rows;
    if (moe && (part == PART_GATEUP || part == PART_DOWN)) {
        // --- MlpOrMoe::forward (quantized_llama.rs:56-123) without the host round trip
        const int K = c.n_expert_used, pairs = B * K;
        if (!L.gate_inp || !L.eslab[0] || !L.eslab[1] || !L.eslab[2]) return (int)hipErrorInvalidValue;
        if (part == PART_DOWN && m->moe_grouped_done) { m->moe_grouped_done = false; return 0; }
        if (part == PART_GATEUP && in.is_prefill && B >= 16 && m->p_moe_xg) {
            // ---- grouped experts (prompt steps): route on the device, sort the (token, slot) pairs by expert on the host --
            // the reference pulls the routing to the host as well (`to_vec2`, quantized_llama.rs:70) -- then every selected
            // expert streams ONCE over all of its tokens (gate/up with the fused norm + SiLU*mul, down), and one kernel adds the
            // weighted expert outputs to the residual (quantized_llama.rs:93-119, 470).  The per-pair path below reads an
            // expert once per pair: exact, but a 2048-token prompt would stream each expert hundreds of times.
            RCHECK(mi355_moe_route(in.moe_ids, in.moe_w, in.xs, L.ffn_norm, c.rms_eps, L.gate_inp, B, hid, c.n_expert, K, st));
            if (g_moe_prompt_device && m->p_moe_tab && c.n_expert <= 16 && pairs >= 96 && L.etype[0] == MI355_GGML_Q4_K &&
                (L.etype[1] == MI355_GGML_Q4_K || L.etype[1] == MI355_GGML_Q6_K) && L.etype[2] == MI355_GGML_Q4_K) {
                // ---- the same, grouped ON THE DEVICE (round 6): no copy of the routing to the host, no stream synchronisation, and ONE launch
                // per kernel for all experts.  Every expert owns whole 64-row blocks of the gathered buffers; the prompt GEMM walks a block
                // table {expert, row end} (mi355_qmm_desc.group_block_table); rows between an expert's end and its last block's end are
                // computed and never stored.  (Q4_K gate / up and a Q4_K or Q6_K down projection -- a Q4_K_M file; anything else takes the host-sorted path.)
                const int nblk = (pairs + 63) / 64 + c.n_expert, rows = 64 * nblk;
                RCHECK(mi355_moe_group_blocks(m->p_moe_inv, m->p_moe_tab, in.moe_ids, pairs, c.n_expert, nblk + 2, st));
                RCHECK(mi355_moe_gather_pos(m->p_moe_xg, in.xs, m->p_moe_inv, pairs, K, hid, st));
                mi355_qmm_desc g;
                memset(&g, 0, sizeof(g));
                g.nseg = 2;
                g.w_tiles[0] = L.eslab[0]; g.ggml_type[0] = L.etype[0]; g.n_rows[0] = L.erows[0];
                g.w_tiles[1] = L.eslab[2]; g.ggml_type[1] = L.etype[2]; g.n_rows[1] = L.erows[2];
                g.moe_expert_stride[0] = L.estride[0]; g.moe_expert_stride[1] = L.estride[2];
                g.x = m->p_moe_xg; g.x_dtype = MI355_DTYPE_F32; g.ldx = hid; g.k = hid; g.num_tokens = rows;
                g.norm_weight = L.ffn_norm; g.norm_eps = c.rms_eps;
                g.epilogue = MI355_EPI_SILU_MUL; g.out = in.h; g.ldo = I;
                g.group_block_table = m->p_moe_tab;
                RCHECK(mi355_qmatmul_fused(&g, st));
                mi355_qmm_desc dn;
                memset(&dn, 0, sizeof(dn));
                dn.nseg = 1;
                dn.w_tiles[0] = L.eslab[1]; dn.ggml_type[0] = L.etype[1]; dn.n_rows[0] = L.erows[1];
                dn.moe_expert_stride[0] = L.estride[1];
                dn.x = in.h; dn.x_dtype = MI355_DTYPE_F32; dn.ldx = I; dn.k = I; dn.num_tokens = rows;
                dn.epilogue = MI355_EPI_STORE; dn.out = in.moe_y; dn.ldo = hid;
                dn.group_block_table = m->p_moe_tab;
                RCHECK(mi355_qmatmul_fused(&dn, st));
                RCHECK(mi355_moe_scatter_combine(in.xs, in.moe_y, in.moe_w, m->p_moe_inv, B, hid, K, st));
                m->moe_grouped_done = true;
                return 0;
            }
            std::vector<int32_t> ids((size_t)pairs), perm((size_t)pairs), inv((size_t)pairs);
            HCHECK(hipMemcpyAsync(ids.data(), in.moe_ids, (size_t)pairs * 4, hipMemcpyDeviceToHost, reinterpret_cast<hipStream_t>(st)));
            HCHECK(hipStreamSynchronize(reinterpret_cast<hipStream_t>(st)));
            std::vector<int> count(c.n_expert + 1, 0);
            for (int p = 0; p < pairs; ++p) {
                if (ids[p] < 0 || ids[p] >= c.n_expert) return (int)hipErrorInvalidValue;
                ++count[ids[p] + 1];
            }
            for (int e = 0; e < c.n_expert; ++e) count[e + 1] += count[e];           // offsets
            std::vector<int> fill(count.begin(), count.end() - 1);
            for (int p = 0; p < pairs; ++p) { const int pos = fill[ids[p]]++; perm[pos] = p; inv[p] = pos; }   // stable: token order inside an expert
            HCHECK(hipMemcpyAsync(m->p_moe_perm, perm.data(), (size_t)pairs * 4, hipMemcpyHostToDevice, reinterpret_cast<hipStream_t>(st)));
            HCHECK(hipMemcpyAsync(m->p_moe_inv, inv.data(), (size_t)pairs * 4, hipMemcpyHostToDevice, reinterpret_cast<hipStream_t>(st)));
            RCHECK(mi355_moe_gather(m->p_moe_xg, in.xs, m->p_moe_perm, pairs, K, hid, st));
            for (int e = 0; e < c.n_expert; ++e) {
                const int off = count[e], n_e = count[e + 1] - count[e];
                if (n_e == 0) continue;
                mi355_qmm_desc g;
                memset(&g, 0, sizeof(g));
                g.nseg = 2;
                g.w_tiles[0] = L.eslab[0] + (size_t)e * L.estride[0]; g.ggml_type[0] = L.etype[0]; g.n_rows[0] = L.erows[0];
                g.w_tiles[1] = L.eslab[2] + (size_t)e * L.estride[2]; g.ggml_type[1] = L.etype[2]; g.n_rows[1] = L.erows[2];
                g.x = m->p_moe_xg + (size_t)off * hid; g.x_dtype = MI355_DTYPE_F32; g.ldx = hid; g.k = hid; g.num_tokens = n_e;
                g.norm_weight = L.ffn_norm; g.norm_eps = c.rms_eps;
                g.epilogue = MI355_EPI_SILU_MUL; g.out = in.h + (size_t)off * I; g.ldo = I;
                RCHECK(mi355_qmatmul_fused(&g, st));
                mi355_qmm_desc dn;
                memset(&dn, 0, sizeof(dn));
                dn.nseg = 1;
                dn.w_tiles[0] = L.eslab[1] + (size_t)e * L.estride[1]; dn.ggml_type[0] = L.etype[1]; dn.n_rows[0] = L.erows[1];
                dn.x = in.h + (size_t)off * I; dn.x_dtype = MI355_DTYPE_F32; dn.ldx = I; dn.k = I; dn.num_tokens = n_e;
                dn.epilogue = MI355_EPI_STORE; dn.out = in.moe_y + (size_t)off * hid; dn.ldo = hid;
                RCHECK(mi355_qmatmul_fused(&dn, st));
            }
            RCHECK(mi355_moe_scatter_combine(in.xs, in.moe_y, in.moe_w, m->p_moe_inv, B, hid, K, st));
            HCHECK(hipStreamSynchronize(reinterpret_cast<hipStream_t>(st)));          // perm / inv leave scope with this call
            m->moe_grouped_done = true;
            return 0;
        }
        if (part == PART_GATEUP && !in.is_prefill && g_moe_group && m->g_moe_xg && pairs <= m->g_cap && moe_group_pays(pairs, c.n_expert, B)) {
            // ---- grouped experts, decided on the device (decode steps with many pairs; graph-safe: no host round trip, fixed launch
            // shapes): route, give every pair a row in its expert's block of `cap` rows (stable), gather, then EVERY expert streams
            // once per 32-row chunk over its whole block (rows past its count hold stale finite values that nobody reads back), and
            // one kernel adds the weighted rows to the residual.  Per pair (below) a batch-32 Mixtral step reads 64 experts per
            // layer; here 8 experts x 2 chunks.
            RCHECK(mi355_moe_route(in.moe_ids, in.moe_w, in.xs, L.ffn_norm, c.rms_eps, L.gate_inp, B, hid, c.n_expert, K, st));
            // grouping, row gather and image staging as ONE launch per chunk where the grouped call takes the one-launch path (key 41 = 1);
            // otherwise (-4) the pairs are grouped and their rows gathered first
            bool staged = false;
            // the chunk loop below computes rows [0, row_limit) of every expert's block: an expert holds at most one row per token (a token's
            // experts are distinct); a rank beyond that -- a corrupted router output -- lands on the NaN dump row and the step fails loudly
            const int row_limit = (B + 31) / 32 * 32;
            if (g_moe_group == 1) {
                const int rs = mi355_internal_moe_stage_grouped(in.xs, in.moe_ids, pairs, K, c.n_expert, m->g_cap, 0, 32, hid, L.ffn_norm, m->g_moe_pos,
                                                                m->g_moe_cnt, m->g_moe_xg, row_limit, st);
                if (rs != 0 && rs != -4) return rs;
                staged = rs == 0;
            }
            if (!staged) {
                RCHECK(mi355_internal_moe_group_limited(m->g_moe_pos, m->g_moe_cnt, in.moe_ids, pairs, c.n_expert, m->g_cap, row_limit, st));
                RCHECK(mi355_moe_gather_pos(m->g_moe_xg, in.xs, m->g_moe_pos, pairs, K, hid, st));
            }
            // key 41 = 1: ALL experts in the z extent of one launch each (mi355_qmm_desc.group_count): 5 launches per chunk instead of 5 per
            // (expert, chunk); = 2: one launch group per expert (A/B).  A chunk no pair landed in (the expert's count, on the device, <= its
            // first row) falls through every kernel
            const int n_grp = g_moe_group == 2 ? 1 : c.n_expert;
            for (int e = 0; e < c.n_expert; e += n_grp) {
                for (int r0 = 0; r0 < B; r0 += 32) {                 // an expert holds at most one row per token
                    const int rows = 32;                              // (the gate is built into the 9..32-token launches)
                    const size_t off = (size_t)e * m->g_cap + r0;
                    if (staged && r0 > 0)                             // the next chunk's images (chunk 0's were staged above)
                        RCHECK(mi355_internal_moe_stage_grouped(in.xs, in.moe_ids, pairs, K, c.n_expert, m->g_cap, r0, 32, hid, L.ffn_norm,
                                                                m->g_moe_pos, m->g_moe_cnt, m->g_moe_xg + (size_t)r0 * hid, row_limit, st));
                    mi355_qmm_desc g;
                    memset(&g, 0, sizeof(g));
                    g.nseg = 2;
                    g.w_tiles[0] = L.eslab[0] + (size_t)e * L.estride[0]; g.ggml_type[0] = L.etype[0]; g.n_rows[0] = L.erows[0];
                    g.w_tiles[1] = L.eslab[2] + (size_t)e * L.estride[2]; g.ggml_type[1] = L.etype[2]; g.n_rows[1] = L.erows[2];
                    g.x = m->g_moe_xg + off * hid; g.x_dtype = MI355_DTYPE_F32; g.ldx = hid; g.k = hid; g.num_tokens = rows;
                    g.norm_weight = L.ffn_norm; g.norm_eps = c.rms_eps;
                    g.epilogue = MI355_EPI_SILU_MUL; g.out = m->g_moe_h + off * I; g.ldo = I;
                    g.rows_dev = m->g_moe_cnt + e; g.rows_min = r0;
                    g.chain_next = 1; g.chain_next_k = I; g.chain_next_norm = nullptr;   // its epilogue stages the down launch's image (same gate)
                    if (n_grp > 1) {
                        g.group_count = n_grp; g.group_x_stride = (int64_t)m->g_cap * hid; g.group_out_stride = (int64_t)m->g_cap * I;
                        g.moe_expert_stride[0] = L.estride[0]; g.moe_expert_stride[1] = L.estride[2];
                    }
                    RCHECK(mi355_qmatmul_fused(&g, st));
                    mi355_qmm_desc dn;
                    memset(&dn, 0, sizeof(dn));
                    dn.nseg = 1;
                    dn.w_tiles[0] = L.eslab[1] + (size_t)e * L.estride[1]; dn.ggml_type[0] = L.etype[1]; dn.n_rows[0] = L.erows[1];
                    dn.x = m->g_moe_h + off * I; dn.x_dtype = MI355_DTYPE_F32; dn.ldx = I; dn.k = I; dn.num_tokens = rows;
                    dn.epilogue = MI355_EPI_STORE; dn.out = m->g_moe_yg + off * hid; dn.ldo = hid;
                    dn.rows_dev = m->g_moe_cnt + e; dn.rows_min = r0;
                    if (n_grp > 1) {
                        dn.group_count = n_grp; dn.group_x_stride = (int64_t)m->g_cap * I; dn.group_out_stride = (int64_t)m->g_cap * hid;
                        dn.moe_expert_stride[0] = L.estride[1];
                    }
                    RCHECK(mi355_qmatmul_fused(&dn, st));
                }
            }
            {   // + residual; where the next mat-mul reads xs on the 9..32-token path the same launch stages its activation image
                int rs = -4;
                if (!m->use_comm && g_moe_group == 1)
                    rs = mi355_internal_moe_scatter_combine_to_image(in.xs, m->g_moe_yg, in.moe_w, m->g_moe_pos, B, hid, K,
                                                                     (l + 1 < c.n_layers) ? m->layers[l + 1].attn_norm : m->output_norm, st);
                if (rs == -4) rs = mi355_moe_scatter_combine(in.xs, m->g_moe_yg, in.moe_w, m->g_moe_pos, B, hid, K, st);
                if (rs) return rs;
            }
            m->moe_grouped_done = true;
            return 0;
        }
        if (part == PART_GATEUP) {
            RCHECK(mi355_moe_route(in.moe_ids, in.moe_w, in.xs, L.ffn_norm, c.rms_eps, L.gate_inp, B, hid, c.n_expert, K, st));
            d.nseg = 2;
            d.w_tiles[0] = L.eslab[0]; d.ggml_type[0] = L.etype[0]; d.n_rows[0] = L.erows[0];
            d.w_tiles[1] = L.eslab[2]; d.ggml_type[1] = L.etype[2]; d.n_rows[1] = L.erows[2];
            d.x = in.xs; d.x_dtype = MI355_DTYPE_F32; d.ldx = hid; d.k = hid; d.num_tokens = 1;
            d.norm_weight = L.ffn_norm; d.norm_eps = c.rms_eps;
            d.epilogue = MI355_EPI_SILU_MUL; d.out = in.h; d.ldo = I;
            d.moe_expert_ids = in.moe_ids; d.moe_pairs = pairs; d.moe_x_div = K;
            d.moe_expert_stride[0] = L.estride[0]; d.moe_expert_stride[1] = L.estride[2];
            return mi355_qmatmul_fused(&d, st);
        }
        d.nseg = 1;
        d.w_tiles[0] = L.eslab[1]; d.ggml_type[0] = L.etype[1]; d.n_rows[0] = L.erows[1];
        d.x = in.h; d.x_dtype = MI355_DTYPE_F32; d.ldx = I; d.k = I; d.num_tokens = 1;
        d.epilogue = MI355_EPI_STORE; d.out = in.moe_y; d.ldo = hid;
        d.moe_expert_ids = in.moe_ids; d.moe_pairs = pairs; d.moe_x_div = 1;
        d.moe_expert_stride[0] = L.estride[1];
        RCHECK(mi355_qmatmul_fused(&d, st));
        // ys.index_add(weighted expert outputs) + residual (quantized_llama.rs:112-113, 470)
        return mi355_moe_combine(in.xs, in.moe_y, in.moe_w, B, hid, K, 1, st);
    }
    if (part == PART_QKV) {
        // --- attention_norm + wq|wk|wv + interleaved RoPE + bf16 cast + cache scatter
        dense_desc(m, l, PART_QKV, in, logits, d);
        return mi355_qmatmul_fused(&d, st);
    }
    if (part == PART_ATTN && c.kv_layout == MI355_KV_PAGED_FP8) {
        // `--kvcache-dtype fp8` (is_fp8_keys, attention.rs:896): every key comes back from the e4m3fn cache
        const float scale = 1.0f / sqrtf((float)D);
        if (in.is_prefill)
            return mi355_prefill_attention_fp8(in.attn, in.q, m->kcache[l], m->vcache[l], in.bt, in.ctx, in.cu_q, in.num_seqs,
                                               in.max_seqlen_q, H, Hkv, D, c.block_size, in.max_blocks, scale, 0.f, 1.f, 1.f,
                                               MI355_DTYPE_BF16, st);
        int ps = choose_partition(B, Hkv, in.ctx_cap);
        if (ps > 0) ps = 32;   // MFMA kernel: 32-token partitions, 4 per workgroup (measured at batch 32: 32 -> 4937, 64 -> 4784, 128 -> ~4500 tok/s)
        // >= 64 (sequence, kv head) pairs: the balanced LDS-DMA stream over the e4m3fn cache (round 5), as on the bf16 cache
        if (ps > 0 && mi355_pa_stream_auto(B, H, Hkv, D, c.block_size)) ps = 64;
        if (g_host_ps_override > 0 && ps > 0) ps = g_host_ps_override;
        if (ps > 0 && (in.ctx_cap + ps - 1) / ps > m->pa_cap_partitions) return (int)hipErrorInvalidValue;
        if (in.ctx_cap > c.max_seq) return (int)hipErrorInvalidValue;
        if (ps == 64 && B >= 9 && B <= 32 && !(H & 1) && !m->use_comm) {
            // as on the bf16 cache below: the merge of the stream's partials also stages wo's activation image
            int32_t w = 0;
            RCHECK(mi355_internal_paged_attention_fp8_partials(in.attn, m->pa_sum, m->pa_max, m->pa_tmp, in.q, m->kcache[l], m->vcache[l], in.bt,
                                                               in.ctx, B, H, Hkv, D, c.block_size, in.max_blocks, in.ctx_cap, ps, scale, 0.f, 1.f, 1.f,
                                                               st, &w));
            if (w == 0) return 0;                                     // the launch did not take the stream: it is complete
            const int max_partitions = (in.ctx_cap + ps - 1) / ps < 1 ? 1 : (in.ctx_cap + ps - 1) / ps;
            const int rc = mi355_internal_pa_stream_reduce_to_image(in.attn, m->pa_tmp, m->pa_max, m->pa_sum, in.ctx, B, H, w, max_partitions, st);
            if (rc != -4) return rc;
            return mi355_internal_pa_stream_reduce(in.attn, m->pa_tmp, m->pa_max, m->pa_sum, in.ctx, B, H, w, max_partitions, st);
        }
        return mi355_paged_attention_fp8(in.attn, m->pa_sum, m->pa_max, m->pa_tmp, in.q, m->kcache[l], m->vcache[l], in.bt, in.ctx,
                                         B, H, Hkv, D, c.block_size, in.max_blocks, in.ctx_cap, ps, scale, 0.f, 1.f, 1.f, st);
    }
    if (part == PART_ATTN && in.is_prefill) {
        // --- K4: every key (cached prefix + this chunk, just written by the QKV epilogue) comes from the cache
        return mi355_prefill_attention(in.attn, in.q, nullptr, nullptr, m->kcache[l], m->vcache[l], in.bt, in.ctx,
                                       in.cu_q, in.num_seqs, in.max_seqlen_q, H, Hkv, D, c.block_size, in.max_blocks,
                                       1.0f / sqrtf((float)D), 0.f, c.kv_layout, MI355_DTYPE_BF16, st);
    }
    if (part == PART_ATTN && m->attn_numerics == 1) {
        // parity mode: scores / probabilities / P.V rounded to bf16 where models/mod.rs:1288-1306 rounds them (tests only)
        if (in.ctx_cap > c.max_seq) return (int)hipErrorInvalidValue;
        return mi355_paged_attention_reference_numerics(in.attn, in.q, m->kcache[l], m->vcache[l], in.bt, in.ctx, B, H, Hkv, D,
                                                        c.block_size, in.max_blocks, in.ctx_cap, 1.0f / sqrtf((float)D), c.kv_layout, st);
    }
    if (part == PART_ATTN) {
        // --- paged attention over the cache (the new token's K/V are already in place)
        int ps = choose_partition(B, Hkv, in.ctx_cap);
        if (ps > 0 && c.kv_layout == MI355_KV_PAGED) ps = 32;   // MFMA kernel: 32-token partitions, 4 per workgroup (measured at batch 32: 32 -> 4937, 64 -> 4784, 128 -> ~4500 tok/s)
        // >= 64 (sequence, kv head) pairs: 64-token stages through the balanced LDS-DMA stream (round 4: 5987 -> 6177 tok/s at batch 32)
        if (ps > 0 && c.kv_layout == MI355_KV_PAGED && mi355_pa_stream_auto(B, H, Hkv, D, c.block_size)) ps = 64;
        if (g_host_ps_override > 0 && ps > 0) ps = g_host_ps_override;
        const float scale = 1.0f / sqrtf((float)D);
        if (ps > 0 && (in.ctx_cap + ps - 1) / ps > m->pa_cap_partitions) return (int)hipErrorInvalidValue;
        if (in.ctx_cap > c.max_seq) return (int)hipErrorInvalidValue;
        if (ps == 0)
            return mi355_paged_attention_v1(in.attn, in.q, m->kcache[l], m->vcache[l], in.bt, in.ctx, B, H, Hkv, D,
                                            c.block_size, in.max_blocks, in.ctx_cap, scale, 0.f, c.kv_layout,
                                            MI355_DTYPE_BF16, st);
        if (ps == 64 && B >= 9 && B <= 32 && !(H & 1) && c.kv_layout == MI355_KV_PAGED && !m->use_comm) {
            // the balanced stream at a batch whose wo runs on the 9..32-token path: the merge of the partials also stages wo's activation
            // image (qmatmul.hip mi355_internal_pa_stream_reduce_to_image): one launch instead of merge + staging
            int32_t w = 0;
            RCHECK(mi355_internal_paged_attention_v2_partials(in.attn, m->pa_sum, m->pa_max, m->pa_tmp, in.q, m->kcache[l], m->vcache[l], in.bt,
                                                              in.ctx, B, H, Hkv, D, c.block_size, in.max_blocks, in.ctx_cap, ps, scale, 0.f,
                                                              c.kv_layout, MI355_DTYPE_BF16, st, &w));
            if (w == 0) return 0;                                     // the launch did not take the stream: it is complete
            const int max_partitions = (in.ctx_cap + ps - 1) / ps < 1 ? 1 : (in.ctx_cap + ps - 1) / ps;
            const int rc = mi355_internal_pa_stream_reduce_to_image(in.attn, m->pa_tmp, m->pa_max, m->pa_sum, in.ctx, B, H, w, max_partitions, st);
            if (rc != -4) return rc;
            return mi355_internal_pa_stream_reduce(in.attn, m->pa_tmp, m->pa_max, m->pa_sum, in.ctx, B, H, w, max_partitions, st);
        }
        return mi355_paged_attention_v2(in.attn, m->pa_sum, m->pa_max, m->pa_tmp, in.q, m->kcache[l], m->vcache[l],
                                        in.bt, in.ctx, B, H, Hkv, D, c.block_size, in.max_blocks, in.ctx_cap, ps,
                                        scale, 0.f, c.kv_layout, MI355_DTYPE_BF16, st);
    }
    if (part == PART_WO || part == PART_DOWN) {
        // --- wo(y.to_dtype(F32)) + residual (+ all-reduce)   (attention.rs:1004-1009, quantized_llama.rs:464)
        // --- w2 + residual (+ all-reduce)                     (quantized_llama.rs:37-42, 470)
        dense_desc(m, l, part, in, logits, d);
        RCHECK(mi355_qmatmul_fused(&d, st));
        return all_reduce_xs(m, in.xs, in.tp_y, B, st);
    }
    if (part == PART_GATEUP) {
        // --- ffn_norm + w1|w3 + silu*mul              (quantized_llama.rs:33-37, 468)
        dense_desc(m, l, PART_GATEUP, in, logits, d);
        return mi355_qmatmul_fused(&d, st);
    }
    return (int)hipErrorInvalidValue;
}

// Single token, one GPU, dense MLP: the four mat-vecs between two attention calls -- wo + residual, ffn_norm + gate/up +
// silu*mul, down + residual, then the NEXT layer's attn_norm + q|k|v + RoPE + cache write (or output_norm + lm_head after
// the last layer) -- run as ONE persistent launch (csrc/qmv_chain.inc).  Returns hipErrorNotSupported when the chain does
// not apply; the caller then issues the launch groups one by one.
int run_chain(Model* m, int l, const StepIn& in, float* logits, int64_t st) {
    const mi355_llama_config& c = m->cfg;
    if (in.B != 1 || in.is_prefill || m->use_comm || c.n_expert > 1 || !m->chain_sync) return (int)hipErrorNotSupported;
    mi355_qmm_desc d[4];
    dense_desc(m, l, PART_WO, in, logits, d[0]);
    dense_desc(m, l, PART_GATEUP, in, logits, d[1]);
    dense_desc(m, l, PART_DOWN, in, logits, d[2]);
    if (l + 1 < c.n_layers) dense_desc(m, l + 1, PART_QKV, in, logits, d[3]);
    else dense_desc(m, 0, PART_HEAD, in, logits, d[3]);
    return mi355_qmatmul_chain(d, 4, m->chain_sync, st);
}

// one decode step over device-resident inputs (everything enqueued on `st`)
int forward_decode(Model* m, const uint32_t* tokens, const int64_t* positions, const int64_t* slots,
                   const uint32_t* bt, const uint32_t* ctx, int B, int max_blocks, int ctx_cap, float* logits,
                   int64_t st) {
    const mi355_llama_config& c = m->cfg;
    if (B < 1 || B > c.max_batch) return (int)hipErrorInvalidValue;
    if ((int)m->kcache.size() != c.n_layers) return (int)hipErrorInvalidValue;
    const StepIn in{tokens, positions, slots, bt, ctx, B, max_blocks, ctx_cap, m->xs, m->q, m->attn, m->h,
                    m->moe_ids, m->moe_w, m->moe_y, m->tp_y, false, nullptr, 0, 0};
    RCHECK(run_part(m, 0, PART_EMBED, in, logits, st));
    // single token: embed, q|k|v of layer 0, then per layer [attention, ONE persistent launch for wo -> gate/up -> down ->
    // the next layer's q|k|v (the lm_head after the last layer)]; any step the chain does not cover falls back group by group
    bool qkv_done = false, head_done = false;
    for (int l = 0; l < c.n_layers; ++l) {
        if (!qkv_done) RCHECK(run_part(m, l, PART_QKV, in, logits, st));
        RCHECK(run_part(m, l, PART_ATTN, in, logits, st));
        const int rc = run_chain(m, l, in, logits, st);
        if (rc == 0) {
            qkv_done = true;
            head_done = l + 1 == c.n_layers;
        } else if (rc == (int)hipErrorNotSupported) {
            qkv_done = false;
            for (int part = PART_WO; part <= PART_DOWN; ++part) RCHECK(run_part(m, l, part, in, logits, st));
        } else {
            return rc;
        }
    }
    if (head_done) return 0;
    return run_part(m, 0, PART_HEAD, in, logits, st);
}

void free_qw(QW& w) {
    if (w.owned && w.tiles) (void)hipFree(w.tiles);
    w = QW{};
}

int set_qw(QW& dst, int type, const void* tiles_dev, int n_rows, int k, bool owned) {
    free_qw(dst);
    dst.tiles = const_cast<void*>(tiles_dev);
    dst.type = type; dst.n_rows = n_rows; dst.k = k; dst.owned = owned;
    return 0;
}

QW* qw_slot(Model* m, int layer, int which) {
    if (layer < 0) return which == MI355_W_OUTPUT ? &m->output : nullptr;
    if (layer >= m->cfg.n_layers || which < MI355_W_WQ || which > MI355_W_W3) return nullptr;
    return &m->layers[layer].w[which];
}

#define MI355_MAX_STEP_GRAPHS 96
void drop_graph(Model* m) {
    for (auto& kv : m->graphs) {
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    m->graphs.clear();
    m->warmed.clear();                                                // whatever invalidated the graphs may also change which kernels / scratch a step uses: warm again
}

}  // namespace

extern "C" void mi355_host_set_partition_override(int v) { g_host_ps_override = v; }
extern "C" int mi355_host_get_partition_override() { return g_host_ps_override; }   // the 16-bit driver honours the same switch
extern "C" void mi355_host_set_moe_group(int v) { g_moe_group = v; }

extern "C" void* mi355_llama_create(const mi355_llama_config* cfg) {
    if (!cfg || cfg->hidden <= 0 || cfg->n_layers <= 0 || cfg->max_batch <= 0 || cfg->head_dim <= 0 ||
        (cfg->hidden % 256) || cfg->max_blocks_per_seq <= 0 || cfg->max_seq <= 0 || cfg->block_size <= 0 ||
        cfg->n_heads <= 0 || cfg->n_kv_heads <= 0 || cfg->vocab <= 0 || cfg->intermediate <= 0)
        return nullptr;
    // vocab-parallel lm_head: every rank holds pad_vocab / world rows.  pad_vocab_size rounds to 64 AFTER it rounded to the world size
    // (distributed.rs:1446-1452), so for a world that does not divide 64 (3, 6, ...) the padded size may not divide: refused here, before
    // a single weight is loaded, instead of at the first step
    if (cfg->tp_world > 1 && pad_vocab(cfg->vocab, cfg->tp_world) % cfg->tp_world != 0) return nullptr;
    Model* m = nullptr;
    try {
    m = new Model();
    m->cfg = *cfg;
    if (m->cfg.tp_world <= 0) { m->cfg.tp_world = 1; m->cfg.tp_rank = 0; }
    const char* force = getenv("MI355_FORCE_COMM");
    m->use_comm = m->cfg.tp_world > 1 || (force && force[0] == '1');
    m->layers.resize(cfg->n_layers);
    const int B = cfg->max_batch, H = local_heads(m), D = cfg->head_dim;
    bool ok = true;
    auto alloc = [&](void** p, size_t bytes) { if (hipMalloc(p, bytes) != hipSuccess) ok = false; };
    alloc((void**)&m->xs, (size_t)B * cfg->hidden * 4);
    alloc((void**)&m->q, (size_t)B * H * D * 2);
    alloc((void**)&m->attn, (size_t)B * H * D * 2);
    const int KE = cfg->n_expert > 1 ? (cfg->n_expert_used > 0 ? cfg->n_expert_used : 1) : 1;
    // MoE under TP: experts and router are replicated on every rank (quantized_llama.rs:344-365), attention is sharded
    if (cfg->n_expert > 1 && (cfg->n_expert_used < 1 || cfg->n_expert_used > cfg->n_expert)) { delete m; return nullptr; }
    alloc((void**)&m->h, (size_t)B * KE * cfg->intermediate * 4);
    if (cfg->n_expert > 1) {
        alloc((void**)&m->moe_ids, (size_t)B * KE * 4);
        alloc((void**)&m->moe_w, (size_t)B * KE * 4);
        alloc((void**)&m->moe_y, (size_t)B * KE * cfg->hidden * 4);
        if (moe_group_pays(B * KE, cfg->n_expert, B)) {          // device-grouped decode: every expert owns B * KE rows of these
            m->g_cap = (B * KE + 31) / 32 * 32;                // whole 32-row chunks: the launch shape of every (expert, chunk)
            const size_t rows = (size_t)cfg->n_expert * m->g_cap + 1;            // + the dump row of an out-of-range expert id (moe_group_kernel)
            alloc((void**)&m->g_moe_xg, rows * cfg->hidden * 4);
            alloc((void**)&m->g_moe_h, rows * cfg->intermediate * 4);
            alloc((void**)&m->g_moe_yg, rows * cfg->hidden * 4);
            alloc((void**)&m->g_moe_pos, (size_t)m->g_cap * 4);
            alloc((void**)&m->g_moe_cnt, (size_t)cfg->n_expert * 4);
            if (m->g_moe_xg) (void)hipMemset(m->g_moe_xg, 0, rows * cfg->hidden * 4);      // rows no pair lands in: finite from the start
            // the DUMP row of the expert outputs is NaN and no launch ever writes it: a pair that was sent there (expert id out of range, or
            // a rank beyond the rows the chunk loop covers) poisons its token's logits instead of combining a row nobody computed
            if (m->g_moe_yg) (void)hipMemset(m->g_moe_yg + (rows - 1) * cfg->hidden, 0xFF, (size_t)cfg->hidden * 4);
        }
    }
    alloc((void**)&m->logits, (size_t)B * cfg->vocab * 4);
    if (m->use_comm) {
        alloc((void**)&m->tp_y, (size_t)B * cfg->hidden * 4);
        const size_t padded = (size_t)pad_vocab(cfg->vocab, m->cfg.tp_world);              // the lm_head shards may carry zero rows
        alloc((void**)&m->logits_local, (size_t)B * padded * 4 / m->cfg.tp_world + 64);
        alloc((void**)&m->logits_gather, (size_t)B * padded * 4 + 64 * m->cfg.tp_world);
    }
    alloc(&m->chain_sync, (size_t)mi355_qmv_chain_sync_bytes());
    if (m->chain_sync && hipMemset(m->chain_sync, 0, (size_t)mi355_qmv_chain_sync_bytes()) != hipSuccess) ok = false;
    m->pa_cap_partitions = (cfg->max_seq + 31) / 32 + 1;   // v2 partitions are >= 32 tokens (choose_partition)
    alloc((void**)&m->pa_tmp, (size_t)B * H * m->pa_cap_partitions * D * 4);
    alloc((void**)&m->pa_max, (size_t)B * H * m->pa_cap_partitions * 4);
    alloc((void**)&m->pa_sum, (size_t)B * H * m->pa_cap_partitions * 4);
    alloc((void**)&m->d_tokens, (size_t)B * 4);
    alloc((void**)&m->d_positions, (size_t)B * 8);
    alloc((void**)&m->d_slots, (size_t)B * 8);
    alloc((void**)&m->d_ctx, (size_t)B * 4);
    alloc((void**)&m->d_bt, (size_t)B * cfg->max_blocks_per_seq * 4);
    // RoPE tables (rotary_emb.rs:14-48): inv_freq f32, idx_theta = pos(f32) * inv_freq(f32), cos/sin in f32
    const int half = D / 2;
    std::vector<float> ct((size_t)cfg->max_seq * half), stb((size_t)cfg->max_seq * half);
    if (mi355_rope_tables(ct.data(), stb.data(), D, cfg->max_seq, (double)cfg->rope_theta, nullptr, cfg->max_seq, 0) != 0) ok = false;
    alloc((void**)&m->cos_t, ct.size() * 4);
    alloc((void**)&m->sin_t, stb.size() * 4);
    if (ok) {
        ok = hipMemcpy(m->cos_t, ct.data(), ct.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(m->sin_t, stb.data(), stb.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) { mi355_llama_destroy(m); return nullptr; }
    return m;
    } catch (...) {                                          // bad_alloc and friends must not cross the C boundary
        if (m) mi355_llama_destroy(m);
        return nullptr;
    }
}

extern "C" int mi355_llama_set_rope_tables(void* mp, const float* cos_host, const float* sin_host, int32_t n_positions) {
    Model* m = static_cast<Model*>(mp);
    if (!m || !cos_host || !sin_host || n_positions < m->cfg.max_seq) return (int)hipErrorInvalidValue;
    const size_t bytes = (size_t)n_positions * (m->cfg.head_dim / 2) * 4;
    float *c = nullptr, *s = nullptr;
    HCHECK(hipMalloc((void**)&c, bytes));
    hipError_t e = hipMalloc((void**)&s, bytes);
    if (e == hipSuccess) e = hipMemcpy(c, cos_host, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(s, sin_host, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipFree(c); if (s) (void)hipFree(s); return (int)e; }
    drop_graph(m);                                                  // captured launches hold the old pointers
    (void)hipFree(m->cos_t); (void)hipFree(m->sin_t);
    m->cos_t = c; m->sin_t = s;
    return 0;
}

extern "C" void mi355_llama_destroy(void* mp) {
    Model* m = static_cast<Model*>(mp);
    if (!m) return;
    drop_graph(m);
    for (auto& L : m->layers) {
        for (auto& w : L.w) free_qw(w);
        if (L.attn_norm) (void)hipFree(L.attn_norm);
        if (L.ffn_norm) (void)hipFree(L.ffn_norm);
        if (L.gate_inp) (void)hipFree(L.gate_inp);
        for (uint8_t* p : L.eslab) if (p) (void)hipFree(p);
    }
    free_qw(m->output);
    void* ptrs[] = {m->tok_embd, m->output_norm, m->cos_t, m->sin_t, m->xs, m->q, m->attn, m->h, m->logits,
                    m->pa_tmp, m->pa_max, m->pa_sum, m->kv_slab, m->d_tokens, m->d_positions, m->d_slots,
                    m->d_ctx, m->d_bt, m->logits_local, m->logits_gather, m->tp_y, m->p_tp_y, m->p_moe_xg, m->p_moe_perm, m->p_moe_inv, m->p_moe_tab, m->p_xs, m->p_q, m->p_attn, m->p_h,
                    m->moe_ids, m->moe_w, m->moe_y, m->p_moe_ids, m->p_moe_w, m->p_moe_y, m->chain_sync, m->g_moe_xg, m->g_moe_h, m->g_moe_yg, m->g_moe_pos, m->g_moe_cnt};
    if (m->comm && m->comm_owned) mi355_comm_destroy(m->comm);
    for (void* p : ptrs) if (p) (void)hipFree(p);
    delete m;
}

extern "C" int mi355_llama_set_qweight(void* mp, int32_t layer, int32_t which, int32_t ggml_type,
                                       const void* native_host, int32_t n_rows, int32_t k) {
    Model* m = static_cast<Model*>(mp);
    QW* slot = m ? qw_slot(m, layer, which) : nullptr;
    if (!slot) return (int)hipErrorInvalidValue;
    if (ggml_type == MI355_GGML_Q8_0) {
        // a re-quantised tensor-parallel shard (quantized_var_builder.rs:234-269): native Q8_0 blocks, used as they are; only
        // the row-parallel projections can need it (o_proj / down_proj: plain epilogues, no fused norm)
        if ((which != MI355_W_WO && which != MI355_W_W2) || n_rows <= 0 || k <= 0 || (k % 32) || !native_host) return (int)hipErrorInvalidValue;
        const size_t nb = (size_t)n_rows * (k / 32) * 34;
        void* dev8 = nullptr;
        HCHECK(hipMalloc(&dev8, nb));
        HCHECK(hipMemcpy(dev8, native_host, nb, hipMemcpyHostToDevice));
        drop_graph(m);
        return set_qw(*slot, ggml_type, dev8, n_rows, k, true);
    }
    const int64_t n = mi355_qweight_repacked_size(ggml_type, n_rows, k);
    if (n < 0) return (int)hipErrorInvalidValue;
    std::vector<uint8_t> tiles((size_t)n);
    RCHECK(mi355_qweight_repack(tiles.data(), native_host, ggml_type, n_rows, k));
    void* dev = nullptr;
    HCHECK(hipMalloc(&dev, (size_t)n));
    HCHECK(hipMemcpy(dev, tiles.data(), (size_t)n, hipMemcpyHostToDevice));
    drop_graph(m);
    return set_qw(*slot, ggml_type, dev, n_rows, k, true);
}

extern "C" int mi355_llama_set_qweight_tiles(void* mp, int32_t layer, int32_t which, int32_t ggml_type,
                                             const void* tiles_dev, int32_t n_rows, int32_t k) {
    Model* m = static_cast<Model*>(mp);
    QW* slot = m ? qw_slot(m, layer, which) : nullptr;
    if (!slot || mi355_qweight_repacked_size(ggml_type, n_rows, k) < 0) return (int)hipErrorInvalidValue;
    drop_graph(m);
    return set_qw(*slot, ggml_type, tiles_dev, n_rows, k, false);
}

extern "C" int mi355_llama_set_moe_expert(void* mp, int32_t layer, int32_t which, int32_t expert, int32_t ggml_type,
                                          const void* native_host, int32_t n_rows, int32_t k) {
    Model* m = static_cast<Model*>(mp);
    if (!m || layer < 0 || layer >= m->cfg.n_layers || expert < 0 || expert >= m->cfg.n_expert || !native_host)
        return (int)hipErrorInvalidValue;
    const int slot = which == MI355_W_W1 ? 0 : (which == MI355_W_W2 ? 1 : (which == MI355_W_W3 ? 2 : -1));
    if (slot < 0) return (int)hipErrorInvalidValue;
    Layer& L = m->layers[layer];
    const int64_t n = mi355_qweight_repacked_size(ggml_type, n_rows, k);
    if (n < 0) return (int)hipErrorInvalidValue;
    if (!L.eslab[slot]) {
        HCHECK(hipMalloc((void**)&L.eslab[slot], (size_t)n * m->cfg.n_expert));
        L.etype[slot] = ggml_type; L.erows[slot] = n_rows; L.ek[slot] = k; L.estride[slot] = n;
    } else if (L.etype[slot] != ggml_type || L.erows[slot] != n_rows || L.ek[slot] != k) {
        return (int)hipErrorInvalidValue;                 // experts of one projection share type and shape
    }
    std::vector<uint8_t> tiles((size_t)n);
    RCHECK(mi355_qweight_repack(tiles.data(), native_host, ggml_type, n_rows, k));
    HCHECK(hipMemcpy(L.eslab[slot] + (size_t)expert * n, tiles.data(), (size_t)n, hipMemcpyHostToDevice));
    drop_graph(m);
    return 0;
}

extern "C" int mi355_llama_set_f32(void* mp, int32_t layer, int32_t which, const float* host, int64_t n) {
    Model* m = static_cast<Model*>(mp);
    if (!m || !host || n <= 0) return (int)hipErrorInvalidValue;
    float** dst = nullptr;
    if (layer < 0) {
        if (which == MI355_W_TOK_EMBD) dst = &m->tok_embd;
        else if (which == MI355_W_OUTPUT_NORM) dst = &m->output_norm;
    } else if (layer < m->cfg.n_layers) {
        if (which == MI355_W_ATTN_NORM) dst = &m->layers[layer].attn_norm;
        else if (which == MI355_W_FFN_NORM) dst = &m->layers[layer].ffn_norm;
        else if (which == MI355_W_GATE_INP) dst = &m->layers[layer].gate_inp;
    }
    if (!dst) return (int)hipErrorInvalidValue;
    if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
    HCHECK(hipMalloc((void**)dst, (size_t)n * 4));
    HCHECK(hipMemcpy(*dst, host, (size_t)n * 4, hipMemcpyHostToDevice));
    drop_graph(m);
    return 0;
}

// CacheEngine::allocate_kv_cache: per layer K and V, zero-initialised, layout by cfg.kv_layout.
extern "C" int mi355_llama_alloc_kv_cache(void* mp, int32_t num_blocks) {
    Model* m = static_cast<Model*>(mp);
    if (!m || num_blocks <= 0) return (int)hipErrorInvalidValue;
    const mi355_llama_config& c = m->cfg;
    if (c.kv_layout == MI355_KV_PAGED_FP8 && (c.head_dim % 16)) return (int)hipErrorInvalidValue;
    const size_t per = (size_t)num_blocks * c.block_size * local_kv_heads(m) * c.head_dim *
                       (c.kv_layout == MI355_KV_PAGED_FP8 ? 1 : 2);                                // e4m3 bytes or bf16
    if (m->kv_slab) { (void)hipFree(m->kv_slab); m->kv_slab = nullptr; }
    HCHECK(hipMalloc(&m->kv_slab, per * 2 * c.n_layers));
    HCHECK(hipMemset(m->kv_slab, 0, per * 2 * c.n_layers));
    HCHECK(hipDeviceSynchronize());
    m->kcache.resize(c.n_layers);
    m->vcache.resize(c.n_layers);
    for (int l = 0; l < c.n_layers; ++l) {
        m->kcache[l] = static_cast<uint8_t*>(m->kv_slab) + per * (2 * l);
        m->vcache[l] = static_cast<uint8_t*>(m->kv_slab) + per * (2 * l + 1);
    }
    m->num_blocks = num_blocks;
    drop_graph(m);
    return 0;
}

extern "C" int64_t mi355_llama_kv_bytes_per_tensor(void* mp);
extern "C" void* mi355_llama_kv_ptr(void* mp, int32_t layer, int32_t which) {
    Model* m = static_cast<Model*>(mp);
    if (!m || layer < 0 || layer >= (int)m->kcache.size()) return nullptr;
    return which == 0 ? m->kcache[layer] : m->vcache[layer];
}

// copy into / out of one layer's K or V tensor (src/dst may be host or device memory)
extern "C" int mi355_llama_kv_copy(void* mp, int32_t layer, int32_t which, void* buf, int64_t bytes, int32_t to_model) {
    Model* m = static_cast<Model*>(mp);
    void* kv = mi355_llama_kv_ptr(mp, layer, which);
    if (!m || !kv || !buf || bytes <= 0 || bytes > mi355_llama_kv_bytes_per_tensor(mp)) return (int)hipErrorInvalidValue;
    HCHECK(hipDeviceSynchronize());
    if (to_model) HCHECK(hipMemcpy(kv, buf, (size_t)bytes, hipMemcpyDefault));
    else HCHECK(hipMemcpy(buf, kv, (size_t)bytes, hipMemcpyDefault));
    return 0;
}

extern "C" int64_t mi355_llama_kv_bytes_per_tensor(void* mp) {
    Model* m = static_cast<Model*>(mp);
    if (!m) return -1;
    return (int64_t)m->num_blocks * m->cfg.block_size * local_kv_heads(m) * m->cfg.head_dim *
           (m->cfg.kv_layout == MI355_KV_PAGED_FP8 ? 1 : 2);
}

static int ensure_prefill_cap(Model* m, int T) {
    if (T <= m->p_cap) return 0;
    void* old[] = {m->p_xs, m->p_q, m->p_attn, m->p_h, m->p_tp_y};
    HCHECK(hipDeviceSynchronize());
    for (void* p : old) if (p) (void)hipFree(p);
    m->p_xs = nullptr; m->p_q = nullptr; m->p_attn = nullptr; m->p_h = nullptr; m->p_tp_y = nullptr; m->p_cap = 0;
    const int H = local_heads(m), D = m->cfg.head_dim;
    const int cap = (T + 255) / 256 * 256;
    HCHECK(hipMalloc((void**)&m->p_xs, (size_t)cap * m->cfg.hidden * 4));
    HCHECK(hipMalloc((void**)&m->p_q, (size_t)cap * H * D * 2));
    HCHECK(hipMalloc((void**)&m->p_attn, (size_t)cap * H * D * 2));
    const int KE = m->cfg.n_expert > 1 ? m->cfg.n_expert_used : 1;
    // (+ 64 rows per expert: the device-grouped prompt path gives every expert whole 64-row blocks)
    const size_t moe_pad = m->cfg.n_expert > 1 ? (size_t)64 * (m->cfg.n_expert + 2) : 0;
    HCHECK(hipMalloc((void**)&m->p_h, ((size_t)cap * KE + moe_pad) * m->cfg.intermediate * 4));
    if (m->use_comm) HCHECK(hipMalloc((void**)&m->p_tp_y, (size_t)cap * m->cfg.hidden * 4));
    if (m->cfg.n_expert > 1) {
        void* oldm[] = {m->p_moe_ids, m->p_moe_w, m->p_moe_y, m->p_moe_xg, m->p_moe_perm, m->p_moe_inv, m->p_moe_tab};
        for (void* p : oldm) if (p) (void)hipFree(p);
        m->p_moe_ids = nullptr; m->p_moe_w = nullptr; m->p_moe_y = nullptr;
        m->p_moe_xg = nullptr; m->p_moe_perm = nullptr; m->p_moe_inv = nullptr; m->p_moe_tab = nullptr;
        HCHECK(hipMalloc((void**)&m->p_moe_tab, ((size_t)cap * KE / 64 + m->cfg.n_expert + 4) * 8));
        HCHECK(hipMalloc((void**)&m->p_moe_xg, ((size_t)cap * KE + moe_pad) * m->cfg.hidden * 4));
        HCHECK(hipMemsetAsync(m->p_moe_xg, 0, ((size_t)cap * KE + moe_pad) * m->cfg.hidden * 4, 0));   // the rows no pair lands in: finite from the start
        HCHECK(hipMalloc((void**)&m->p_moe_perm, (size_t)cap * KE * 4));
        HCHECK(hipMalloc((void**)&m->p_moe_inv, (size_t)cap * KE * 4));
        HCHECK(hipMalloc((void**)&m->p_moe_ids, (size_t)cap * KE * 4));
        HCHECK(hipMalloc((void**)&m->p_moe_w, (size_t)cap * KE * 4));
        HCHECK(hipMalloc((void**)&m->p_moe_y, ((size_t)cap * KE + moe_pad) * m->cfg.hidden * 4));
        HCHECK(hipDeviceSynchronize());
    }
    m->p_cap = cap;
    return 0;
}

// GGUFLLaMa::forward_inner on a PROMPT step (quantized_llama.rs:424-506 with input_metadata.is_prefill):
// tokens of all sequences flattened [T]; logits f32 [num_seqs, vocab] of each sequence's last chunk token.
extern "C" int mi355_llama_forward_prefill(void* mp, const uint32_t* tokens, const int64_t* positions,
                                           const int64_t* slot_mapping, const uint32_t* block_tables,
                                           const uint32_t* context_lens, const uint32_t* cu_seqlens_q,
                                           int32_t num_seqs, int32_t num_tokens, int32_t max_seqlen_q,
                                           int32_t max_blocks, float* logits, int64_t stream) {
    Model* m = static_cast<Model*>(mp);
    if (!m || !logits || num_seqs < 1 || num_tokens < num_seqs || num_seqs > m->cfg.max_batch)
        return (int)hipErrorInvalidValue;
    const mi355_llama_config& c = m->cfg;
    if ((int)m->kcache.size() != c.n_layers) return (int)hipErrorInvalidValue;
    RCHECK(ensure_prefill_cap(m, num_tokens));
    StepIn in{tokens, positions, slot_mapping, block_tables, context_lens, num_tokens, max_blocks, 0,
              m->p_xs, m->p_q, m->p_attn, m->p_h, m->p_moe_ids, m->p_moe_w, m->p_moe_y, m->p_tp_y, true, cu_seqlens_q, num_seqs, max_seqlen_q};
    RCHECK(run_part(m, 0, PART_EMBED, in, logits, stream));
    for (int l = 0; l < c.n_layers; ++l)
        for (int part = PART_QKV; part <= PART_DOWN; ++part) RCHECK(run_part(m, l, part, in, logits, stream));
    // last-token row select, then final norm + lm_head on num_seqs rows (the decode buffers are large enough)
    hipLaunchKernelGGL(select_last_rows_kernel, dim3(num_seqs), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       m->xs, m->p_xs, cu_seqlens_q, c.hidden);
    in.xs = m->xs; in.B = num_seqs;
    return run_part(m, 0, PART_HEAD, in, logits, stream);
}

extern "C" int mi355_llama_forward_decode(void* mp, const uint32_t* tokens, const int64_t* positions,
                                          const int64_t* slot_mapping, const uint32_t* block_tables,
                                          const uint32_t* context_lens, int32_t batch, int32_t max_blocks,
                                          int32_t max_context_len, float* logits, int64_t stream) {
    Model* m = static_cast<Model*>(mp);
    if (!m || !logits) return (int)hipErrorInvalidValue;
    return forward_decode(m, tokens, positions, slot_mapping, block_tables, context_lens, batch, max_blocks,
                          max_context_len, logits, stream);
}

// ---- greedy decode loop on static device buffers (graph replay) ---------------------------------------------
// tokens_host[b]   : last token of sequence b (the one this step consumes)
// seq_lens_host[b] : sequence length INCLUDING that token (= context_len of this step)
// block_tables_host: [batch, max_blocks] (blocks for every step that will be run must already be reserved,
//                    as the scheduler does ahead of the step)
// ctx_cap          : upper bound of the context length over the steps to come (sizes the attention grid)
extern "C" int mi355_llama_decode_begin(void* mp, const uint32_t* tokens_host, const uint32_t* seq_lens_host,
                                        const uint32_t* block_tables_host, int32_t batch, int32_t max_blocks,
                                        int32_t ctx_cap, int64_t stream) {
    Model* m = static_cast<Model*>(mp);
    if (!m || batch < 1 || batch > m->cfg.max_batch || max_blocks < 1 || max_blocks > m->cfg.max_blocks_per_seq)
        return (int)hipErrorInvalidValue;
    const int bs = m->cfg.block_size;
    std::vector<int64_t> pos(batch), slot(batch);
    for (int b = 0; b < batch; ++b) {
        if (seq_lens_host[b] < 1) return (int)hipErrorInvalidValue;
        pos[b] = (int64_t)seq_lens_host[b] - 1;
        if (pos[b] / bs >= max_blocks) return (int)hipErrorInvalidValue;          // "Block table is too small"
        slot[b] = (int64_t)block_tables_host[(size_t)b * max_blocks + pos[b] / bs] * bs + pos[b] % bs;
    }
    // the greedy loop advances the context on the device: everything it will index is checked here and in decode_step
    int ctx_max = 0;
    for (int b = 0; b < batch; ++b) ctx_max = std::max(ctx_max, (int)seq_lens_host[b]);
    if (ctx_cap < ctx_max || ctx_cap > m->cfg.max_seq || ctx_max > max_blocks * bs) return (int)hipErrorInvalidValue;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    HCHECK(hipMemcpyAsync(m->d_tokens, tokens_host, (size_t)batch * 4, hipMemcpyHostToDevice, st));
    HCHECK(hipMemcpyAsync(m->d_ctx, seq_lens_host, (size_t)batch * 4, hipMemcpyHostToDevice, st));
    HCHECK(hipMemcpyAsync(m->d_bt, block_tables_host, (size_t)batch * max_blocks * 4, hipMemcpyHostToDevice, st));
    HCHECK(hipMemcpyAsync(m->d_positions, pos.data(), (size_t)batch * 8, hipMemcpyHostToDevice, st));
    HCHECK(hipMemcpyAsync(m->d_slots, slot.data(), (size_t)batch * 8, hipMemcpyHostToDevice, st));
    HCHECK(hipStreamSynchronize(st));                     // host vectors go out of scope
    m->cur_batch = batch; m->cur_max_blocks = max_blocks; m->cur_ctx_cap = ctx_cap; m->cur_ctx_max = ctx_max;
    return 0;
}

static int record_step(Model* m, int64_t stream) {
    const int B = m->cur_batch;
    RCHECK(forward_decode(m, m->d_tokens, m->d_positions, m->d_slots, m->d_bt, m->d_ctx, B, m->cur_max_blocks,
                          m->cur_ctx_cap, m->logits, stream));
    // greedy sample, then prepare the next step's inputs on the device
    uint32_t* next = reinterpret_cast<uint32_t*>(m->q);   // scratch: q is dead after the last layer
    // under TP the lm_head holds pad_vocab / world rows; the gathered logits row is narrowed to the real vocabulary
    RCHECK(mi355_argmax_f32(next, m->logits, B, m->use_comm ? m->cfg.vocab : m->output.n_rows, stream));
    hipLaunchKernelGGL(advance_kernel, dim3((B + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream),
                       m->d_tokens, next, m->d_positions, m->d_slots, m->d_ctx, m->d_bt, m->cur_max_blocks,
                       m->cfg.block_size, B);
    return (int)hipGetLastError();
}

/* PARITY MODE switch (tests): 0 = the product attention kernels (f32 scores and probabilities), 1 = decode attention with the
 * reference CPU path's bf16 rounding points (models/mod.rs:1288-1306) -- slow; fp8 caches and prompt steps keep their kernels. */
extern "C" void mi355_internal_qmm_set_exact(int32_t on);             // qmatmul.hip (qmm_exact.inc): process-wide, tests only
extern "C" int mi355_llama_set_attention_numerics(void* mp, int32_t mode) {
    Model* m = static_cast<Model*>(mp);
    if (!m || mode < 0 || mode > 2) return (int)hipErrorInvalidValue;
    if (mode >= 1 && m->cfg.kv_layout == MI355_KV_PAGED_FP8) return (int)hipErrorNotSupported;
    drop_graph(m);
    m->attn_numerics = mode >= 1 ? 1 : 0;
    mi355_internal_qmm_set_exact(mode == 2 ? 1 : 0);                  // 2: + every mat-vec of the step exact to f32 rounding
    return 0;
}

extern "C" int mi355_llama_set_graph(void* mp, int32_t enable) {
    Model* m = static_cast<Model*>(mp);
    if (!m) return (int)hipErrorInvalidValue;
    m->use_graph = enable != 0;
    m->graph_tp = enable == 2;
    if (!m->use_graph) drop_graph(m);
    m->warmed.clear();                                                // a (re-)enabled graph mode starts with an eager step per shape
    return 0;
}

// One greedy decode step: logits -> argmax -> next-step inputs.  With graphs enabled the launch sequence is
// captured once per (batch, max_blocks, ctx_cap) and replayed (graph.rs:471-661 captures per batch size).
extern "C" int mi355_llama_decode_step(void* mp, int64_t stream) {
    Model* m = static_cast<Model*>(mp);
    if (!m || m->cur_batch < 1) return (int)hipErrorInvalidValue;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // this step attends over cur_ctx_max tokens, then `advance_kernel` looks up the slot of position cur_ctx_max: both
    // must stay inside the attention grid (ctx_cap), the block-table row and the RoPE tables
    if (m->cur_ctx_max > m->cur_ctx_cap || m->cur_ctx_max > m->cfg.max_seq ||
        (m->cur_ctx_max + m->cfg.block_size - 1) / m->cfg.block_size > m->cur_max_blocks)
        return (int)hipErrorInvalidValue;
    // (the host mirror of the context length advances only when a step was actually enqueued: a refused or failed step leaves the
    // device-side context where it was, and a retry must pass the same checks again -- ADVICE r4)
    auto stepped = [m](int rc) { if (rc == 0) ++m->cur_ctx_max; return rc; };
    auto eager = [m, stream]() { ++m->eager_steps; return record_step(m, stream); };
    // TP steps run eagerly (RCCL in-stream) unless the caller opted in with set_graph(2) AND the communicator is RCCL's
    // own (host-supplied collectives stage through the host and cannot be captured)
    // TP steps are captured like single-GPU ones when every collective of the step is device-native (RCCL on its side
    // stream joins the capture as a fork / join; the one-shot peer kernel is an ordinary node whose sequence numbers live
    // on the device).  Host-supplied collectives stage through the host and cannot be captured: those steps stay eager.
    const bool tp_eager = m->use_comm && !(m->comm && m->comm->nccl);
    if (!m->use_graph || stream == 0 || tp_eager) return stepped(eager());
    const std::array<int, 3> shape{m->cur_batch, m->cur_max_blocks, m->cur_ctx_cap};
    if (!m->warmed.count(shape)) {
        // first step of a new shape runs eagerly: lazily-set kernel attributes and occupancy queries must not
        // happen inside a stream capture
        m->warmed.insert(shape);
        return stepped(eager());
    }
    auto it = m->graphs.find(shape);
    if (it == m->graphs.end()) {
        if (m->graphs.size() >= MI355_MAX_STEP_GRAPHS) {              // evict the least recently replayed shape
            auto lru = m->graphs.begin();
            for (auto q = m->graphs.begin(); q != m->graphs.end(); ++q) if (q->second.used < lru->second.used) lru = q;
            if (lru->second.exec) (void)hipGraphExecDestroy(lru->second.exec);
            if (lru->second.graph) (void)hipGraphDestroy(lru->second.graph);
            m->graphs.erase(lru);
        }
        HCHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        const int rc = record_step(m, stream);
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(st, &g);
        if (rc != 0) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (e != hipSuccess) return (int)e;
        Model::StepGraph sg;
        sg.graph = g;
        const hipError_t ie = hipGraphInstantiate(&sg.exec, sg.graph, nullptr, nullptr, 0);
        if (ie != hipSuccess) { (void)hipGraphDestroy(g); return (int)ie; }
        it = m->graphs.emplace(shape, sg).first;
        ++m->graph_captures;
    }
    it->second.used = ++m->graph_clock;
    HCHECK(hipGraphLaunch(it->second.exec, st));
    return stepped(0);
}

// how many step graphs this model captured + instantiated so far / how many steps of the greedy loop ran eagerly: a bench that
// brackets its timed region with these can show that no capture (tens of ms) and no eager step fell inside it
extern "C" int64_t mi355_llama_graph_captures(void* mp) {
    Model* m = static_cast<Model*>(mp);
    return m ? m->graph_captures : -1;
}
extern "C" int64_t mi355_llama_eager_steps(void* mp) {
    Model* m = static_cast<Model*>(mp);
    return m ? m->eager_steps : -1;
}

// D2H of the tokens the last step sampled (they are the inputs of the next step)
extern "C" int mi355_llama_decode_read_tokens(void* mp, uint32_t* host_out, int64_t stream) {
    Model* m = static_cast<Model*>(mp);
    if (!m || m->cur_batch < 1) return (int)hipErrorInvalidValue;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    HCHECK(hipMemcpyAsync(host_out, m->d_tokens, (size_t)m->cur_batch * 4, hipMemcpyDeviceToHost, st));
    // the one-shot peer all-reduce's sticky error word rides along in the same synchronisation: a peer that missed the spin
    // bound poisoned that step's activations with NaN (comm.hip), and the tokens read here must not be used
    uint32_t p2p_err = 0;
    const bool p2p = m->use_comm && m->comm && m->comm->p2p && m->comm->local;
    if (p2p) HCHECK(hipMemcpyAsync(&p2p_err, &m->comm->local->err, 4, hipMemcpyDeviceToHost, st));
    HCHECK(hipStreamSynchronize(st));
    if (p2p && p2p_err != 0) return (int)hipErrorPeerAccessNotEnabled;   // "a peer was not there": the step is invalid on this rank
    return 0;
}

extern "C" float* mi355_llama_logits_ptr(void* mp) {
    Model* m = static_cast<Model*>(mp);
    return m ? m->logits : nullptr;
}

// ---- GGUFLLaMa::from_gguf (quantized_llama.rs:203-420) over the GGUF reader ------------------------------------
// tp_world > 1: every rank reads the same file and keeps its raw byte-range shard of each tensor, as
// `get_sharded_no_shape` does (quantized_var_builder.rs:135-183,222-233): attn_q / ffn_gate / ffn_up / output on rows,
// attn_k / attn_v on rows of the kv-head shard (`kv_head_shard`, distributed.rs:725-765: replicated groups when
// Hkv < W), attn_output / ffn_down on k-blocks; token_embd, the norms and a Mixtral layer's router + experts are
// loaded whole (quantized_llama.rs:262-268,344-365).  o_proj / down_proj shards that cut a k-block take the reference's
// dequantise -> narrow -> re-quantise (Q8_0) fallback; the lm_head of a vocabulary `pad_vocab_size` pads keeps its own blocks
// and gets all-zero blocks for the rows beyond the vocabulary (VocabParallelLinear, distributed.rs:1596-1616).
namespace {
int load_gguf_impl(const char* path, int32_t max_batch, int32_t max_blocks_per_seq, int32_t block_size, int32_t kv_layout,
                   int32_t max_seq, int32_t tp_rank, int32_t tp_world, void** model_out, mi355_llama_config* cfg_out,
                   bool check_only = false) {
    if (!path || (!model_out && !check_only) || tp_world < 1 || tp_rank < 0 || tp_rank >= tp_world) return (int)hipErrorInvalidValue;
    if (model_out) *model_out = nullptr;
    void* g = mi355_gguf_open(path);
    if (!g) return (int)hipErrorFileNotFound;
    struct Closer { void* g; ~Closer() { mi355_gguf_close(g); } } closer{g};
    char arch[64];
    if (mi355_gguf_get_str(g, "general.architecture", arch, sizeof(arch)) <= 0) return (int)hipErrorInvalidValue;
    auto key = [&](const char* suffix) { return std::string(arch) + "." + suffix; };
    auto u = [&](const std::string& k, uint64_t* out) { return mi355_gguf_get_u64(g, k.c_str(), out) == 1; };
    uint64_t head_count = 0, head_count_kv = 0, block_count = 0, embd = 0, ctx_len = 8192, key_len = 0, n_expert = 0;
    if (!u(key("attention.head_count"), &head_count) || !u(key("attention.head_count_kv"), &head_count_kv) ||
        !u(key("block_count"), &block_count) || !u(key("embedding_length"), &embd))
        return (int)hipErrorInvalidValue;
    (void)u(key("context_length"), &ctx_len);
    (void)u(key("expert_count"), &n_expert);
    uint64_t n_expert_used = 0;
    (void)u(key("expert_used_count"), &n_expert_used);
    // metadata is untrusted: bound everything before it is narrowed to int32 or used as a divisor
    const uint64_t lim = 1u << 30;
    const bool sane = head_count && head_count <= lim && head_count_kv && head_count_kv <= lim && block_count &&
                      block_count <= 4096 && embd && embd <= lim && ctx_len && ctx_len <= lim && n_expert <= 1024 &&
                      (n_expert <= 1 || (n_expert_used >= 1 && n_expert_used <= n_expert));
    if (!sane) return (int)hipErrorInvalidValue;
    if (!u(key("attention.key_length"), &key_len)) key_len = embd / head_count;
    if (key_len == 0 || key_len > 4096) return (int)hipErrorInvalidValue;
    // partial rotary (`rope.dimension_count` != head_dim -> partial_rotary_factor, quantized_llama.rs:289-298): the GGUF
    // step here rotates the whole head; refuse rather than rotate the wrong dimensions
    uint64_t rope_dim = 0;
    if (u(key("rope.dimension_count"), &rope_dim) && rope_dim != key_len) return (int)hipErrorNotSupported;
    double eps = 0, theta = 10000.0;
    if (mi355_gguf_get_f64(g, key("attention.layer_norm_rms_epsilon").c_str(), &eps) != 1) return (int)hipErrorInvalidValue;
    (void)mi355_gguf_get_f64(g, key("rope.freq_base").c_str(), &theta);

    auto info = [&](const std::string& name, int64_t* dims, int32_t* type, uint64_t* nbytes) {
        const int i = mi355_gguf_find(g, name.c_str());
        if (i < 0) return -1;
        int32_t nd = 0;
        if (mi355_gguf_tensor_info(g, i, nullptr, 0, dims, &nd, type, nbytes) != 0) return -1;
        return i;
    };
    int64_t d[4]; int32_t ty; uint64_t nb;
    const int i_embd = info("token_embd.weight", d, &ty, &nb);
    if (i_embd < 0) return (int)hipErrorInvalidValue;
    const int64_t vocab = d[0];
    int64_t dff[4]; int32_t tff; uint64_t nbff;
    if (info(n_expert > 1 ? "blk.0.ffn_gate.0.weight" : "blk.0.ffn_gate.weight", dff, &tff, &nbff) < 0) return (int)hipErrorInvalidValue;

    if (vocab <= 0 || vocab > (int64_t)lim || dff[0] <= 0 || dff[0] > (int64_t)lim) return (int)hipErrorInvalidValue;
    if (ty != 0 && ty != MI355_GGML_Q4_K && ty != MI355_GGML_Q6_K) return (int)hipErrorNotSupported;   // token_embd: F32 or a k-quant we dequantise
    mi355_llama_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.hidden = (int32_t)embd; cfg.n_layers = (int32_t)block_count; cfg.n_heads = (int32_t)head_count;
    cfg.n_kv_heads = (int32_t)head_count_kv; cfg.head_dim = (int32_t)key_len; cfg.intermediate = (int32_t)dff[0];
    cfg.vocab = (int32_t)vocab;
    cfg.max_seq = (max_seq > 0 && (uint64_t)max_seq < ctx_len) ? max_seq : (int32_t)ctx_len;
    cfg.block_size = block_size; cfg.kv_layout = kv_layout; cfg.max_batch = max_batch; cfg.max_blocks_per_seq = max_blocks_per_seq;
    cfg.rms_eps = (float)eps; cfg.rope_theta = (float)theta; cfg.tp_rank = tp_rank; cfg.tp_world = tp_world;
    cfg.n_expert = n_expert > 1 ? (int32_t)n_expert : 0; cfg.n_expert_used = n_expert > 1 ? (int32_t)n_expert_used : 0;
    // shard plan of this rank (attention.rs:553-554, distributed.rs:725-765,1446-1452)
    int kv_rank = tp_rank, kv_world = tp_world;
    if (tp_world > 1) {
        if (cfg.n_heads % tp_world) return (int)hipErrorInvalidValue;
        if (cfg.n_kv_heads >= tp_world) {
            if (cfg.n_kv_heads % tp_world) return (int)hipErrorInvalidValue;
        } else {
            if (tp_world % cfg.n_kv_heads) return (int)hipErrorInvalidValue;
            kv_rank = tp_rank / (tp_world / cfg.n_kv_heads); kv_world = cfg.n_kv_heads;   // one replicated head per rank
        }
        // a vocabulary that is not a fixed point of pad_vocab_size gets zero rows in the lm_head shards (below); the gathered logits
        // are narrowed back (run_part, PART_HEAD)
    }
    // ---- the tensor table against the configuration, BEFORE anything is allocated on the device: a file whose shapes
    // disagree with its own metadata would otherwise make the kernels read past a weight (the reference checks the
    // same dims while building the layers, attention.rs:850-870).  Host only: testable without a GPU.
    {
        const int64_t hid = cfg.hidden, HD = (int64_t)cfg.n_heads * cfg.head_dim, KD = (int64_t)cfg.n_kv_heads * cfg.head_dim;
        const int64_t I = cfg.intermediate;
        if (cfg.hidden <= 0 || cfg.n_layers <= 0 || cfg.n_layers > 4096 || cfg.n_heads <= 0 || cfg.n_kv_heads <= 0 ||
            cfg.head_dim <= 0 || I <= 0 || vocab <= 0 || (hid % 256) || cfg.n_expert > 1024)
            return (int)hipErrorInvalidValue;
        // dim < 0: loaded whole; else this rank keeps rows (0) / k-blocks (1) of a `world`-way split
        auto expect_q = [&](const std::string& name, int64_t rows, int64_t cols, int dim, int world) -> int {
            int64_t dd[4]; int32_t t; uint64_t n;
            if (info(name, dd, &t, &n) < 0) return (int)hipErrorInvalidValue;
            if (t != MI355_GGML_Q4_K && t != MI355_GGML_Q6_K) return (int)hipErrorNotSupported;
            if (dd[0] != rows || dd[1] != cols || dd[2] != 1 || dd[3] != 1) return (int)hipErrorInvalidValue;
            if (dim == 0 && world > 1) { if (rows % world) return (int)hipErrorInvalidValue; rows /= world; }
            if (dim == 1 && world > 1) { if (cols % world) return (int)hipErrorInvalidValue; cols /= world; }
            if (cols % 256) {
                // a dim-1 shard that cuts a k-quant block is re-quantised to Q8_0 (quantized_var_builder.rs:234-269): 32-wide blocks
                if (dim == 1 && world > 1) return (cols % 32) ? (int)hipErrorNotSupported : 0;
                return (int)hipErrorInvalidValue;
            }
            // whole 16-row tiles are required only where the fused QKV epilogue needs them (mi355_qmatmul_fused checks the
            // same); everywhere else the last tile may be ragged (odd vocabularies)
            const bool qkv = name.find("attn_q.") != std::string::npos || name.find("attn_k.") != std::string::npos ||
                             name.find("attn_v.") != std::string::npos;
            return (qkv && rows % 16) ? (int)hipErrorInvalidValue : 0;
        };
        auto expect_f32 = [&](const std::string& name, int64_t elems) -> int {
            int64_t dd[4]; int32_t t; uint64_t n;
            if (info(name, dd, &t, &n) < 0 || t != 0) return (int)hipErrorInvalidValue;
            return (int64_t)(n / 4) == elems ? 0 : (int)hipErrorInvalidValue;
        };
        if (d[0] != vocab || d[1] != hid || d[2] != 1 || d[3] != 1) return (int)hipErrorInvalidValue;      // token_embd
        RCHECK(expect_f32("output_norm.weight", hid));
        // (the lm_head is sharded by rows of the PADDED vocabulary: any row count works, the shards carry zero rows)
        RCHECK(expect_q(mi355_gguf_find(g, "output.weight") >= 0 ? "output.weight" : "token_embd.weight", vocab, hid, -1, 1));
        for (int l = 0; l < cfg.n_layers; ++l) {
            const std::string p = "blk." + std::to_string(l) + ".";
            RCHECK(expect_q(p + "attn_q.weight", HD, hid, 0, tp_world));
            RCHECK(expect_q(p + "attn_k.weight", KD, hid, 0, kv_world));
            RCHECK(expect_q(p + "attn_v.weight", KD, hid, 0, kv_world));
            RCHECK(expect_q(p + "attn_output.weight", hid, HD, 1, tp_world));
            RCHECK(expect_f32(p + "attn_norm.weight", hid));
            RCHECK(expect_f32(p + "ffn_norm.weight", hid));
            // tensors the reference's QuantizedAttention would use when present (qkv bias, per-head q/k norm:
            // attention.rs:811-866) and this step does not apply: refuse instead of computing something else
            for (const char* extra : {"attn_q.bias", "attn_k.bias", "attn_v.bias", "attn_q_norm.weight", "attn_k_norm.weight"})
                if (mi355_gguf_find(g, (p + extra).c_str()) >= 0) return (int)hipErrorNotSupported;
            if (cfg.n_expert > 1) {
                RCHECK(expect_f32(p + "ffn_gate_inp.weight", (int64_t)cfg.n_expert * hid));
                for (int e = 0; e < cfg.n_expert; ++e) {
                    const std::string es = std::to_string(e) + ".weight";
                    RCHECK(expect_q(p + "ffn_gate." + es, I, hid, -1, 1));
                    RCHECK(expect_q(p + "ffn_down." + es, hid, I, -1, 1));
                    RCHECK(expect_q(p + "ffn_up." + es, I, hid, -1, 1));
                }
            } else {
                RCHECK(expect_q(p + "ffn_gate.weight", I, hid, 0, tp_world));
                RCHECK(expect_q(p + "ffn_down.weight", hid, I, 1, tp_world));
                RCHECK(expect_q(p + "ffn_up.weight", I, hid, 0, tp_world));
            }
        }
    }
    if (check_only) {
        if (cfg_out) *cfg_out = cfg;
        return 0;
    }
    Model* m = static_cast<Model*>(mi355_llama_create(&cfg));
    if (!m) return (int)hipErrorInvalidValue;
    struct Guard { Model* m; bool keep = false; ~Guard() { if (!keep) mi355_llama_destroy(m); } } guard{m};

    // dim < 0: the whole tensor; dim 0 / 1: this rank's row / k-block shard
    auto load_q = [&](const std::string& name, int layer, int which, int dim = -1, int rank = 0, int world = 1) -> int {
        int64_t dd[4]; int32_t t; uint64_t n;
        const int i = info(name, dd, &t, &n);
        if (i < 0) return (int)hipErrorInvalidValue;
        if (t != MI355_GGML_Q4_K && t != MI355_GGML_Q6_K) return (int)hipErrorNotSupported;   // other ggml types: out of scope
        if (dim < 0 || world <= 1)
            return mi355_llama_set_qweight(m, layer, which, t, mi355_gguf_tensor_data(g, i), (int32_t)dd[0], (int32_t)dd[1]);
        const int64_t bytes = mi355_gguf_tensor_shard(g, i, dim, rank, world, nullptr, 0);
        if (bytes == -2 && dim == 1 && (which == MI355_W_WO || which == MI355_W_W2)) {
            // the shard cuts a k-quant block: the reference's dequantise -> narrow -> re-quantise-to-Q8_0 fallback
            // (quantized_var_builder.rs:234-269)
            const int64_t b8 = mi355_gguf_tensor_shard_q8_0(g, i, rank, world, nullptr, 0);
            if (b8 < 0) return (int)hipErrorNotSupported;
            std::vector<uint8_t> shard8((size_t)b8);
            if (mi355_gguf_tensor_shard_q8_0(g, i, rank, world, shard8.data(), b8) != b8) return (int)hipErrorInvalidValue;
            return mi355_llama_set_qweight(m, layer, which, MI355_GGML_Q8_0, shard8.data(), (int32_t)dd[0], (int32_t)(dd[1] / world));
        }
        if (bytes < 0) return bytes == -2 ? (int)hipErrorNotSupported : (int)hipErrorInvalidValue;
        std::vector<uint8_t> shard((size_t)bytes);
        if (mi355_gguf_tensor_shard(g, i, dim, rank, world, shard.data(), bytes) != bytes) return (int)hipErrorInvalidValue;
        const int32_t rows = (int32_t)(dim == 0 ? dd[0] / world : dd[0]), k = (int32_t)(dim == 1 ? dd[1] / world : dd[1]);
        return mi355_llama_set_qweight(m, layer, which, t, shard.data(), rows, k);
    };
    auto load_f32 = [&](const std::string& name, int layer, int which) -> int {
        int64_t dd[4]; int32_t t; uint64_t n;
        const int i = info(name, dd, &t, &n);
        if (i < 0 || t != 0) return (int)hipErrorInvalidValue;
        return mi355_llama_set_f32(m, layer, which, static_cast<const float*>(mi355_gguf_tensor_data(g, i)), (int64_t)(n / 4));
    };
    // token_embd: dequantised once (quantized_llama.rs:262-264)
    if (ty == 0) {
        RCHECK(mi355_llama_set_f32(m, -1, MI355_W_TOK_EMBD, static_cast<const float*>(mi355_gguf_tensor_data(g, i_embd)), vocab * (int64_t)embd));
    } else if (ty == MI355_GGML_Q4_K || ty == MI355_GGML_Q6_K) {
        void* native = nullptr;
        HCHECK(hipMalloc(&native, nb));
        HCHECK(hipMemcpy(native, mi355_gguf_tensor_data(g, i_embd), nb, hipMemcpyHostToDevice));
        if (m->tok_embd) { (void)hipFree(m->tok_embd); m->tok_embd = nullptr; }
        hipError_t e = hipMalloc((void**)&m->tok_embd, (size_t)vocab * embd * 4);
        int rc = e == hipSuccess ? mi355_dequantize(m->tok_embd, native, ty, vocab * (int64_t)embd, 0) : (int)e;
        (void)hipDeviceSynchronize();
        (void)hipFree(native);
        if (rc) return rc;
    } else {
        return (int)hipErrorNotSupported;
    }
    RCHECK(load_f32("output_norm.weight", -1, MI355_W_OUTPUT_NORM));
    {   // vocab-parallel lm_head: rows [rank * local, (rank + 1) * local) of the vocabulary padded to pad_vocab_size, zero rows beyond
        // the real one (VocabParallelLinear, distributed.rs:1596-1616; layout note in DESIGN.md section 7)
        const std::string name = mi355_gguf_find(g, "output.weight") >= 0 ? "output.weight" : "token_embd.weight";
        if (tp_world <= 1) {
            RCHECK(load_q(name, -1, MI355_W_OUTPUT));
        } else {
            int64_t dd[4]; int32_t t; uint64_t n;
            const int i = info(name, dd, &t, &n);
            if (i < 0) return (int)hipErrorInvalidValue;
            if (t != MI355_GGML_Q4_K && t != MI355_GGML_Q6_K) return (int)hipErrorNotSupported;
            if (pad_vocab(vocab, tp_world) % tp_world != 0) return (int)hipErrorInvalidValue;   // (also refused by mi355_llama_create)
            const int64_t local = pad_vocab(vocab, tp_world) / tp_world;
            const int64_t bytes = mi355_gguf_tensor_rows_padded(g, i, (int64_t)tp_rank * local, local, nullptr, 0);
            if (bytes < 0) return (int)hipErrorInvalidValue;
            std::vector<uint8_t> shard((size_t)bytes);
            if (mi355_gguf_tensor_rows_padded(g, i, (int64_t)tp_rank * local, local, shard.data(), bytes) != bytes) return (int)hipErrorInvalidValue;
            RCHECK(mi355_llama_set_qweight(m, -1, MI355_W_OUTPUT, t, shard.data(), (int32_t)local, (int32_t)dd[1]));
        }
    }
    for (int l = 0; l < cfg.n_layers; ++l) {
        const std::string p = "blk." + std::to_string(l) + ".";
        RCHECK(load_q(p + "attn_q.weight", l, MI355_W_WQ, 0, tp_rank, tp_world));
        RCHECK(load_q(p + "attn_k.weight", l, MI355_W_WK, 0, kv_rank, kv_world));
        RCHECK(load_q(p + "attn_v.weight", l, MI355_W_WV, 0, kv_rank, kv_world));
        RCHECK(load_q(p + "attn_output.weight", l, MI355_W_WO, 1, tp_rank, tp_world));
        if (cfg.n_expert > 1) {                               // quantized_llama.rs:347-365
            RCHECK(load_f32(p + "ffn_gate_inp.weight", l, MI355_W_GATE_INP));
            for (int e = 0; e < cfg.n_expert; ++e) {
                const char* names[3] = {"ffn_gate.", "ffn_down.", "ffn_up."};
                const int which[3] = {MI355_W_W1, MI355_W_W2, MI355_W_W3};
                for (int q = 0; q < 3; ++q) {
                    int64_t dd[4]; int32_t t; uint64_t nbq;
                    const int i = info(p + names[q] + std::to_string(e) + ".weight", dd, &t, &nbq);
                    if (i < 0) return (int)hipErrorInvalidValue;
                    RCHECK(mi355_llama_set_moe_expert(m, l, which[q], e, t, mi355_gguf_tensor_data(g, i), (int32_t)dd[0], (int32_t)dd[1]));
                }
            }
        } else {
            RCHECK(load_q(p + "ffn_gate.weight", l, MI355_W_W1, 0, tp_rank, tp_world));
            RCHECK(load_q(p + "ffn_down.weight", l, MI355_W_W2, 1, tp_rank, tp_world));
            RCHECK(load_q(p + "ffn_up.weight", l, MI355_W_W3, 0, tp_rank, tp_world));
        }
        RCHECK(load_f32(p + "attn_norm.weight", l, MI355_W_ATTN_NORM));
        RCHECK(load_f32(p + "ffn_norm.weight", l, MI355_W_FFN_NORM));
    }
    guard.keep = true;
    *model_out = m;
    if (cfg_out) *cfg_out = cfg;
    return 0;
}
}  // namespace

extern "C" int mi355_llama_load_gguf(const char* path, int32_t max_batch, int32_t max_blocks_per_seq, int32_t block_size,
                                     int32_t kv_layout, int32_t max_seq, void** model_out, mi355_llama_config* cfg_out) {
    return load_gguf_impl(path, max_batch, max_blocks_per_seq, block_size, kv_layout, max_seq, 0, 1, model_out, cfg_out);
}

extern "C" int mi355_llama_check_gguf(const char* path, int32_t tp_rank, int32_t tp_world, mi355_llama_config* cfg_out) {
    return load_gguf_impl(path, 1, 1, 16, MI355_KV_PAGED, 0, tp_rank, tp_world, nullptr, cfg_out, true);
}

extern "C" int mi355_llama_load_gguf_tp(const char* path, int32_t max_batch, int32_t max_blocks_per_seq, int32_t block_size,
                                        int32_t kv_layout, int32_t max_seq, int32_t tp_rank, int32_t tp_world,
                                        void** model_out, mi355_llama_config* cfg_out) {
    return load_gguf_impl(path, max_batch, max_blocks_per_seq, block_size, kv_layout, max_seq, tp_rank, tp_world, model_out,
                          cfg_out);
}

// ---- tensor-parallel communicator -----------------------------------------------------------------------------
// rank 0 makes the id (ncclGetUniqueId), the launcher ships the 128 bytes to the other ranks (the reference
// passes it through the DAEMON_PAYLOAD env / TCP, communicator.rs:761-769,877), every rank calls init.
extern "C" int mi355_llama_init_comm(void* mp, const void* id128) {
    Model* m = static_cast<Model*>(mp);
    if (!m || !id128) return (int)hipErrorInvalidValue;
    void* c = mi355_comm_create(id128, m->cfg.tp_rank, m->cfg.tp_world);
    if (!c) return (int)hipErrorSharedObjectInitFailed;
    if (m->comm && m->comm_owned) mi355_comm_destroy(m->comm);
    m->comm = static_cast<Comm*>(c);
    m->comm_owned = true;
    return 0;
}
extern "C" void* mi355_llama_comm_handle(void* mp) {
    Model* m = static_cast<Model*>(mp);
    return m ? static_cast<void*>(m->comm) : nullptr;
}
// attach a communicator the caller owns (mi355_comm_create / mi355_comm_create_external)
extern "C" int mi355_llama_set_comm(void* mp, void* comm) {
    Model* m = static_cast<Model*>(mp);
    if (!m) return (int)hipErrorInvalidValue;
    if (m->comm && m->comm_owned) mi355_comm_destroy(m->comm);
    m->comm = static_cast<Comm*>(comm);
    m->comm_owned = false;
    return 0;
}

// ---- per-part launch for measurement: runs launch group `part` of layer `layer` on the static step inputs
extern "C" int mi355_llama_run_part(void* mp, int32_t layer, int32_t part, int64_t stream) {
    Model* m = static_cast<Model*>(mp);
    if (!m || m->cur_batch < 1 || part < PART_QKV || part > PART_EMBED) return (int)hipErrorInvalidValue;
    if (part <= PART_DOWN && (layer < 0 || layer >= m->cfg.n_layers)) return (int)hipErrorInvalidValue;
    const StepIn in{m->d_tokens, m->d_positions, m->d_slots, m->d_bt, m->d_ctx, m->cur_batch, m->cur_max_blocks, m->cur_ctx_cap,
                    m->xs, m->q, m->attn, m->h, m->moe_ids, m->moe_w, m->moe_y, m->tp_y, false, nullptr, 0, 0};
    return run_part(m, part <= PART_DOWN ? layer : 0, part, in, m->logits, stream);
}

extern "C" void* mi355_llama_act_ptr(void* mp, int32_t which) {
    Model* m = static_cast<Model*>(mp);
    if (!m) return nullptr;
    return which == 0 ? (void*)m->xs : which == 1 ? (void*)m->q : which == 2 ? (void*)m->attn : which == 3 ? (void*)m->h : nullptr;
}
