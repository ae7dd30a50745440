// Host layer for the safetensors (16-bit) llama-family models: `Llama::forward_inner` (src/openai/models/llama.rs:
// 139-201), `Block::forward` (:46-63), `Attention::forward_ext` (layers/attention.rs:585-734), `Mlp::forward`
// (layers/mlp.rs:440-458, packed gate_up :324-352), with Qwen2's qkv bias (qwen.rs).  Every op result is rounded to
// the model dtype exactly where candle rounds (the residual stream is 16-bit here, unlike the GGUF path).
// Kernels: dense_gemv.hip (linears + fused residual / silu*up), elementwise.hip (rms_norm, rope on 16-bit data
// = upcast-rotate-downcast), cache_kernels.hip (reshape_and_cache), paged_attention.hip / prefill_attention.hip.
// Step driver (round 4): `mi355_dense_decode_begin / _step / _read_tokens` -- the greedy loop on static device buffers with the
// device-side input advance and argmax of the GGUF layer, one hipGraph per (batch, table width, ctx bucket): what
// src/backend/graph.rs:471-661,685-807 + pipelines/pipeline.rs:2091-2135 do for every model family.  Tensor parallel through
// mi355_dense_set_comm (captured when the communicator is device-native).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <array>
#include <map>
#include <set>
#include <unordered_set>
#include <vector>

#include "../../include/mi355_vllm.h"
#include "common.h"
#include "comm.h"
#include "step_inputs.h"

extern "C" int mi355_internal_rope_cache(void* q, void* k, const void* v, void* key_cache, void* value_cache, const float* cos_t,
                                         const float* sin_t, const int64_t* positions, const int64_t* slot_mapping,
                                         int32_t num_tokens, int32_t num_heads, int32_t num_kv_heads, int32_t head_dim,
                                         int32_t rotary_dim, int32_t is_rope_i, int32_t block_size, int32_t layout, int32_t dtype,
                                         int64_t stream);
extern "C" int mi355_internal_linear3(void* const* outs, const void* x, const void* const* ws, const void* const* scales,
                                      const void* const* biases, const int32_t* ns, int32_t num_tokens, int32_t k,
                                      int32_t group_size, int32_t is_gptq, int32_t dtype, const void* norm_w, float norm_eps,
                                      const float* ss_in, const void* rope, int64_t stream);
// mirror of dense_gemv.hip's DenseRope: RoPE + KV-cache write in the epilogue of the q/k/v launch
struct DenseRope {
    const float* cos_t; const float* sin_t;
    const int64_t* positions; const int64_t* slots;
    uint16_t* kcache; uint16_t* vcache;
    int32_t n_kv_heads, head_dim, block_size, flash;
};
extern "C" int mi355_internal_gptq_small_linear(void* out, const void* x, const void* qweight_tiled, const void* scales, const void* bias,
                                                const void* residual, const void* norm_w, float norm_eps, const float* ss_in, float* ss_out,
                                                int32_t num_tokens, int32_t n, int32_t k, int32_t group_size, int32_t dtype,
                                                int32_t epilogue, int64_t stream);

extern "C" int mi355_pa_stream_auto(int32_t num_seqs, int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, int32_t block_size);   // paged_attention.hip
extern "C" int mi355_host_get_partition_override();   // host_model.cpp: mi355_set_tuning(5, partition_size), experiments

namespace {
int g_dense_tile = 1;          // tuning key 42: 0 = the 16-bit host layer keeps its projections row-major (A/B; read at the first step)

#define DCHECK(expr) do { const int rc_ = (expr); if (rc_ != 0) return rc_; } while (0)
#define DHIP(expr) do { const hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

// GPTQ (4-bit, symmetric, group scales) variant of a projection: qweight (checkpoint [K/8, N] u32, linear.rs:226-252) re-ordered at
// load time into 16-column x 256-k tiles (mi355_gptq_tile_repack) + scales [K/g, N] 16-bit in checkpoint layout; gate and up
// are concatenated along N for the fused silu*up epilogue.
struct QLin { uint32_t* qw = nullptr; uint16_t* scales = nullptr; int group = 0; };

struct DLayer {
    uint16_t *wq = nullptr, *wk = nullptr, *wv = nullptr, *wo = nullptr, *gate_up = nullptr, *w2 = nullptr;
    QLin gq[7];                // indexed by MI355_W_WQ .. MI355_W_W3 (W1 holds the packed gate_up, W3 unused)
    uint16_t *bq = nullptr, *bk = nullptr, *bv = nullptr;
    uint16_t *attn_norm = nullptr, *ffn_norm = nullptr, *attn_norm_b = nullptr, *ffn_norm_b = nullptr;
};

struct DModel {
    mi355_dense_config cfg{};
    std::vector<DLayer> layers;
    uint16_t *tok_embd = nullptr, *output_norm = nullptr, *output_norm_b = nullptr, *output = nullptr;
    float *cos_t = nullptr, *sin_t = nullptr;
    // activations (grow-only, T rows)
    int cap = 0;
    uint16_t *xs = nullptr, *xn = nullptr, *q = nullptr, *k = nullptr, *v = nullptr, *attn = nullptr, *h = nullptr, *lg16 = nullptr;
    float *pa_tmp = nullptr, *pa_max = nullptr, *pa_sum = nullptr;
    // 16-bit projections re-ordered into 16-row x 256-k tiles (mi355_dense_tile_repack) before the first step: the pointers in here
    std::unordered_set<const void*> tiled;
    bool finalized = false;
    float* ss = nullptr;                  // [4][hidden / 16] sums of squares of the residual stream's rows, per 16-column tile: left by
                                          // the 1..4-token 4-bit launch that wrote xs, read by the next one that norms it
    int pa_cap_partitions = 0;
    void* kv_slab = nullptr;
    std::vector<void*> kcache, vcache;
    int num_blocks = 0;
    void* comm = nullptr;                 // borrowed communicator (tp_world > 1, or a 1-rank plumbing test)
    uint16_t* lg_gather = nullptr;        // [W, B, V/W]
    // greedy decode loop on static device buffers (mi355_dense_decode_*): step inputs, sampled tokens, f32 logits
    uint32_t *d_tokens = nullptr, *d_ctx = nullptr, *d_bt = nullptr, *d_next = nullptr;
    int64_t *d_positions = nullptr, *d_slots = nullptr;
    float* d_logits = nullptr;
    int cur_batch = 0, cur_max_blocks = 0, cur_ctx_cap = 0, cur_ctx_max = 0;
    // one hipGraph of the greedy step per (batch, table width, ctx bucket), as the GGUF layer keeps them (graph.rs:471-661)
    struct StepGraph { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; uint64_t used = 0; };
    std::map<std::array<int, 3>, StepGraph> graphs;
    std::set<std::array<int, 3>> warmed;                  // shapes whose first (EAGER) step ran
    uint64_t graph_clock = 0;
    int64_t graph_captures = 0, eager_steps = 0;      // mi355_dense_graph_captures / _eager_steps
    bool use_graph = true;
    // test hook (mi355_dense_set_layer_window): run layers [win_first, win_last] only, from a supplied residual stream
    int win_first = -1, win_last = -1;
    const void* win_in = nullptr;
    void* win_out = nullptr;
};

// captured steps hold raw pointers: whatever re-allocates a buffer a step touches drops them first
void dense_drop_graph(DModel* m) {
    for (auto& kv : m->graphs) {
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    m->graphs.clear();
    m->warmed.clear();
}

// [W, B, Vl] -> [B, V]   (VocabParallelLinear: all-gather, un-interleave, narrow to the real vocabulary V <= W * Vl -- the columns
// beyond it are the zero rows of a padded vocabulary, distributed.rs:1637-1663)
__global__ void gather_transpose16_kernel(uint16_t* out, const uint16_t* in, int W, int B, int Vl, int V) {
    const int64_t n = (int64_t)W * B * Vl;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % Vl), b = (int)((i / Vl) % B), w = (int)(i / ((int64_t)Vl * B));
        const int64_t col = (int64_t)w * Vl + v;
        if (col < V) out[(int64_t)b * V + col] = in[i];
    }
}
// the width of a logits row: the real vocabulary under tensor parallelism (cfg.vocab is the rank's shard, zero rows of a padded
// vocabulary included), else the lm_head's rows
inline int logits_width(const mi355_dense_config& c) {
    const int W = c.tp_world > 1 ? c.tp_world : 1;
    return (W > 1 && c.vocab_total > 0) ? c.vocab_total : c.vocab * W;
}

// xs = round(xs + y): the block's residual add after the row-parallel all-reduce (llama.rs:55-58 over
// TensorParallelRowLinear::forward, distributed.rs:696-711)
__global__ void add16_kernel(uint16_t* xs, const uint16_t* y, int64_t n, int is_bf16) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (is_bf16) xs[i] = f32_to_bf16(bf16_to_f32(xs[i]) + bf16_to_f32(y[i]));
        else xs[i] = f32_to_f16_bits(f16_bits_to_f32(xs[i]) + f16_bits_to_f32(y[i]));
    }
}

__global__ void embedding16_kernel(uint16_t* __restrict__ out, const uint16_t* __restrict__ table,
                                   const uint32_t* __restrict__ ids, int hidden) {
    const uint16_t* src = table + (size_t)ids[blockIdx.x] * hidden;
    uint16_t* dst = out + (size_t)blockIdx.x * hidden;
    for (int i = threadIdx.x; i < hidden; i += blockDim.x) dst[i] = src[i];
}
__global__ void select_last_rows16_kernel(uint16_t* dst, const uint16_t* src, const uint32_t* cu_q, int hidden) {
    const size_t row = (size_t)cu_q[blockIdx.x + 1] - 1;
    for (int i = threadIdx.x; i < hidden; i += blockDim.x) dst[(size_t)blockIdx.x * hidden + i] = src[row * hidden + i];
}

int ensure_cap(DModel* m, int T) {
    if (T <= m->cap) return 0;
    DHIP(hipDeviceSynchronize());
    dense_drop_graph(m);
    void* old[] = {m->xs, m->xn, m->q, m->k, m->v, m->attn, m->h, m->lg16};
    for (void* p : old) if (p) (void)hipFree(p);
    m->xs = m->xn = m->q = m->k = m->v = m->attn = m->h = m->lg16 = nullptr;
    const mi355_dense_config& c = m->cfg;
    const int cap = (T + 63) / 64 * 64;
    const size_t hid = c.hidden, HD = (size_t)c.n_heads * c.head_dim, KD = (size_t)c.n_kv_heads * c.head_dim;
    DHIP(hipMalloc((void**)&m->xs, cap * hid * 2));
    DHIP(hipMalloc((void**)&m->xn, cap * hid * 2));
    DHIP(hipMalloc((void**)&m->q, cap * HD * 2));
    DHIP(hipMalloc((void**)&m->k, cap * KD * 2));
    DHIP(hipMalloc((void**)&m->v, cap * KD * 2));
    DHIP(hipMalloc((void**)&m->attn, cap * HD * 2));
    DHIP(hipMalloc((void**)&m->h, (size_t)cap * c.intermediate * 2));
    const int W = c.tp_world > 1 ? c.tp_world : 1;
    DHIP(hipMalloc((void**)&m->lg16, (size_t)c.max_batch * c.vocab * W * 2));
    if (m->lg_gather) (void)hipFree(m->lg_gather);
    DHIP(hipMalloc((void**)&m->lg_gather, (size_t)c.max_batch * c.vocab * W * 2));
    m->cap = cap;
    return 0;
}

// NormX::forward (layers/others.rs:11-34): RMSNorm, or LayerNorm for StableLM
int norm(const DModel* m, uint16_t* out, const uint16_t* x, const uint16_t* w, const uint16_t* b, int T, int64_t stream) {
    const mi355_dense_config& c = m->cfg;
    if (c.norm_type == 1) return mi355_layer_norm(out, x, w, b, T, c.hidden, c.rms_eps, c.dtype, stream);
    return mi355_rms_norm(out, x, w, T, c.hidden, c.rms_eps, c.dtype, c.dtype, stream);
}

// Linear::forward or the QLinear GPTQ arm (linear.rs:124-172 / 854-906), same fused epilogues
int linear(const DModel* m, const uint16_t* w, const QLin& g, void* out, const void* x, const void* bias, const void* resid,
           int T, int n, int k, int epi, int64_t stream) {
    if (g.qw)
        return mi355_gptq_linear_tiled(out, x, g.qw, g.scales, nullptr, MI355_ZERO_SYM8, 0, bias, resid, T, n, k, g.group, m->cfg.dtype, epi, stream);
    if (!w) return (int)hipErrorInvalidValue;
    if (m->tiled.count(w)) return mi355_linear_tiled(out, x, w, bias, resid, T, n, k, m->cfg.dtype, epi, stream);
    return mi355_linear(out, x, w, bias, resid, T, n, k, m->cfg.dtype, epi, stream);
}

// Before the first step: every 16-bit projection whose shape allows it (n % 16 == 0, k % 256 == 0) is re-ordered into the tile
// image the streaming kernels and the prompt GEMM read as contiguous KiB (row-major rows bound the weight stream at 3.9 TB/s:
// DESIGN.md section 4).  q, k and v together or not at all (they share one launch).  Weights are frozen from here on.
int tile_one(DModel* m, uint16_t** w, int n, int k) {
    if (!*w || m->tiled.count(*w) || (n & 15) || (k & 255)) return 0;
    uint16_t* t = nullptr;
    DHIP(hipMalloc((void**)&t, (size_t)n * k * 2));
    int rc = mi355_dense_tile_repack(*w, t, n, k, k, 0);
    if (!rc) rc = (int)hipDeviceSynchronize();
    if (rc) { (void)hipFree(t); return rc; }
    (void)hipFree(*w);
    *w = t;
    m->tiled.insert(t);
    return 0;
}
int finalize_weights(DModel* m) {
    if (m->finalized) return 0;
    const mi355_dense_config& c = m->cfg;
    const int hid = c.hidden, HD = c.n_heads * c.head_dim, KD = c.n_kv_heads * c.head_dim, I = c.intermediate;
    if (!g_dense_tile) { m->finalized = true; return 0; }
    for (auto& L : m->layers) {
        if (L.wq && L.wk && L.wv && !(HD & 15) && !(KD & 15) && !(hid & 255)) {
            DCHECK(tile_one(m, &L.wq, HD, hid)); DCHECK(tile_one(m, &L.wk, KD, hid)); DCHECK(tile_one(m, &L.wv, KD, hid));
        }
        DCHECK(tile_one(m, &L.wo, hid, HD));
        DCHECK(tile_one(m, &L.gate_up, 2 * I, hid));
        DCHECK(tile_one(m, &L.w2, hid, I));
    }
    DCHECK(tile_one(m, &m->output, c.vocab, hid));
    m->finalized = true;
    return 0;
}

// xs += x . w^T (the residual epilogue of o_proj / down_proj).  With 4-bit weights and 1..4 tokens the launch also leaves the sums
// of squares of the new xs rows in m->ss (*ss_valid), so the consumer's RmsNorm needs no launch of its own.
int resid_linear(const DModel* m, const uint16_t* w, const QLin& g, const void* x, int T, int n, int k, bool* ss_valid, int64_t stream) {
    *ss_valid = false;
    if (g.qw && T <= 4 && m->cfg.norm_type == 0) {
        const int rc = mi355_internal_gptq_small_linear(m->xs, x, g.qw, g.scales, nullptr, m->xs, nullptr, 0.f, nullptr, m->ss, T, n, k,
                                                        g.group, m->cfg.dtype, MI355_EPI_RESID, stream);
        if (rc == 0) { *ss_valid = true; return 0; }
        if (rc != -4) return rc;
    }
    return linear(m, w, g, m->xs, x, nullptr, m->xs, T, n, k, MI355_EPI_RESID, stream);
}

int choose_partition(int batch, int kv_heads, int ctx_cap) {
    if (ctx_cap <= 256) return 0;
    int per_seq = (2048 + batch * kv_heads - 1) / (batch * kv_heads);
    if (per_seq < 1) per_seq = 1;
    int ps = (ctx_cap + per_seq - 1) / per_seq;
    ps = ((ps + 31) / 32) * 32;
    if (ps < 32) ps = 32;
    return ps < ctx_cap ? ps : 0;
}

}  // namespace

extern "C" {

void mi355_dense_set_tile(int v) { g_dense_tile = v; }

void* mi355_dense_create(const mi355_dense_config* cfg) {
    if (!cfg || cfg->hidden <= 0 || (cfg->hidden % 256) || (cfg->intermediate % 256) || cfg->n_layers <= 0 ||
        cfg->max_batch <= 0 || cfg->head_dim <= 0 || (cfg->dtype != MI355_DTYPE_BF16 && cfg->dtype != MI355_DTYPE_F16))
        return nullptr;
    if (cfg->dtype != MI355_DTYPE_BF16) return nullptr;       // the attention kernels of this path are bf16
    if (cfg->kv_fp8 && (cfg->kv_layout != MI355_KV_PAGED || (cfg->head_dim % 16))) return nullptr;
    // vocab_total (the real vocabulary a tensor-parallel lm_head's gathered logits are narrowed to): inside the gathered row
    if (cfg->vocab_total < 0 || (cfg->vocab_total > 0 && (int64_t)cfg->vocab_total > (int64_t)cfg->vocab * (cfg->tp_world > 1 ? cfg->tp_world : 1)))
        return nullptr;
    DModel* m = new DModel();
    m->cfg = *cfg;
    m->layers.resize(cfg->n_layers);
    if (m->cfg.rotary_dim <= 0 || m->cfg.rotary_dim > cfg->head_dim) m->cfg.rotary_dim = cfg->head_dim;
    const int D = cfg->head_dim, rot = m->cfg.rotary_dim, half = rot / 2;
    std::vector<float> ct((size_t)cfg->max_seq * half), st((size_t)cfg->max_seq * half);
    const bool tables_ok = mi355_rope_tables(ct.data(), st.data(), rot, cfg->max_seq, (double)cfg->rope_theta, nullptr, cfg->max_seq, 0) == 0;
    bool ok = tables_ok && hipMalloc((void**)&m->cos_t, ct.size() * 4) == hipSuccess && hipMalloc((void**)&m->sin_t, st.size() * 4) == hipSuccess;
    ok = ok && hipMemcpy(m->cos_t, ct.data(), ct.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(m->sin_t, st.data(), st.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    m->pa_cap_partitions = (cfg->max_seq + 31) / 32 + 1;
    const size_t B = cfg->max_batch, H = cfg->n_heads;
    ok = ok && hipMalloc((void**)&m->pa_tmp, B * H * m->pa_cap_partitions * D * 4) == hipSuccess &&
         hipMalloc((void**)&m->pa_max, B * H * m->pa_cap_partitions * 4) == hipSuccess &&
         hipMalloc((void**)&m->pa_sum, B * H * m->pa_cap_partitions * 4) == hipSuccess &&
         hipMalloc((void**)&m->ss, (size_t)4 * (cfg->hidden / 16) * 4) == hipSuccess;
    if (!ok) { mi355_dense_destroy(m); return nullptr; }
    return m;
}

void mi355_dense_destroy(void* mp) {
    DModel* m = static_cast<DModel*>(mp);
    if (!m) return;
    for (auto& L : m->layers) {
        void* ps[] = {L.wq, L.wk, L.wv, L.wo, L.gate_up, L.w2, L.bq, L.bk, L.bv, L.attn_norm, L.ffn_norm, L.attn_norm_b, L.ffn_norm_b};
        for (void* p : ps) if (p) (void)hipFree(p);
        for (auto& g : L.gq) { if (g.qw) (void)hipFree(g.qw); if (g.scales) (void)hipFree(g.scales); }
    }
    for (auto& kv : m->graphs) {
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    void* ps[] = {m->tok_embd, m->output_norm, m->output_norm_b, m->output, m->cos_t, m->sin_t, m->xs, m->xn, m->q, m->k, m->v, m->attn,
                  m->h, m->lg16, m->pa_tmp, m->pa_max, m->pa_sum, m->ss, m->kv_slab, m->lg_gather, m->d_tokens, m->d_ctx, m->d_bt,
                  m->d_next, m->d_positions, m->d_slots, m->d_logits};
    for (void* p : ps) if (p) (void)hipFree(p);
    delete m;
}

/* 16-bit tensors from the HOST in checkpoint layout ([out, in] row-major); gate (W1) and up (W3) are packed into one
 * [2I, hidden] matrix as the reference's unquantised MLP does (mlp.rs:324-352). */
static int dense_set_weight_impl(void* mp, int32_t layer, int32_t which, const void* host, int64_t n_elems, hipMemcpyKind kind);
int mi355_dense_set_weight(void* mp, int32_t layer, int32_t which, const void* host, int64_t n_elems) {
    return dense_set_weight_impl(mp, layer, which, host, n_elems, hipMemcpyHostToDevice);
}
/* same, from a DEVICE buffer (copied) */
int mi355_dense_set_weight_dev(void* mp, int32_t layer, int32_t which, const void* dev, int64_t n_elems) {
    return dense_set_weight_impl(mp, layer, which, dev, n_elems, hipMemcpyDeviceToDevice);
}
static int dense_set_weight_impl(void* mp, int32_t layer, int32_t which, const void* host, int64_t n_elems, hipMemcpyKind kind) {
    DModel* m = static_cast<DModel*>(mp);
    if (!m || !host || n_elems <= 0) return (int)hipErrorInvalidValue;
    const mi355_dense_config& c = m->cfg;
    const int64_t hid = c.hidden, HD = (int64_t)c.n_heads * c.head_dim, KD = (int64_t)c.n_kv_heads * c.head_dim, I = c.intermediate;
    uint16_t** slot = nullptr;
    int64_t expect = 0, offset = 0, total = 0;
    if (layer < 0) {
        // the embedding table is replicated under TP (full vocabulary), lm_head is vocab-parallel (distributed.rs:1632)
        if (which == MI355_W_TOK_EMBD) { slot = &m->tok_embd; expect = (int64_t)logits_width(c) * hid; }
        else if (which == MI355_W_OUTPUT_NORM) { slot = &m->output_norm; expect = hid; }
        else if (which == MI355_W_OUTPUT) { slot = &m->output; expect = (int64_t)c.vocab * hid; }
        else if (which == MI355_W_OUTPUT_NORM_B) { slot = &m->output_norm_b; expect = hid; }
        else return (int)hipErrorInvalidValue;
    } else {
        if (layer >= c.n_layers) return (int)hipErrorInvalidValue;
        DLayer& L = m->layers[layer];
        switch (which) {
            case MI355_W_WQ: slot = &L.wq; expect = HD * hid; break;
            case MI355_W_WK: slot = &L.wk; expect = KD * hid; break;
            case MI355_W_WV: slot = &L.wv; expect = KD * hid; break;
            case MI355_W_WO: slot = &L.wo; expect = hid * HD; break;
            case MI355_W_W1: slot = &L.gate_up; expect = I * hid; total = 2 * I * hid; offset = 0; break;
            case MI355_W_W3: slot = &L.gate_up; expect = I * hid; total = 2 * I * hid; offset = I * hid; break;
            case MI355_W_W2: slot = &L.w2; expect = hid * I; break;
            case MI355_W_ATTN_NORM: slot = &L.attn_norm; expect = hid; break;
            case MI355_W_FFN_NORM: slot = &L.ffn_norm; expect = hid; break;
            case MI355_W_BQ: slot = &L.bq; expect = HD; break;
            case MI355_W_BK: slot = &L.bk; expect = KD; break;
            case MI355_W_BV: slot = &L.bv; expect = KD; break;
            case MI355_W_ATTN_NORM_B: slot = &L.attn_norm_b; expect = hid; break;
            case MI355_W_FFN_NORM_B: slot = &L.ffn_norm_b; expect = hid; break;
            default: return (int)hipErrorInvalidValue;
        }
    }
    if (n_elems != expect) return (int)hipErrorInvalidValue;
    if (m->finalized) return (int)hipErrorInvalidValue;      // mi355_dense_finalize / the first step froze the weights: every slot alike
    if (total == 0) total = expect;
    if (!*slot) DHIP(hipMalloc((void**)slot, (size_t)total * 2));
    DHIP(hipMemcpy(*slot + offset, host, (size_t)n_elems * 2, kind));
    return 0;
}

/* GPTQ projection from HOST checkpoint tensors: qweight u32 [k/8, n], scales 16-bit [k/g, n] (natural order, sym
 * 4-bit, no act-order: the Marlin-eligible case of linear.rs:319-325).  gate_proj (W1) and up_proj (W3) are
 * concatenated along n so the silu*up epilogue stays fused. */
int mi355_dense_set_gptq(void* mp, int32_t layer, int32_t which, const void* qweight_host, const void* scales_host,
                         int32_t n, int32_t k, int32_t group_size) {
    DModel* m = static_cast<DModel*>(mp);
    if (!m || layer < 0 || layer >= m->cfg.n_layers || !qweight_host || !scales_host || m->finalized) return (int)hipErrorInvalidValue;
    if (which < MI355_W_WQ || which > MI355_W_W3 || (k % 256) || (n % 16)) return (int)hipErrorInvalidValue;
    const int g = (group_size <= 0 || group_size > k) ? k : group_size;
    const int I = m->cfg.intermediate;
    const bool gu = which == MI355_W_W1 || which == MI355_W_W3;
    QLin& q = m->layers[layer].gq[gu ? MI355_W_W1 : which];
    const int ntot = gu ? 2 * I : n, col0 = which == MI355_W_W3 ? I : 0;
    if (gu && n != I) return (int)hipErrorInvalidValue;
    if (q.group && q.group != g) return (int)hipErrorInvalidValue;
    q.group = g;
    if (!q.qw) DHIP(hipMalloc((void**)&q.qw, (size_t)(k / 8) * ntot * 4));
    if (!q.scales) DHIP(hipMalloc((void**)&q.scales, (size_t)(k / g) * ntot * 2));
    {   // checkpoint layout -> staging buffer -> tiles [col0 / 16, (col0 + n) / 16) of the tiled image
        uint32_t* stage = nullptr;
        DHIP(hipMalloc((void**)&stage, (size_t)(k / 8) * n * 4));
        int rc = (int)hipMemcpy(stage, qweight_host, (size_t)(k / 8) * n * 4, hipMemcpyHostToDevice);
        if (!rc) rc = mi355_gptq_tile_repack(stage, q.qw, k, n, col0 / 16, 0);
        if (!rc) rc = (int)hipDeviceSynchronize();
        (void)hipFree(stage);
        if (rc) return rc;
    }
    DHIP(hipMemcpy2D(q.scales + col0, (size_t)ntot * 2, scales_host, (size_t)n * 2, (size_t)n * 2, k / g, hipMemcpyHostToDevice));
    return 0;
}

int mi355_dense_set_rope_tables(void* mp, const float* cos_host, const float* sin_host, int32_t n_positions) {
    DModel* m = static_cast<DModel*>(mp);
    if (!m || !cos_host || !sin_host || n_positions < m->cfg.max_seq) return (int)hipErrorInvalidValue;
    const size_t bytes = (size_t)n_positions * (m->cfg.rotary_dim / 2) * 4;
    float *c = nullptr, *s = nullptr;
    dense_drop_graph(m);
    DHIP(hipMalloc((void**)&c, bytes));
    DHIP(hipMalloc((void**)&s, bytes));
    DHIP(hipMemcpy(c, cos_host, bytes, hipMemcpyHostToDevice));
    DHIP(hipMemcpy(s, sin_host, bytes, hipMemcpyHostToDevice));
    DHIP(hipDeviceSynchronize());
    (void)hipFree(m->cos_t); (void)hipFree(m->sin_t);
    m->cos_t = c; m->sin_t = s;
    return 0;
}

int mi355_dense_set_comm(void* mp, void* comm) {
    DModel* m = static_cast<DModel*>(mp);
    if (!m) return (int)hipErrorInvalidValue;
    dense_drop_graph(m);
    m->comm = comm;
    return 0;
}

int mi355_dense_alloc_kv_cache(void* mp, int32_t num_blocks) {
    DModel* m = static_cast<DModel*>(mp);
    if (!m || num_blocks <= 0) return (int)hipErrorInvalidValue;
    const mi355_dense_config& c = m->cfg;
    const size_t per = (size_t)num_blocks * c.block_size * c.n_kv_heads * c.head_dim * (c.kv_fp8 ? 1 : 2);
    dense_drop_graph(m);
    if (m->kv_slab) { (void)hipFree(m->kv_slab); m->kv_slab = nullptr; }
    DHIP(hipMalloc(&m->kv_slab, per * 2 * c.n_layers));
    DHIP(hipMemset(m->kv_slab, 0, per * 2 * c.n_layers));
    DHIP(hipDeviceSynchronize());
    m->kcache.resize(c.n_layers); m->vcache.resize(c.n_layers);
    for (int l = 0; l < c.n_layers; ++l) {
        m->kcache[l] = static_cast<uint8_t*>(m->kv_slab) + per * (2 * l);
        m->vcache[l] = static_cast<uint8_t*>(m->kv_slab) + per * (2 * l + 1);
    }
    m->num_blocks = num_blocks;
    return 0;
}
/* Test hook for the full-size parity legs (tests/fullsize_dense.py): the next mi355_dense_forward calls skip the embedding and
 * the head, start from the 16-bit residual stream xs_in_dev [num_tokens, hidden], run layers first..last (inclusive) exactly as
 * the whole forward runs them, and copy the stream after the last of them to xs_out_dev.  first < 0 switches the window off. */
int mi355_dense_set_layer_window(void* mp, int32_t first, int32_t last, const void* xs_in_dev, void* xs_out_dev) {
    DModel* m = static_cast<DModel*>(mp);
    if (!m) return (int)hipErrorInvalidValue;
    if (first < 0) { m->win_first = m->win_last = -1; m->win_in = nullptr; m->win_out = nullptr; return 0; }
    if (last < first || last >= m->cfg.n_layers || !xs_in_dev || !xs_out_dev) return (int)hipErrorInvalidValue;
    m->win_first = first; m->win_last = last; m->win_in = xs_in_dev; m->win_out = xs_out_dev;
    return 0;
}
void* mi355_dense_kv_ptr(void* mp, int32_t layer, int32_t which) {
    DModel* m = static_cast<DModel*>(mp);
    if (!m || layer < 0 || layer >= (int)m->kcache.size()) return nullptr;
    return which == 0 ? m->kcache[layer] : m->vcache[layer];
}

/* One step (prompt when cu_seqlens_q != NULL, else decode).  Inputs are DEVICE arrays exactly as prepare_prompt /
 * prepare_decode build them.  logits: f32 [num_seqs, vocab] (`.to_dtype(F32)` of the 16-bit lm_head output). */
int mi355_dense_forward(void* mp, const uint32_t* tokens, const int64_t* positions, const int64_t* slot_mapping,
                        const uint32_t* block_tables, const uint32_t* context_lens, const uint32_t* cu_seqlens_q,
                        int32_t num_seqs, int32_t num_tokens, int32_t max_seqlen_q, int32_t max_blocks,
                        int32_t max_context_len, float* logits, int64_t stream) {
    DModel* m = static_cast<DModel*>(mp);
    if (!m || !logits || num_seqs < 1 || num_tokens < num_seqs || num_seqs > m->cfg.max_batch) return (int)hipErrorInvalidValue;
    const mi355_dense_config& c = m->cfg;
    if ((int)m->kcache.size() != c.n_layers || !m->tok_embd || !m->output || !m->output_norm) return (int)hipErrorInvalidValue;
    const bool prefill = cu_seqlens_q != nullptr;
    if (!prefill && num_tokens != num_seqs) return (int)hipErrorInvalidValue;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (num_tokens > m->cap || !m->finalized) {
        // growing the workspace and the one-off weight repack allocate, synchronise and free: never inside a stream capture
        // (loaders call mi355_dense_finalize, the step driver finalizes in decode_begin)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (stream != 0 && hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return (int)hipErrorStreamCaptureUnsupported;
    }
    DCHECK(ensure_cap(m, num_tokens));
    DCHECK(finalize_weights(m));
    const int T = num_tokens, H = c.n_heads, Hkv = c.n_kv_heads, D = c.head_dim, hid = c.hidden, I = c.intermediate;
    const int dt = c.dtype;
    const float scale = 1.0f / sqrtf((float)D);
    const bool window = m->win_first >= 0;
    if (window) DHIP(hipMemcpyAsync(m->xs, m->win_in, (size_t)T * hid * 2, hipMemcpyDeviceToDevice, st));
    else hipLaunchKernelGGL(embedding16_kernel, dim3(T), dim3(256), 0, st, m->xs, m->tok_embd, tokens, hid);
    bool ss_valid = false;                    // m->ss holds the sums of squares of the current xs rows
    for (int l = window ? m->win_first : 0; l <= (window ? m->win_last : c.n_layers - 1); ++l) {
        DLayer& L = m->layers[l];
        if (!L.attn_norm || !L.ffn_norm) return (int)hipErrorInvalidValue;
        // x = rms_1(xs) (llama.rs:53-54), then the q,k,v projections (+bias) (attention.rs:597-607)
        // one launch for the three projections when they share the weight format (decode-sized steps); with 4-bit weights
        // and 1..4 tokens that launch also applies the RmsNorm while it stages the activations (no norm launch)
        int rc3 = -4;
        bool normed = false, roped = false;
        {
            const QLin &gq = L.gq[MI355_W_WQ], &gk = L.gq[MI355_W_WK], &gv = L.gq[MI355_W_WV];
            const bool all_q = gq.qw && gk.qw && gv.qw && gq.group == gk.group && gq.group == gv.group;
            const bool all_d = !gq.qw && !gk.qw && !gv.qw && L.wq && L.wk && L.wv;
            if (T <= 64 && (all_q || all_d)) {
                void* outs[3] = {m->q, m->k, m->v};
                const void* ws[3] = {all_q ? (const void*)gq.qw : L.wq, all_q ? (const void*)gk.qw : L.wk, all_q ? (const void*)gv.qw : L.wv};
                const void* sc[3] = {gq.scales, gk.scales, gv.scales};
                const void* bs[3] = {L.bq, L.bk, L.bv};
                const int32_t ns[3] = {H * D, Hkv * D, Hkv * D};
                // a decode step of 4-bit weights on a bf16 cache with full, non-interleaved rotary: RoPE and the cache write ride in the
                // q/k/v launch's epilogue (-4 from the launch = not this shape: the launch is repeated without them)
                DenseRope rp{m->cos_t, m->sin_t, positions, slot_mapping, (uint16_t*)m->kcache[l], (uint16_t*)m->vcache[l], Hkv, D,
                             c.block_size, c.kv_layout == MI355_KV_FLASH ? 1 : 0};
                // (1..4 tokens: the 4-bit small kernel; 5..32 tokens: the row-tile-pair form of the batch kernel, 4-bit and 16-bit weights)
                const bool rope_ok = (all_q ? T <= 32 : (T > 4 && T <= 32)) && !prefill && !c.kv_fp8 && c.rotary_dim == D && !c.rope_interleaved;
                if (all_q && ss_valid && c.norm_type == 0) {            // the producer of xs left its sums of squares: no norm launch
                    for (int pass = rope_ok ? 0 : 1; pass < 2 && rc3 != 0; ++pass) {
                        rc3 = mi355_internal_linear3(outs, m->xs, ws, sc, bs, ns, T, hid, gq.group, 2, m->cfg.dtype, L.attn_norm, c.rms_eps, m->ss,
                                                     pass == 0 ? &rp : nullptr, stream);
                        if (rc3 != 0 && rc3 != -4) return rc3;
                        roped = rc3 == 0 && pass == 0;
                    }
                }
                if (rc3 != 0) {
                    DCHECK(norm(m, m->xn, m->xs, L.attn_norm, L.attn_norm_b, T, stream));
                    normed = true;
                    for (int pass = rope_ok ? 0 : 1; pass < 2 && rc3 != 0; ++pass) {
                        rc3 = mi355_internal_linear3(outs, m->xn, ws, sc, bs, ns, T, hid, gq.group, all_q ? 2 : (m->tiled.count(L.wq) ? 3 : 0), m->cfg.dtype, nullptr, 0.f, nullptr,
                                                     pass == 0 ? &rp : nullptr, stream);
                        if (rc3 != 0 && rc3 != -4) return rc3;
                        roped = rc3 == 0 && pass == 0;
                    }
                }
            }
        }
        if (rc3 != 0 && !normed) DCHECK(norm(m, m->xn, m->xs, L.attn_norm, L.attn_norm_b, T, stream));
        if (rc3 != 0) {
        DCHECK(linear(m, L.wq, L.gq[MI355_W_WQ], m->q, m->xn, L.bq, nullptr, T, H * D, hid, MI355_EPI_STORE, stream));
        DCHECK(linear(m, L.wk, L.gq[MI355_W_WK], m->k, m->xn, L.bk, nullptr, T, Hkv * D, hid, MI355_EPI_STORE, stream));
        DCHECK(linear(m, L.wv, L.gq[MI355_W_WV], m->v, m->xn, L.bv, nullptr, T, Hkv * D, hid, MI355_EPI_STORE, stream));
        }
        // q,k -> f32 -> rope -> model dtype                                  attention.rs:644-690
        // (decode steps without an fp8 cache: RoPE and the cache write share one launch)
        int rc_rc = roped ? 0 : -4;
        if (!roped && !c.kv_fp8 && !prefill)
            rc_rc = mi355_internal_rope_cache(m->q, m->k, m->v, m->kcache[l], m->vcache[l], m->cos_t, m->sin_t, positions, slot_mapping,
                                              T, H, Hkv, D, c.rotary_dim, c.rope_interleaved, c.block_size, c.kv_layout, dt, stream);
        if (rc_rc != 0 && rc_rc != -4) return rc_rc;
        const bool fused_rc = rc_rc == 0;
        if (!fused_rc)
        DCHECK(mi355_rope_inplace(m->q, m->k, m->cos_t, m->sin_t, positions, T, H, Hkv, D, c.rotary_dim, c.rope_interleaved, dt, stream));
        // PagedAttention::forward: cache write, then prefill / decode attention   attention.rs:707-719
        if (c.kv_fp8) {
            // `--kvcache-dtype fp8` (is_fp8_keys, attention.rs:574): e4m3fn cache, every key read back from it
            DCHECK(mi355_reshape_and_cache_fp8(m->k, m->v, m->kcache[l], m->vcache[l], slot_mapping, T, Hkv, D, c.block_size,
                                               c.kv_layout, 1.f, 1.f, stream));
            if (prefill) {
                DCHECK(mi355_prefill_attention_fp8(m->attn, m->q, m->kcache[l], m->vcache[l], block_tables, context_lens,
                                                   cu_seqlens_q, num_seqs, max_seqlen_q, H, Hkv, D, c.block_size, max_blocks,
                                                   scale, 0.f, 1.f, 1.f, dt, stream));
            } else {
                int ps = choose_partition(T, Hkv, max_context_len);
                if (ps > 0) ps = 32;   // MFMA kernel: 32-token partitions, 4 per workgroup (measured at batch 32: 32 -> 4937, 64 -> 4784, 128 -> ~4500 tok/s)
                if (ps > 0 && mi355_pa_stream_auto(T, H, Hkv, D, c.block_size)) ps = 64;   // the balanced LDS-DMA stream over the e4m3fn cache, as the GGUF driver
                if (ps > 0 && mi355_host_get_partition_override() > 0) ps = mi355_host_get_partition_override();
                if (ps > 0 && (max_context_len + ps - 1) / ps > m->pa_cap_partitions) return (int)hipErrorInvalidValue;
                DCHECK(mi355_paged_attention_fp8(m->attn, m->pa_sum, m->pa_max, m->pa_tmp, m->q, m->kcache[l], m->vcache[l],
                                                 block_tables, context_lens, T, H, Hkv, D, c.block_size, max_blocks,
                                                 max_context_len, ps, scale, 0.f, 1.f, 1.f, stream));
            }
        } else {
        if (!fused_rc)
        DCHECK(mi355_reshape_and_cache(m->k, m->v, m->kcache[l], m->vcache[l], slot_mapping, T, Hkv, D, c.block_size, 2,
                                       c.kv_layout, stream));
        if (prefill) {
            DCHECK(mi355_prefill_attention(m->attn, m->q, nullptr, nullptr, m->kcache[l], m->vcache[l], block_tables,
                                           context_lens, cu_seqlens_q, num_seqs, max_seqlen_q, H, Hkv, D, c.block_size,
                                           max_blocks, scale, 0.f, c.kv_layout, dt, stream));
        } else {
            int ps = choose_partition(T, Hkv, max_context_len);
            if (ps > 0 && c.kv_layout == MI355_KV_PAGED) ps = 32;   // MFMA kernel: 32-token partitions, 4 per workgroup (measured at batch 32: 32 -> 4937, 64 -> 4784, 128 -> ~4500 tok/s)
            if (ps > 0 && c.kv_layout == MI355_KV_PAGED && dt == MI355_DTYPE_BF16 && mi355_pa_stream_auto(T, H, Hkv, D, c.block_size)) ps = 64;   // the balanced LDS-DMA stream, as the GGUF driver
            if (ps > 0 && c.kv_layout == MI355_KV_PAGED && mi355_host_get_partition_override() > 0) ps = mi355_host_get_partition_override();   // experiments: mi355_set_tuning(5, n), as the GGUF driver
            if (ps > 0 && (max_context_len + ps - 1) / ps > m->pa_cap_partitions) return (int)hipErrorInvalidValue;
            if (ps == 0)
                DCHECK(mi355_paged_attention_v1(m->attn, m->q, m->kcache[l], m->vcache[l], block_tables, context_lens, T, H,
                                                Hkv, D, c.block_size, max_blocks, max_context_len, scale, 0.f, c.kv_layout, dt, stream));
            else
                DCHECK(mi355_paged_attention_v2(m->attn, m->pa_sum, m->pa_max, m->pa_tmp, m->q, m->kcache[l], m->vcache[l],
                                                block_tables, context_lens, T, H, Hkv, D, c.block_size, max_blocks,
                                                max_context_len, ps, scale, 0.f, c.kv_layout, dt, stream));
        }
        }
        // xs = o_proj(y) + residual                                          llama.rs:55-58
        // TP (TensorParallelRowLinear, distributed.rs:696-711): the rounded partial products are all-reduced in the
        // model dtype, then the block adds the residual -- the same three rounding points as the reference
        const bool tp = m->comm != nullptr;
        const int add_grid = (int)std::min<int64_t>(((int64_t)T * hid + 255) / 256, 2048);
        if (tp) {
            DCHECK(linear(m, L.wo, L.gq[MI355_W_WO], m->xn, m->attn, nullptr, nullptr, T, hid, H * D, MI355_EPI_STORE, stream));
            DCHECK(mi355_comm_all_reduce(m->comm, m->xn, (int64_t)T * hid, dt, stream));
            hipLaunchKernelGGL(add16_kernel, dim3(add_grid), dim3(256), 0, st, m->xs, m->xn, (int64_t)T * hid, dt == MI355_DTYPE_BF16);
        } else {
            DCHECK(resid_linear(m, L.wo, L.gq[MI355_W_WO], m->attn, T, hid, H * D, &ss_valid, stream));
        }
        // xs = down(silu(gate) * up) + residual                              llama.rs:59-61, mlp.rs:440-458
        int rc_gu = -4;
        if (L.gq[MI355_W_W1].qw && ss_valid && c.norm_type == 0) {       // wo left the sums of squares of xs: the RmsNorm rides in the gate/up launch
            const QLin& g = L.gq[MI355_W_W1];
            rc_gu = mi355_internal_gptq_small_linear(m->h, m->xs, g.qw, g.scales, nullptr, nullptr, L.ffn_norm, c.rms_eps, m->ss, nullptr,
                                                     T, 2 * I, hid, g.group, dt, MI355_EPI_SILU_MUL, stream);
            if (rc_gu != 0 && rc_gu != -4) return rc_gu;
        }
        if (rc_gu != 0) {
        DCHECK(norm(m, m->xn, m->xs, L.ffn_norm, L.ffn_norm_b, T, stream));
        DCHECK(linear(m, L.gate_up, L.gq[MI355_W_W1], m->h, m->xn, nullptr, nullptr, T, 2 * I, hid, MI355_EPI_SILU_MUL, stream));
        }
        if (tp) {
            DCHECK(linear(m, L.w2, L.gq[MI355_W_W2], m->xn, m->h, nullptr, nullptr, T, hid, I, MI355_EPI_STORE, stream));
            DCHECK(mi355_comm_all_reduce(m->comm, m->xn, (int64_t)T * hid, dt, stream));
            hipLaunchKernelGGL(add16_kernel, dim3(add_grid), dim3(256), 0, st, m->xs, m->xn, (int64_t)T * hid, dt == MI355_DTYPE_BF16);
        } else {
            DCHECK(resid_linear(m, L.w2, L.gq[MI355_W_W2], m->h, T, hid, I, &ss_valid, stream));
        }
    }
    if (window) {                                                           // the residual stream after the window's last layer
        DHIP(hipMemcpyAsync(m->win_out, m->xs, (size_t)T * hid * 2, hipMemcpyDeviceToDevice, st));
        return 0;
    }
    const uint16_t* last = m->xs;
    if (prefill) {                                                          // llama.rs:190-194
        hipLaunchKernelGGL(select_last_rows16_kernel, dim3(num_seqs), dim3(256), 0, st, m->xn, m->xs, cu_seqlens_q, hid);
        DHIP(hipMemcpyAsync(m->xs, m->xn, (size_t)num_seqs * hid * 2, hipMemcpyDeviceToDevice, st));
        last = m->xs;
    }
    DCHECK(norm(m, m->xn, last, m->output_norm, m->output_norm_b, num_seqs, stream));
    if (m->comm) {                                                          // VocabParallelLinear (distributed.rs:1632-1667)
        const int W = c.tp_world > 1 ? c.tp_world : 1;
        DCHECK(linear(m, m->output, QLin{}, m->lg16, m->xn, nullptr, nullptr, num_seqs, c.vocab, hid, MI355_EPI_STORE, stream));
        DCHECK(mi355_comm_all_gather(m->comm, m->lg16, m->lg_gather, (int64_t)num_seqs * c.vocab, dt, stream));
        // narrowed to the real vocabulary as VocabParallelLinear::forward does before sampling (the greedy loop sampled
        // over the padded row, where a zero row beats an all-negative row and the token id leaves the embedding table)
        hipLaunchKernelGGL(gather_transpose16_kernel, dim3(512), dim3(256), 0, st, m->lg16, m->lg_gather, W, num_seqs, c.vocab, logits_width(c));
        return mi355_cast(logits, m->lg16, (int64_t)num_seqs * logits_width(c), dt, MI355_DTYPE_F32, stream);
    }
    DCHECK(linear(m, m->output, QLin{}, m->lg16, m->xn, nullptr, nullptr, num_seqs, c.vocab, hid, MI355_EPI_STORE, stream));
    return mi355_cast(logits, m->lg16, (int64_t)num_seqs * c.vocab, dt, MI355_DTYPE_F32, stream);
}

/* Freeze the weights: the one-off re-ordering of the 16-bit projections into tiles (mi355_dense_tile_repack) that the first forward
 * would otherwise do.  Loaders call it once after the last set_weight; afterwards set_weight refuses EVERY projection slot. */
int mi355_dense_finalize(void* mp) {
    DModel* m = static_cast<DModel*>(mp);
    if (!m) return (int)hipErrorInvalidValue;
    return finalize_weights(m);
}

/* ---- greedy decode loop on static device buffers (graph replay): the GGUF layer's mi355_llama_decode_* for this host layer ---- */
int mi355_dense_set_graph(void* mp, int32_t enable) {
    DModel* m = static_cast<DModel*>(mp);
    if (!m) return (int)hipErrorInvalidValue;
    m->use_graph = enable != 0;
    if (!m->use_graph) dense_drop_graph(m);
    m->warmed.clear();
    return 0;
}
int mi355_dense_decode_begin(void* mp, const uint32_t* tokens_host, const uint32_t* seq_lens_host, const uint32_t* block_tables_host,
                             int32_t batch, int32_t max_blocks, int32_t ctx_cap, int64_t stream) {
    DModel* m = static_cast<DModel*>(mp);
    if (!m || !tokens_host || !seq_lens_host || !block_tables_host || batch < 1 || batch > m->cfg.max_batch || max_blocks < 1 ||
        m->cfg.max_blocks_per_seq < 1 || max_blocks > m->cfg.max_blocks_per_seq)
        return (int)hipErrorInvalidValue;
    const mi355_dense_config& c = m->cfg;
    const int bs = c.block_size, W = c.tp_world > 1 ? c.tp_world : 1;
    std::vector<int64_t> pos(batch), slot(batch);
    int ctx_max = 0;
    for (int b = 0; b < batch; ++b) {
        if (seq_lens_host[b] < 1) return (int)hipErrorInvalidValue;
        pos[b] = (int64_t)seq_lens_host[b] - 1;
        if (pos[b] / bs >= max_blocks) return (int)hipErrorInvalidValue;          // "Block table is too small"
        slot[b] = (int64_t)block_tables_host[(size_t)b * max_blocks + pos[b] / bs] * bs + pos[b] % bs;
        ctx_max = std::max(ctx_max, (int)seq_lens_host[b]);
    }
    if (ctx_cap < ctx_max || ctx_cap > c.max_seq || ctx_max > max_blocks * bs) return (int)hipErrorInvalidValue;
    if (!m->d_tokens) {
        const size_t B = c.max_batch;
        DHIP(hipMalloc((void**)&m->d_tokens, B * 4)); DHIP(hipMalloc((void**)&m->d_ctx, B * 4)); DHIP(hipMalloc((void**)&m->d_next, B * 4));
        DHIP(hipMalloc((void**)&m->d_bt, B * c.max_blocks_per_seq * 4));
        DHIP(hipMalloc((void**)&m->d_positions, B * 8)); DHIP(hipMalloc((void**)&m->d_slots, B * 8));
        DHIP(hipMalloc((void**)&m->d_logits, B * (size_t)c.vocab * W * 4));
    }
    DCHECK(ensure_cap(m, batch));                          // everything that allocates happens here, never inside a captured step
    DCHECK(finalize_weights(m));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    DHIP(hipMemcpyAsync(m->d_tokens, tokens_host, (size_t)batch * 4, hipMemcpyHostToDevice, st));
    DHIP(hipMemcpyAsync(m->d_ctx, seq_lens_host, (size_t)batch * 4, hipMemcpyHostToDevice, st));
    DHIP(hipMemcpyAsync(m->d_bt, block_tables_host, (size_t)batch * max_blocks * 4, hipMemcpyHostToDevice, st));
    DHIP(hipMemcpyAsync(m->d_positions, pos.data(), (size_t)batch * 8, hipMemcpyHostToDevice, st));
    DHIP(hipMemcpyAsync(m->d_slots, slot.data(), (size_t)batch * 8, hipMemcpyHostToDevice, st));
    DHIP(hipStreamSynchronize(st));                        // host vectors go out of scope
    m->cur_batch = batch; m->cur_max_blocks = max_blocks; m->cur_ctx_cap = ctx_cap; m->cur_ctx_max = ctx_max;
    return 0;
}
static int dense_record_step(DModel* m, int64_t stream) {
    const int B = m->cur_batch;
    DCHECK(mi355_dense_forward(m, m->d_tokens, m->d_positions, m->d_slots, m->d_bt, m->d_ctx, nullptr, B, B, 0, m->cur_max_blocks,
                               m->cur_ctx_cap, m->d_logits, stream));
    // greedy sample (`sample_argmax`, logits_processor.rs:92-95: first maximum), then the next step's inputs on the device
    DCHECK(mi355_argmax_f32(m->d_next, m->d_logits, B, logits_width(m->cfg), stream));
    hipLaunchKernelGGL(advance_kernel, dim3((B + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), m->d_tokens, m->d_next,
                       m->d_positions, m->d_slots, m->d_ctx, m->d_bt, m->cur_max_blocks, m->cfg.block_size, B);
    return (int)hipGetLastError();
}
/* One greedy step: forward -> argmax -> next-step inputs; captured once per (batch, max_blocks, ctx_cap) and replayed. */
int mi355_dense_decode_step(void* mp, int64_t stream) {
    DModel* m = static_cast<DModel*>(mp);
    if (!m || m->cur_batch < 1) return (int)hipErrorInvalidValue;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // this step attends over cur_ctx_max tokens, then advance_kernel looks up the slot of position cur_ctx_max: both must stay
    // inside the attention grid (ctx_cap), the block-table row and the RoPE tables
    if (m->cur_ctx_max > m->cur_ctx_cap || m->cur_ctx_max > m->cfg.max_seq ||
        (m->cur_ctx_max + m->cfg.block_size - 1) / m->cfg.block_size > m->cur_max_blocks)
        return (int)hipErrorInvalidValue;
    // (the host mirror of the context length advances only when a step was actually enqueued: a refused or failed step leaves the
    // device-side context where it was, and a retry must pass the same checks again -- ADVICE r4)
    auto stepped = [m](int rc) { if (rc == 0) ++m->cur_ctx_max; return rc; };
    auto eager = [m, stream]() { ++m->eager_steps; return dense_record_step(m, stream); };
    // host-supplied collectives are host calls: such TP steps stay eager (a host call made during capture is not replayed)
    const Comm* cm = static_cast<const Comm*>(m->comm);
    const bool tp_eager = cm && (cm->ar || cm->ag || !cm->nccl);
    if (!m->use_graph || stream == 0 || tp_eager) return stepped(eager());
    const std::array<int, 3> shape{m->cur_batch, m->cur_max_blocks, m->cur_ctx_cap};
    if (!m->warmed.count(shape)) {
        // the first step of a new shape runs eagerly: lazily-set kernel attributes and scratch growth must not happen inside a capture
        m->warmed.insert(shape);
        return stepped(eager());
    }
    auto it = m->graphs.find(shape);
    if (it == m->graphs.end()) {
        if (m->graphs.size() >= 96) {                                 // evict the least recently replayed shape
            auto lru = m->graphs.begin();
            for (auto q = m->graphs.begin(); q != m->graphs.end(); ++q) if (q->second.used < lru->second.used) lru = q;
            if (lru->second.exec) (void)hipGraphExecDestroy(lru->second.exec);
            if (lru->second.graph) (void)hipGraphDestroy(lru->second.graph);
            m->graphs.erase(lru);
        }
        DHIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        const int rc = dense_record_step(m, stream);
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(st, &g);
        if (rc != 0) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (e != hipSuccess) return (int)e;
        DModel::StepGraph sg;
        sg.graph = g;
        const hipError_t ie = hipGraphInstantiate(&sg.exec, sg.graph, nullptr, nullptr, 0);
        if (ie != hipSuccess) { (void)hipGraphDestroy(g); return (int)ie; }
        it = m->graphs.emplace(shape, sg).first;
        ++m->graph_captures;
    }
    it->second.used = ++m->graph_clock;
    DHIP(hipGraphLaunch(it->second.exec, st));
    return stepped(0);
}
/* step graphs captured + instantiated so far / steps of the greedy loop that ran eagerly (see mi355_llama_graph_captures) */
int64_t mi355_dense_graph_captures(void* mp) {
    DModel* m = static_cast<DModel*>(mp);
    return m ? m->graph_captures : -1;
}
int64_t mi355_dense_eager_steps(void* mp) {
    DModel* m = static_cast<DModel*>(mp);
    return m ? m->eager_steps : -1;
}
/* D2H of the tokens the last step sampled (= the inputs of the next step); synchronises the stream */
int mi355_dense_decode_read_tokens(void* mp, uint32_t* host_out, int64_t stream) {
    DModel* m = static_cast<DModel*>(mp);
    if (!m || !host_out || m->cur_batch < 1) return (int)hipErrorInvalidValue;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    DHIP(hipMemcpyAsync(host_out, m->d_tokens, (size_t)m->cur_batch * 4, hipMemcpyDeviceToHost, st));
    uint32_t p2p_err = 0;
    const Comm* cm = static_cast<const Comm*>(m->comm);
    const bool p2p = cm && cm->p2p && cm->local;
    if (p2p) DHIP(hipMemcpyAsync(&p2p_err, &cm->local->err, 4, hipMemcpyDeviceToHost, st));
    DHIP(hipStreamSynchronize(st));
    if (p2p && p2p_err != 0) return (int)hipErrorPeerAccessNotEnabled;     // a peer missed the spin bound: the step is invalid on this rank
    return 0;
}
/* f32 [batch, vocab] logits of the last step of the loop (device pointer); tensor parallel: [batch, vocab_total] (the gathered row
 * narrowed to the real vocabulary; vocab x tp_world when vocab_total is 0) */
float* mi355_dense_logits_ptr(void* mp) { DModel* m = static_cast<DModel*>(mp); return m ? m->d_logits : nullptr; }

}  // extern "C"
