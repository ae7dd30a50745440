"""CPU oracle (TEST INFRASTRUCTURE ONLY) -- GGUFLLaMa decode/prefill step restated in numpy.

Follows the reference's op order and dtype casts:
  layer loop / residuals / final norm / lm_head      src/openai/models/quantized_llama.rs:424-506
  attention block (f32 activations, bf16 attention)  src/openai/models/layers/attention.rs:910-1011
  MLP  w2(silu(w1 x) * w3 x)                         src/openai/models/quantized_llama.rs:30-45
  interleaved RoPE (is_gpt_neox = false)             src/openai/models/quantized_llama.rs:313-318
PARITY UNPINNED (no reference fixtures exist for this path).
Quantised mat-vec uses oracle O1 (dequantise -> f64 dot) unless o2=True.
"""
from dataclasses import dataclass
import numpy as np

from . import kquants as kq
from . import ops


@dataclass
class LlamaConfig:
    hidden: int = 4096
    n_layers: int = 32
    n_heads: int = 32
    n_kv_heads: int = 8
    head_dim: int = 128
    intermediate: int = 14336
    vocab: int = 128256
    rms_eps: float = 1e-5
    rope_theta: float = 500000.0
    max_seq: int = 8192
    block_size: int = 64

    @staticmethod
    def llama3_8b():
        """dims read from GGUF metadata at src/openai/models/quantized_llama.rs:231-260 (public model-card values)."""
        return LlamaConfig()

    @staticmethod
    def tiny(vocab=512, hidden=256, n_layers=2, n_heads=4, n_kv_heads=2, head_dim=64,
             intermediate=512, max_seq=256, block_size=16):
        return LlamaConfig(hidden=hidden, n_layers=n_layers, n_heads=n_heads, n_kv_heads=n_kv_heads,
                           head_dim=head_dim, intermediate=intermediate, vocab=vocab, max_seq=max_seq,
                           block_size=block_size, rope_theta=10000.0)


def q4km_type_for(name, layer, n_layers):
    """llama.cpp Q4_K_M mixture [EXT]: output.weight Q6_K; attn_v and ffn_down Q6_K when
    use_more_bits(layer): layer < n/8 or layer >= 7n/8 or (layer - n/8) % 3 == 2; rest Q4_K."""
    if name == "output":
        return kq.GGML_Q6_K
    if name in ("wv", "w2"):
        n8 = n_layers // 8
        if layer < n8 or layer >= 7 * n8 or (layer - n8) % 3 == 2:
            return kq.GGML_Q6_K
    return kq.GGML_Q4_K


def make_weights(cfg, seed=1234, recipe="q4_k_m", std=0.02):
    """Synthetic weights: N(0, std) f32 -> quantised."""
    rng = np.random.default_rng(seed)
    H, Hkv, D, hid, I = cfg.n_heads, cfg.n_kv_heads, cfg.head_dim, cfg.hidden, cfg.intermediate

    def qt(name, layer, rows, cols):
        w = rng.normal(0.0, std, size=(rows, cols)).astype(np.float32)
        t = q4km_type_for(name, layer, cfg.n_layers) if recipe == "q4_k_m" else kq.GGML_Q4_K
        if recipe != "q4_k_m" and name == "output":
            t = kq.GGML_Q6_K
        return (t, kq.quantize(w, t))

    W = {"tok_embd": rng.normal(0.0, std, size=(cfg.vocab, hid)).astype(np.float32), "layers": []}
    for l in range(cfg.n_layers):
        W["layers"].append({
            "attn_norm": (1.0 + rng.normal(0, 0.02, hid)).astype(np.float32),
            "wq": qt("wq", l, H * D, hid), "wk": qt("wk", l, Hkv * D, hid),
            "wv": qt("wv", l, Hkv * D, hid), "wo": qt("wo", l, hid, H * D),
            "ffn_norm": (1.0 + rng.normal(0, 0.02, hid)).astype(np.float32),
            "w1": qt("w1", l, I, hid), "w2": qt("w2", l, hid, I), "w3": qt("w3", l, I, hid),
        })
    W["output_norm"] = (1.0 + rng.normal(0, 0.02, hid)).astype(np.float32)
    W["output"] = qt("output", 0, cfg.vocab, hid)
    return W


def weight_bytes(cfg, W):
    """Algorithmic weight bytes per decode step (token_embd and norm vectors excluded)."""
    n = 0
    for lw in W["layers"]:
        for k in ("wq", "wk", "wv", "wo", "w1", "w2", "w3"):
            n += lw[k][1].size
    return n + W["output"][1].size


def _qmm(x, tw, o2):
    t, blocks = tw
    return kq.qmatmul_o2(x, blocks, t) if o2 else kq.qmatmul_o1(x, blocks, t)


def moe_route(x, gate_inp, top_k):
    """MlpOrMoe::forward routing (quantized_llama.rs:63-91): softmax over the router logits, top-k by a STABLE
    descending sort (ties -> lower expert id), weights renormalised over the selected experts.
    Returns ids [T,k] int32, weights [T,k] f32."""
    logits = np.asarray(x, np.float64) @ np.asarray(gate_inp, np.float64).T
    p = np.exp(logits - logits.max(-1, keepdims=True))
    p = (p / p.sum(-1, keepdims=True)).astype(np.float32)
    ids = np.argsort(-p, axis=-1, kind="stable")[:, :top_k].astype(np.int32)
    w = np.take_along_axis(p, ids, axis=-1)
    return ids, (w / w.sum(-1, keepdims=True)).astype(np.float32)


def moe_forward(x, lw, top_k, o2=False):
    """ys = sum_j w_j * expert_j(x) (quantized_llama.rs:93-119); lw["experts"] = list of {"w1","w2","w3"}."""
    ids, w = moe_route(x, lw["gate_inp"], top_k)
    ys = np.zeros_like(np.asarray(x, np.float32))
    for t in range(x.shape[0]):
        for j in range(top_k):
            ex = lw["experts"][int(ids[t, j])]
            h = ops.silu_mul(_qmm(x[t:t + 1], ex["w1"], o2), _qmm(x[t:t + 1], ex["w3"], o2))
            ys[t] += w[t, j] * _qmm(h, ex["w2"], o2)[0]
    return ys


def make_moe_weights(cfg, n_expert, seed=1234, std=0.02):
    """Mixtral-style synthetic weights: the dense MLP of every layer replaced by n_expert Q4_K experts + an F32 router
    (tensor names ffn_gate_inp / ffn_{gate,down,up}.{i}, quantized_llama.rs:347-365)."""
    W = make_weights(cfg, seed=seed, std=std)
    rng = np.random.default_rng(seed + 99)
    I, hid = cfg.intermediate, cfg.hidden
    for lw in W["layers"]:
        for k in ("w1", "w2", "w3"):
            lw.pop(k)
        lw["gate_inp"] = rng.normal(0.0, 0.5, size=(n_expert, hid)).astype(np.float32)
        lw["experts"] = []
        for _ in range(n_expert):
            def q(rows, cols):
                w = rng.normal(0.0, std, size=(rows, cols)).astype(np.float32)
                return (kq.GGML_Q4_K, kq.quantize(w, kq.GGML_Q4_K))
            lw["experts"].append({"w1": q(I, hid), "w2": q(hid, I), "w3": q(I, hid)})
    return W


class OracleLlama:
    def __init__(self, cfg, W, flash_layout=True, o2=False, comm=None):
        """comm: None, or an object with all_reduce(np.ndarray)->np.ndarray and all_gather(np.ndarray)->list
        (tensor-parallel run: cfg/W are then the LOCAL shard, see candle_vllm_amd/tp.py; collectives C1/C2/C3
        of src/openai/distributed.rs:696-711,1632-1667)."""
        self.cfg, self.W, self.flash, self.o2, self.comm = cfg, W, flash_layout, o2, comm
        self.moe_top_k = 2                                           # llama.expert_used_count (Mixtral: 2)
        self.kv_fp8 = False                                          # `--kvcache-dtype fp8`: e4m3fn cache (paged, x = 16)
        self.cos, self.sin = ops.rope_tables(cfg.rope_theta, cfg.head_dim, cfg.max_seq)
        self.scale = 1.0 / np.sqrt(float(cfg.head_dim))

    def new_cache(self, num_blocks):
        """src/scheduler/cache_engine.rs:298-341."""
        c = self.cfg
        if self.kv_fp8:
            ks, vs = ops.kv_cache_shapes(num_blocks, c.block_size, c.n_kv_heads, c.head_dim, 1, False)
            return [(np.zeros(ks, np.uint8), np.zeros(vs, np.uint8)) for _ in range(c.n_layers)]
        ks, vs = ops.kv_cache_shapes(num_blocks, c.block_size, c.n_kv_heads, c.head_dim, 2, self.flash)
        return [(np.zeros(ks, np.uint16), np.zeros(vs, np.uint16)) for _ in range(c.n_layers)]

    def _attend_fp8(self, meta, qb, k, v, kc, vc, is_prefill):
        """e4m3fn cache: the bf16-rounded k, v are quantised on write and every key comes back dequantised."""
        ops.reshape_and_cache_fp8(ops.round_bf16(k), ops.round_bf16(v), kc, vc, meta["slot_mapping"], False)
        kb, vb = ops.fp8_cache_as_bf16_bits(kc, vc, False)
        if not is_prefill:
            return ops.paged_attention_decode(qb, kb, vb, meta["block_tables"], meta["context_lens"], self.scale, False)
        ys, cu = [], meta["cu_seqlens_q"]
        for i in range(len(cu) - 1):
            a, b = int(cu[i]), int(cu[i + 1])
            n = int(meta["context_lens"][i])
            kk, vv = ops.gather_kv(kb, vb, meta["block_tables"][i], n, False)
            ys.append(ops.prefill_attention(qb[a:b], ops.bf16_bits_to_f32(kk), ops.bf16_bits_to_f32(vv), self.scale, cached=n - (b - a)))
        return np.concatenate(ys, 0)

    def forward(self, meta, kv_caches, is_prefill=False, trace=None):
        """meta: dict from ops.prepare_decode / prepare_prompt.  Returns logits f32 [B, V]."""
        c, W = self.cfg, self.W
        toks = meta["input_ids"]
        pos = meta["positions"]
        T = len(toks)
        xs = W["tok_embd"][toks].astype(np.float32)
        for l, lw in enumerate(W["layers"]):
            x = ops.rms_norm(xs, lw["attn_norm"], c.rms_eps)
            q = _qmm(x, lw["wq"], self.o2).reshape(T, c.n_heads, c.head_dim)
            k = _qmm(x, lw["wk"], self.o2).reshape(T, c.n_kv_heads, c.head_dim)
            v = _qmm(x, lw["wv"], self.o2).reshape(T, c.n_kv_heads, c.head_dim)
            q = ops.rope_apply(q, self.cos, self.sin, pos, interleaved=True)
            k = ops.rope_apply(k, self.cos, self.sin, pos, interleaved=True)
            qb, kb, vb = ops.round_bf16(q), ops.f32_to_bf16_bits(k), ops.f32_to_bf16_bits(v)
            kc, vc = kv_caches[l]
            if self.kv_fp8:
                y = self._attend_fp8(meta, qb, k, v, kc, vc, is_prefill)
            else:
                ops.reshape_and_cache(kb, vb, kc, vc, meta["slot_mapping"], self.flash)
            if self.kv_fp8:
                pass
            elif is_prefill:
                ys = []
                cu = meta["cu_seqlens_q"]
                for i in range(len(cu) - 1):
                    a, b = int(cu[i]), int(cu[i + 1])
                    ys.append(ops.prefill_attention(qb[a:b], ops.bf16_bits_to_f32(kb[a:b]),
                                                    ops.bf16_bits_to_f32(vb[a:b]), self.scale))
                y = np.concatenate(ys, 0)
            else:
                y = ops.paged_attention_decode(qb, kc, vc, meta["block_tables"], meta["context_lens"],
                                               self.scale, self.flash)
            y = y.reshape(T, c.n_heads * c.head_dim)
            attn = _qmm(y, lw["wo"], self.o2)
            if self.comm is not None:
                attn = self.comm.all_reduce(attn)                      # C1 (attention.rs:1005-1009)
            xs = attn + xs
            x = ops.rms_norm(xs, lw["ffn_norm"], c.rms_eps)
            if "experts" in lw:
                # experts are loaded UNSHARDED on every rank (`get_no_shape`, all_reduce: None --
                # quantized_llama.rs:344-365): the MoE block is replicated, only attention / lm_head are parallel
                mlp = moe_forward(x, lw, self.moe_top_k, self.o2)
            else:
                h = ops.silu_mul(_qmm(x, lw["w1"], self.o2), _qmm(x, lw["w3"], self.o2))
                mlp = _qmm(h, lw["w2"], self.o2)
                if self.comm is not None:
                    mlp = self.comm.all_reduce(mlp)                    # C2 (quantized_llama.rs:38-42)
            xs = mlp + xs
            if trace is not None:
                trace.append(xs.copy())
        if is_prefill:
            idx = np.asarray(meta["cu_seqlens_q"][1:], np.int64) - 1
            xs = xs[idx]
        xs = ops.rms_norm(xs, W["output_norm"], c.rms_eps)
        logits = _qmm(xs, W["output"], self.o2)
        if self.comm is not None:                                      # C3: vocab-parallel all-gather
            logits = np.concatenate(self.comm.all_gather(logits), axis=-1)
            total = getattr(self.cfg, "vocab_total", None)             # padded vocabulary: narrow back (distributed.rs:1657-1660)
            if total is not None and logits.shape[-1] > total:
                logits = logits[..., :total]
        return logits
