"""CPU oracle (TEST INFRASTRUCTURE ONLY) -- the 16-bit safetensors llama-family step restated in numpy with candle's
rounding points: every op result is rounded to the model dtype (bf16).
  Llama::forward_inner / Block::forward        src/openai/models/llama.rs:46-63,139-201
  Attention::forward_ext                       src/openai/models/layers/attention.rs:585-734 (q,k -> f32 -> rope -> dtype)
  Mlp::forward (packed gate_up)                src/openai/models/layers/mlp.rs:324-352,440-458
  Linear::forward                              src/openai/models/linear.rs:124-172
  Qwen2 qkv bias                               src/openai/models/qwen.rs
PARITY UNPINNED (no reference fixtures exist for this path)."""
from dataclasses import dataclass

import numpy as np

from . import gptq as G
from . import ops

DT = "bf16"


def R(a):
    """candle rounds every op result to the model dtype: src/openai/models/llama.rs:46-63."""
    return G.round_dt(a, DT)


@dataclass
class DenseConfig:
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
    qkv_bias: bool = False
    layer_norm: bool = False        # StableLM: LayerNorm with bias (stable_lm.rs:61-72)
    rotary_dim: int = 0             # 0 = head_dim; StableLM: 0.25 * head_dim (stable_lm.rs:28)
    kv_fp8: bool = False            # `--kvcache-dtype fp8`: e4m3fn cache, every key/value read back from it

    @staticmethod
    def stablelm_3b():
        """BASELINE configs[0] shapes: StableLM-3B-4e1t"""
        return DenseConfig(hidden=2560, n_layers=32, n_heads=32, n_kv_heads=32, head_dim=80, intermediate=6912,
                           vocab=50304, rms_eps=1e-5, rope_theta=10000.0, max_seq=4096, block_size=64,
                           layer_norm=True, rotary_dim=20)

    @staticmethod
    def tiny_stablelm():
        return DenseConfig(hidden=1280, n_layers=2, n_heads=16, n_kv_heads=16, head_dim=80, intermediate=512, vocab=512,
                           rope_theta=10000.0, max_seq=256, block_size=16, qkv_bias=True, layer_norm=True, rotary_dim=20)

    @staticmethod
    def tiny(qkv_bias=False):
        return DenseConfig(hidden=256, n_layers=2, n_heads=4, n_kv_heads=2, head_dim=64, intermediate=512, vocab=512,
                           rope_theta=10000.0, max_seq=256, block_size=16, qkv_bias=qkv_bias)


def make_weights(cfg, seed=4321, std=0.05):
    """tensor set of src/openai/models/llama.rs:203-260 (synthetic values)."""
    rng = np.random.default_rng(seed)
    H, Hkv, D, hid, I = cfg.n_heads, cfg.n_kv_heads, cfg.head_dim, cfg.hidden, cfg.intermediate

    def w(*shape, s=std):
        return R(rng.normal(0.0, s, size=shape).astype(np.float32))
    W = {"tok_embd": w(cfg.vocab, hid, s=0.5), "layers": []}
    for _ in range(cfg.n_layers):
        lw = {"attn_norm": R(1.0 + rng.normal(0, 0.05, hid)), "ffn_norm": R(1.0 + rng.normal(0, 0.05, hid)),
              "wq": w(H * D, hid), "wk": w(Hkv * D, hid), "wv": w(Hkv * D, hid), "wo": w(hid, H * D),
              "w1": w(I, hid), "w3": w(I, hid), "w2": w(hid, I)}
        if cfg.qkv_bias:
            lw.update({"bq": w(H * D, s=0.1), "bk": w(Hkv * D, s=0.1), "bv": w(Hkv * D, s=0.1)})
        if cfg.layer_norm:
            lw.update({"attn_norm_b": w(hid, s=0.05), "ffn_norm_b": w(hid, s=0.05)})
        W["layers"].append(lw)
    if cfg.layer_norm:
        W["output_norm_b"] = w(hid, s=0.05)
    W["output_norm"] = R(1.0 + rng.normal(0, 0.05, hid))
    W["output"] = w(cfg.vocab, hid)
    return W


def quantize_gptq(W, group=128, seed=7):
    """replace every projection by an RTN 4-bit symmetric GPTQ tensor (bits=4, sym, desc_act=false: the
    Marlin-eligible case, linear.rs:319-325).  The dict keeps qweight / scales for the device and
    the dequantised [out, in] matrix the oracle multiplies with."""
    out = {k: v for k, v in W.items() if k != "layers"}
    out["layers"] = []
    for lw in W["layers"]:
        nl = dict(lw)
        for name in ("wq", "wk", "wv", "wo", "w1", "w2", "w3"):
            w = np.asarray(lw[name], np.float32).T                      # [k, n]
            K, N = w.shape
            g = group if 0 < group < K else K
            wg = w.reshape(K // g, g, N)
            s = R(np.maximum(np.abs(wg).max(1), 1e-8) / 7.0)            # scales in the model dtype
            q = np.clip(np.rint(wg / s[:, None, :]) + 8, 0, 15).astype(np.int64).reshape(K, N)
            deq = G.gptq_dequant(q, s, None, g)                         # f64 [k, n]
            nl[name] = {"qweight": G.gptq_pack(q), "scales": s, "group": g, "deq": deq}
        out["layers"].append(nl)
    return out


def _lin(x, w, b=None):
    """Linear::forward or the QLinear GPTQ arm (linear.rs:124-172 / 854-906)"""
    if isinstance(w, dict):
        if "deq" not in w:                                              # full-size legs: dequantise on first use, keep (shared by the layers)
            w["deq"] = G.gptq_dequant(G.gptq_unpack(w["qweight"]).astype(np.int64), w["scales"], None, w["group"])
        return G.gptq_linear(x, w["deq"], b, DT)
    return G.linear16(x, w, b, DT)


def rms_norm16(x, w, eps):
    """candle rms_norm on 16-bit data: f32 internally, result rounded (layers/others.rs NormX).
    candle_nn::RmsNorm in the model dtype: src/openai/distributed.rs:1164-1205, src/openai/models/llama.rs:53-54."""
    x = np.asarray(x, np.float32)
    inv = 1.0 / np.sqrt((x.astype(np.float64) ** 2).mean(-1, keepdims=True) + eps)
    return R(x * inv * np.asarray(w, np.float32))


def layer_norm16(x, w, b, eps):
    """candle_nn LayerNorm on 16-bit data: f32 statistics and affine, one rounding [EXT fused path]."""
    x = np.asarray(x, np.float64)
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    y = (x - mu) / np.sqrt(var + eps) * np.asarray(w, np.float64)
    if b is not None:
        y = y + np.asarray(b, np.float64)
    return R(y.astype(np.float32))


class OracleDenseLlama:
    def __init__(self, cfg, W, flash_layout=True, comm=None, rope_scaling=None, max_position_embeddings=0):
        """comm: None, or an object with all_reduce(np.ndarray)->np.ndarray and all_gather(np.ndarray)->list (tensor
        parallel; cfg / W are then this rank's shard).  The collectives run in the model dtype: the reduced sum is
        rounded (distributed.rs:696-711), the gathered logits are exact (distributed.rs:1637-1663)."""
        self.cfg, self.W, self.flash, self.comm = cfg, W, flash_layout, comm
        self.rot = cfg.rotary_dim or cfg.head_dim
        # rope_scaling: the reference's `rope_scaling` dict (ScalingRotaryEmbedding::new, rotary_emb.rs:107-341)
        self.cos, self.sin = ops.rope_tables_scaled(cfg.rope_theta, self.rot, cfg.max_seq, rope_scaling, max_position_embeddings)
        self.scale = 1.0 / np.sqrt(float(cfg.head_dim))

    def _row_lin(self, x, w, resid):
        """o_proj / down_proj + residual; under TP: round(partial) -> all-reduce in dtype -> + residual"""
        p = _lin(x, w)
        if self.comm is not None:
            p = R(self.comm.all_reduce(p))
        return R(p + resid)

    def _norm(self, x, w, b):
        if self.cfg.layer_norm:
            return layer_norm16(x, w, b, self.cfg.rms_eps)
        return rms_norm16(x, w, self.cfg.rms_eps)

    def new_cache(self, num_blocks):
        """src/scheduler/cache_engine.rs:298-341 (shapes), :304-311 for the fp8 cache."""
        c = self.cfg
        if c.kv_fp8:
            ks, vs = ops.kv_cache_shapes(num_blocks, c.block_size, c.n_kv_heads, c.head_dim, 1, False)
            return [(np.zeros(ks, np.uint8), np.zeros(vs, np.uint8)) for _ in range(c.n_layers)]
        ks, vs = ops.kv_cache_shapes(num_blocks, c.block_size, c.n_kv_heads, c.head_dim, 2, self.flash)
        return [(np.zeros(ks, np.uint16), np.zeros(vs, np.uint16)) for _ in range(c.n_layers)]

    def _attend_fp8(self, meta, q, k, v, kc, vc, is_prefill):
        """fp8 cache: write e4m3fn, then every key (prefix and current tokens) comes back dequantised."""
        c = self.cfg
        ops.reshape_and_cache_fp8(k, v, kc, vc, meta["slot_mapping"], False)
        kb, vb = ops.fp8_cache_as_bf16_bits(kc, vc, False)
        if not is_prefill:
            return ops.paged_attention_decode(q, kb, vb, meta["block_tables"], meta["context_lens"], self.scale, False)
        ys, cu = [], meta["cu_seqlens_q"]
        for i in range(len(cu) - 1):
            a, b = int(cu[i]), int(cu[i + 1])
            n = int(meta["context_lens"][i])
            table = meta["block_tables"][i]
            kk, vv = ops.gather_kv(kb, vb, table, n, False)
            ys.append(ops.prefill_attention(q[a:b], ops.bf16_bits_to_f32(kk), ops.bf16_bits_to_f32(vv), self.scale, cached=n - (b - a)))
        return np.concatenate(ys, 0)

    def forward(self, meta, kv_caches, is_prefill=False, trace=None, trace_mid=None):
        """Llama::forward_inner src/openai/models/llama.rs:139-201 with Attention::forward_ext src/openai/models/layers/attention.rs:585-734.
        trace: a list that receives the residual stream at every layer entry and after the last layer (full-size parity legs);
        trace_mid: a list that receives the stream between the two branches of every layer (after o_proj + residual)."""
        c, W = self.cfg, self.W
        toks, pos = meta["input_ids"], meta["positions"]
        T = len(toks)
        xs = W["tok_embd"][toks].astype(np.float32)
        for l, lw in enumerate(W["layers"]):
            if trace is not None:
                trace.append(xs.copy())
            x = self._norm(xs, lw["attn_norm"], lw.get("attn_norm_b"))
            q = _lin(x, lw["wq"], lw.get("bq")).reshape(T, c.n_heads, c.head_dim)
            k = _lin(x, lw["wk"], lw.get("bk")).reshape(T, c.n_kv_heads, c.head_dim)
            v = _lin(x, lw["wv"], lw.get("bv")).reshape(T, c.n_kv_heads, c.head_dim)
            q = R(ops.rope_apply(q, self.cos, self.sin, pos, interleaved=False, rotary_dim=self.rot))   # f32 rope, back to dtype
            k = R(ops.rope_apply(k, self.cos, self.sin, pos, interleaved=False, rotary_dim=self.rot))
            kc, vc = kv_caches[l]
            if c.kv_fp8:
                y = self._attend_fp8(meta, q, k, v, kc, vc, is_prefill)
                y = y.reshape(T, c.n_heads * c.head_dim)
                xs = self._row_lin(y, lw["wo"], xs)
                if trace_mid is not None:
                    trace_mid.append(xs.copy())
                x = self._norm(xs, lw["ffn_norm"], lw.get("ffn_norm_b"))
                gate, up = _lin(x, lw["w1"]), _lin(x, lw["w3"])
                xs = self._row_lin(G.silu_mul16(gate, up, DT), lw["w2"], xs)
                continue
            kb, vb = ops.f32_to_bf16_bits(k), ops.f32_to_bf16_bits(v)
            ops.reshape_and_cache(kb, vb, kc, vc, meta["slot_mapping"], self.flash)
            if is_prefill:
                ys, cu = [], meta["cu_seqlens_q"]
                for i in range(len(cu) - 1):
                    a, b = int(cu[i]), int(cu[i + 1])
                    ys.append(ops.prefill_attention(q[a:b], k[a:b], v[a:b], self.scale))
                y = np.concatenate(ys, 0)
            else:
                y = ops.paged_attention_decode(q, kc, vc, meta["block_tables"], meta["context_lens"], self.scale, self.flash)
            y = y.reshape(T, c.n_heads * c.head_dim)
            xs = self._row_lin(y, lw["wo"], xs)
            if trace_mid is not None:
                trace_mid.append(xs.copy())
            x = self._norm(xs, lw["ffn_norm"], lw.get("ffn_norm_b"))
            gate, up = _lin(x, lw["w1"]), _lin(x, lw["w3"])
            h = G.silu_mul16(gate, up, DT)
            xs = self._row_lin(h, lw["w2"], xs)
        if trace is not None:
            trace.append(xs.copy())
        if is_prefill:
            xs = xs[np.asarray(meta["cu_seqlens_q"][1:], np.int64) - 1]
        xs = self._norm(xs, W["output_norm"], W.get("output_norm_b"))
        logits = G.linear16(xs, W["output"], None, DT).astype(np.float32)
        if self.comm is not None:
            logits = np.concatenate(self.comm.all_gather(logits), axis=-1)
        return logits
