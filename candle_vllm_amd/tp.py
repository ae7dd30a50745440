"""Tensor-parallel shard plan of the GGUF llama weights and of the 16-bit / GPTQ safetensors weights (host side,
pure index arithmetic).

Follows src/openai/distributed.rs: q heads H/W, kv heads max(Hkv/W,1) with replication when Hkv < W
(`kv_head_shard`, :725-765); wq/wk/wv/w1/w3/output split on rows (`shard(0, rank, W)`, :804), wo/w2 split on
input columns (`shard(1, ...)`, :1154) -- for k-quants a column shard is a k-block range, which requires
cols/W to be a multiple of 256 (the reference re-quantises to Q8_0 otherwise,
quantized_var_builder.rs:234-269; not built)."""
import copy

import numpy as np


def kv_head_shard(total_kv_heads, rank, world):
    if total_kv_heads >= world:
        if total_kv_heads % world:
            raise ValueError("KV heads must be divisible by tensor parallel world_size when partitioned")
        return total_kv_heads // world, rank, world
    if world % total_kv_heads:
        raise ValueError("tensor parallel world_size must be divisible by KV heads when KV heads are replicated")
    return 1, rank // (world // total_kv_heads), total_kv_heads


def _rows(tw, rank, world):
    t, b = tw
    n = b.shape[0]
    if n % world:
        raise ValueError("rows not divisible by the shard count")
    c = n // world
    return (t, np.ascontiguousarray(b[rank * c:(rank + 1) * c]))


def _cols(tw, rank, world):
    t, b = tw
    nb = b.shape[1]
    if nb % world:
        raise ValueError("column shard is not k-block aligned (Q8_0 re-quantisation fallback not built)")
    c = nb // world
    return (t, np.ascontiguousarray(b[:, rank * c:(rank + 1) * c]))


def shard_config(cfg, rank, world):
    """local LlamaConfig (heads / kv heads / intermediate / vocab of this rank)"""
    if cfg.n_heads % world:
        raise ValueError("num_attention_heads must be divisible by world_size (attention.rs:553-554)")
    local = copy.copy(cfg)
    local.n_heads = cfg.n_heads // world
    local.n_kv_heads = kv_head_shard(cfg.n_kv_heads, rank, world)[0]
    local.intermediate = cfg.intermediate // world
    local.vocab = cfg.vocab // world
    return local


def shard_weights(W, cfg, rank, world):
    """W: oracle.llama.make_weights dict (global) -> this rank's dict.  tok_embd and the norm vectors are
    replicated."""
    _, kv_rank, kv_world = kv_head_shard(cfg.n_kv_heads, rank, world)
    out = {"tok_embd": W["tok_embd"], "output_norm": W["output_norm"], "output": _rows(W["output"], rank, world),
           "layers": []}
    for lw in W["layers"]:
        out["layers"].append({
            "attn_norm": lw["attn_norm"], "ffn_norm": lw["ffn_norm"],
            "wq": _rows(lw["wq"], rank, world),
            "wk": _rows(lw["wk"], kv_rank, kv_world), "wv": _rows(lw["wv"], kv_rank, kv_world),
            "wo": _cols(lw["wo"], rank, world),
            "w1": _rows(lw["w1"], rank, world), "w3": _rows(lw["w3"], rank, world),
            "w2": _cols(lw["w2"], rank, world),
        })
    return out


# ---- 16-bit safetensors / GPTQ weights (oracle.dense_llama.make_weights / quantize_gptq dicts) -----------------------
# TensorParallelColumnLinear::load_with_hints shards dim 0 of [out, in] (distributed.rs:492-534); the GPTQ tensors
# are [k/8, n] / [k/g, n], so an OUT shard is a column range and an IN shard a row range (distributed.rs:565-654:
# qweight/scales/qzeros shard on dim 1 for column-parallel, dim 0 for row-parallel).

def _dense_out(w, rank, world):
    if isinstance(w, dict):
        n = w["qweight"].shape[1]
        if n % world:
            raise ValueError("out features not divisible by the shard count")
        c = n // world
        sl = slice(rank * c, (rank + 1) * c)
        return {"qweight": np.ascontiguousarray(w["qweight"][:, sl]), "scales": np.ascontiguousarray(w["scales"][:, sl]),
                "group": w["group"], "deq": np.ascontiguousarray(w["deq"][:, sl])}
    n = w.shape[0]
    if n % world:
        raise ValueError("out features not divisible by the shard count")
    c = n // world
    return np.ascontiguousarray(w[rank * c:(rank + 1) * c])


def _dense_in(w, rank, world):
    if isinstance(w, dict):
        k, g = w["deq"].shape[0], w["group"]
        if k % world or (k // world) % g or (k // world) % 8:
            raise ValueError("row-parallel GPTQ shard is not group aligned")
        c = k // world
        return {"qweight": np.ascontiguousarray(w["qweight"][rank * c // 8:(rank + 1) * c // 8]),
                "scales": np.ascontiguousarray(w["scales"][rank * c // g:(rank + 1) * c // g]),
                "group": g, "deq": np.ascontiguousarray(w["deq"][rank * c:(rank + 1) * c])}
    k = w.shape[1]
    if k % world:
        raise ValueError("in features not divisible by the shard count")
    c = k // world
    return np.ascontiguousarray(w[:, rank * c:(rank + 1) * c])


def shard_dense_config(cfg, rank, world):
    """local oracle.dense_llama.DenseConfig of this rank"""
    return shard_config(cfg, rank, world)


def shard_dense_weights(W, cfg, rank, world):
    """global oracle.dense_llama weights -> this rank's.  Embedding, norm vectors (and LayerNorm biases) are
    replicated; q/k/v biases follow their projection's out shard; lm_head is vocab-parallel."""
    _, kv_rank, kv_world = kv_head_shard(cfg.n_kv_heads, rank, world)
    out = {k: v for k, v in W.items() if k not in ("layers", "output")}
    out["output"] = _dense_out(W["output"], rank, world)
    out["layers"] = []
    for lw in W["layers"]:
        nl = {k: v for k, v in lw.items() if k.endswith("norm") or k.endswith("norm_b")}
        nl["wq"] = _dense_out(lw["wq"], rank, world)
        nl["wk"] = _dense_out(lw["wk"], kv_rank, kv_world)
        nl["wv"] = _dense_out(lw["wv"], kv_rank, kv_world)
        nl["wo"] = _dense_in(lw["wo"], rank, world)
        nl["w1"] = _dense_out(lw["w1"], rank, world)
        nl["w3"] = _dense_out(lw["w3"], rank, world)
        nl["w2"] = _dense_in(lw["w2"], rank, world)
        if "bq" in lw:
            nl["bq"] = _dense_out(lw["bq"], rank, world)
            nl["bk"] = _dense_out(lw["bk"], kv_rank, kv_world)
            nl["bv"] = _dense_out(lw["bv"], kv_rank, kv_world)
        out["layers"].append(nl)
    return out
