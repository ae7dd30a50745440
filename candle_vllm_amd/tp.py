"""Tensor-parallel shard plan of the GGUF llama weights (host side, pure index arithmetic).

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
