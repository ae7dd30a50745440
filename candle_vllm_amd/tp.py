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


VOCAB_PADDING_SIZE = 64


def pad_vocab_size(vocab_size, world_size):
    """distributed.rs:1446-1452: the vocabulary a vocab-parallel lm_head is padded to (zero rows; the gathered logits
    are narrowed back to vocab_size, :1661-1664).  Every BASELINE model's vocabulary is already a fixed point."""
    padded = (vocab_size + VOCAB_PADDING_SIZE - 1) // VOCAB_PADDING_SIZE * VOCAB_PADDING_SIZE
    per_rank = (padded + world_size - 1) // world_size * world_size
    return (per_rank + VOCAB_PADDING_SIZE - 1) // VOCAB_PADDING_SIZE * VOCAB_PADDING_SIZE


def _zero_blocks(t, n_rows, nb):
    """all-zero quantised rows: d (and dmin) = 0 and every code 0 dequantise to 0.0 in Q4_K, Q6_K and Q8_0 alike"""
    width = {12: 144, 14: 210, 8: 34}[t]
    return np.zeros((n_rows, nb, width), np.uint8)


def _rows_padded(tw, rank, world, vocab):
    """vocab-parallel lm_head (VocabParallelLinear, distributed.rs:1570-1630): the vocabulary is padded to pad_vocab_size, rank r
    owns rows [r * local, (r + 1) * local) of the padded matrix, the rows beyond the real vocabulary are ZERO rows, and the gathered
    logits are narrowed back to the real vocabulary.  (When the vocabulary does not divide, the reference's loader itself splits
    vocab // world rows per rank BEFORE padding, so that every rank's zero rows end up in front of the next rank's real rows and the
    narrowed logits are shifted -- DESIGN.md section 7; this is the layout of VocabParallelLinear::from_weight_bias, distributed.rs:1534-1550:
    pad the whole matrix, then narrow(rank * local, local): logits == the unsharded model's.)"""
    t, b = tw
    if pad_vocab_size(vocab, world) % world:
        raise ValueError(f"pad_vocab_size({vocab}, {world}) does not divide by the world size (a world that does not divide 64)")
    local = pad_vocab_size(vocab, world) // world
    lo, hi = rank * local, min((rank + 1) * local, vocab)
    real = b[lo:hi] if hi > lo else b[:0]
    if real.shape[0] == local:
        return (t, np.ascontiguousarray(real))
    return (t, np.ascontiguousarray(np.concatenate([real, _zero_blocks(t, local - real.shape[0], b.shape[1])], axis=0)))


def _rows(tw, rank, world):
    t, b = tw
    n = b.shape[0]
    if n % world:
        raise ValueError("rows not divisible by the shard count")
    c = n // world
    return (t, np.ascontiguousarray(b[rank * c:(rank + 1) * c]))


def _cols(tw, rank, world, requant=None):
    """column (k) shard: a k-block range when the blocks divide; else the reference's fallback -- dequantise, narrow,
    re-quantise to Q8_0 (quantized_var_builder.rs:234-269) -- through `requant(blocks, type, rank, world)`, which the caller
    supplies (the test-side restatement oracle.kquants.requantize_shard_q8_0; the product does it in
    mi355_gguf_tensor_shard_q8_0 when it loads a file)"""
    t, b = tw
    nb = b.shape[1]
    if nb % world:
        if requant is None or (nb * 256) % (32 * world):
            raise ValueError("column shard is not k-block aligned and no re-quantiser was supplied")
        return (8, requant(b, t, rank, world))                       # GGML_Q8_0
    c = nb // world
    return (t, np.ascontiguousarray(b[:, rank * c:(rank + 1) * c]))


def shard_config(cfg, rank, world):
    """local LlamaConfig (heads / kv heads / intermediate / vocab of this rank)"""
    if cfg.n_heads % world:
        raise ValueError("num_attention_heads must be divisible by world_size (attention.rs:553-554)")
    local = copy.copy(cfg)
    local.n_heads = cfg.n_heads // world
    local.n_kv_heads = kv_head_shard(cfg.n_kv_heads, rank, world)[0]
    local.intermediate = cfg.intermediate if getattr(cfg, "n_expert", 0) else cfg.intermediate // world   # MoE: replicated
    if pad_vocab_size(cfg.vocab, world) % world:
        raise ValueError(f"pad_vocab_size({cfg.vocab}, {world}) does not divide by the world size (a world that does not divide 64)")
    local.vocab = pad_vocab_size(cfg.vocab, world) // world          # rows of this rank's lm_head shard (zero rows included)
    local.vocab_total = cfg.vocab                                    # what the gathered logits are narrowed to
    return local


def shard_weights(W, cfg, rank, world, requant=None):
    """W: oracle.llama.make_weights dict (global) -> this rank's dict.  tok_embd and the norm vectors are
    replicated."""
    _, kv_rank, kv_world = kv_head_shard(cfg.n_kv_heads, rank, world)
    out = {"tok_embd": W["tok_embd"], "output_norm": W["output_norm"], "output": _rows_padded(W["output"], rank, world, cfg.vocab),
           "layers": []}
    for lw in W["layers"]:
        nl = {"attn_norm": lw["attn_norm"], "ffn_norm": lw["ffn_norm"],
              "wq": _rows(lw["wq"], rank, world),
              "wk": _rows(lw["wk"], kv_rank, kv_world), "wv": _rows(lw["wv"], kv_rank, kv_world),
              "wo": _cols(lw["wo"], rank, world, requant)}
        if "experts" in lw:
            # Mixtral GGUF: router and experts are loaded whole on every rank (quantized_llama.rs:344-365) -- the
            # MoE block is replicated, attention and the lm_head carry the parallelism
            nl["gate_inp"], nl["experts"] = lw["gate_inp"], lw["experts"]
        else:
            nl.update({"w1": _rows(lw["w1"], rank, world), "w3": _rows(lw["w3"], rank, world),
                       "w2": _cols(lw["w2"], rank, world, requant)})
        out["layers"].append(nl)
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


def _dense_out_padded(w, rank, world, vocab):
    """the vocab-parallel lm_head of the 16-bit path in the layout of VocabParallelLinear::from_weight_bias (distributed.rs:1534-1550):
    rows [rank * local, (rank + 1) * local) of the matrix padded with zero rows to pad_vocab_size (same layout as `_rows_padded`)"""
    if isinstance(w, dict):
        return _dense_out(w, rank, world)
    local = pad_vocab_size(vocab, world) // world
    lo, hi = rank * local, min((rank + 1) * local, vocab)
    real = w[lo:hi] if hi > lo else w[:0]
    if real.shape[0] == local:
        return np.ascontiguousarray(real)
    return np.ascontiguousarray(np.concatenate([real, np.zeros((local - real.shape[0],) + w.shape[1:], w.dtype)], axis=0))


def shard_dense_config(cfg, rank, world):
    """local oracle.dense_llama.DenseConfig of this rank"""
    return shard_config(cfg, rank, world)


def shard_dense_weights(W, cfg, rank, world):
    """global oracle.dense_llama weights -> this rank's.  Embedding, norm vectors (and LayerNorm biases) are
    replicated; q/k/v biases follow their projection's out shard; lm_head is vocab-parallel."""
    _, kv_rank, kv_world = kv_head_shard(cfg.n_kv_heads, rank, world)
    out = {k: v for k, v in W.items() if k not in ("layers", "output")}
    out["output"] = _dense_out_padded(W["output"], rank, world, cfg.vocab)
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


# ---- host-supplied collectives (mi355_comm_create_external) over torch.distributed ---------------------------------
class TorchDistComm:
    """A communicator whose all-reduce / all-gather are done by the HOST through a torch.distributed process group --
    what a host that already owns its communicator plugs in instead of a second RCCL one.  With the gloo backend the
    payload is staged through host memory, which lets two ranks share ONE GPU: the multi-rank host logic of the
    library (shard shapes, residual placement, vocab-parallel gather) then runs on a single-GPU box."""

    _ESIZE = {0: 4, 1: 2, 2: 2}                      # MI355_DTYPE_F32 / F16 / BF16

    def __init__(self, group=None):
        import ctypes
        import torch
        import torch.distributed as dist
        from ._lib import lib, ALLREDUCE_FN, ALLGATHER_FN
        self._torch, self._dist, self._group = torch, dist, group
        self._hip = ctypes.cdll.LoadLibrary("libamdhip64.so")
        self._ar = ALLREDUCE_FN(self._all_reduce)      # keep the callbacks alive as long as the handle
        self._ag = ALLGATHER_FN(self._all_gather)
        self.handle = lib.mi355_comm_create_external(self._ar, self._ag, None)
        if not self.handle:
            raise RuntimeError("mi355_comm_create_external failed")

    def _d2h(self, ptr, nbytes):
        import ctypes
        host = np.empty(nbytes, np.uint8)
        self._torch.cuda.synchronize()
        if self._hip.hipMemcpy(ctypes.c_void_p(host.ctypes.data), ctypes.c_void_p(ptr), ctypes.c_size_t(nbytes), 2) != 0:
            raise RuntimeError("hipMemcpy D2H failed")
        return host

    def _h2d(self, ptr, host):
        import ctypes
        if self._hip.hipMemcpy(ctypes.c_void_p(ptr), ctypes.c_void_p(host.ctypes.data), ctypes.c_size_t(host.nbytes), 1) != 0:
            raise RuntimeError("hipMemcpy H2D failed")

    @staticmethod
    def _to_f32(raw, dtype):
        if dtype == 0:
            return raw.view(np.float32).copy()
        if dtype == 1:
            return raw.view(np.float16).astype(np.float32)
        return (raw.view(np.uint16).astype(np.uint32) << 16).view(np.float32)

    @staticmethod
    def _from_f32(vals, dtype):
        if dtype == 0:
            return vals.astype(np.float32).view(np.uint8)
        if dtype == 1:
            return vals.astype(np.float16).view(np.uint8)
        u = np.ascontiguousarray(vals, np.float32).view(np.uint32)
        return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16).view(np.uint8)      # RNE, as the device sum rounds

    def _all_reduce(self, user, buf, count, dtype, stream):
        try:
            vals = self._to_f32(self._d2h(buf, count * self._ESIZE[dtype]), dtype)
            t = self._torch.from_numpy(vals)
            self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self._group)
            self._h2d(buf, np.ascontiguousarray(self._from_f32(t.numpy(), dtype)))
            return 0
        except Exception:                             # never unwind through the C frames
            return 999

    def _all_gather(self, user, send, recv, count, dtype, stream):
        try:
            raw = self._torch.from_numpy(self._d2h(send, count * self._ESIZE[dtype]))
            outs = [self._torch.empty_like(raw) for _ in range(self._dist.get_world_size(self._group))]
            self._dist.all_gather(outs, raw, group=self._group)
            self._h2d(recv, np.ascontiguousarray(np.concatenate([o.numpy() for o in outs])))
            return 0
        except Exception:
            return 999

    def set_options(self, side_stream=1, wire_bf16=0):
        """wire_bf16 = 1: the reference's all-reduce numerics (bf16 partials, bf16 sum, then + residual; attention.rs:1003-1008)"""
        from ._lib import lib
        if lib.mi355_comm_set_options(self.handle, int(side_stream), int(wire_bf16)) != 0:
            raise RuntimeError("mi355_comm_set_options failed")

    def attach_p2p(self):
        """one-shot peer-to-peer all-reduce for decode-sized messages: every rank exports its region, the 64-byte IPC
        handles travel through the process group (as the RCCL unique id does), every rank opens its peers' regions"""
        import ctypes
        from ._lib import lib
        h = ctypes.create_string_buffer(64)
        rc = lib.mi355_comm_p2p_export(self.handle, ctypes.addressof(h))
        if rc != 0:
            raise RuntimeError(f"mi355_comm_p2p_export failed with hipError {rc}")
        world, rank = self._dist.get_world_size(self._group), self._dist.get_rank(self._group)
        all_h = [None] * world
        self._dist.all_gather_object(all_h, bytes(h.raw), group=self._group)
        blob = ctypes.create_string_buffer(b"".join(all_h), 64 * world)
        rc = lib.mi355_comm_p2p_attach(self.handle, ctypes.addressof(blob), rank, world)
        if rc != 0:
            raise RuntimeError(f"mi355_comm_p2p_attach failed with hipError {rc}")

    def close(self):
        from ._lib import lib
        if self.handle:
            lib.mi355_comm_destroy(self.handle)
            self.handle = None
