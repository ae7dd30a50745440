"""Host-side mirror of the reference's operator interface for the decode hot path, over the C ABI.

Every function takes torch tensors that live on the MI355X (torch is only the owner of device memory and
of the stream), hands raw pointers to `libmi355vllm.so`, and raises on any failure -- there is no eager /
CPU fallback.  Names, argument order and semantics follow the reference call sites:

  copy_blocks(key_caches, value_caches, block_mapping)        src/backend/cache.rs:15-165
  swap_blocks(src, dst, mapping)                               src/scheduler/cache_engine.rs:527-535
  PagedAttention(...).forward(q,k,v,mask,kc,vc,meta,softcap)   src/openai/models/layers/attention.rs:983-995
  FusedRope.apply_inplace[_partial]                            src/openai/models/layers/rotary_emb.rs:58-70
  QMatMul.forward(x_f32)                                       src/openai/models/layers/attention.rs:920-922
  rms_norm / silu_mul                                          layers/qrmsnorm.rs:28-31, quantized_llama.rs:33-37
  InputMetadata                                                src/openai/pipelines/inputs.rs:552-568
"""
import ctypes
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch

from ._lib import lib, QmmDesc

DT_F32, DT_F16, DT_BF16, DT_U8 = 0, 1, 2, 3
KV_FLASH, KV_PAGED = 0, 1
GGML_Q4_K, GGML_Q6_K = 12, 14
EPI_STORE, EPI_RESID, EPI_SILU_MUL, EPI_QKV_ROPE_CACHE = 0, 1, 2, 3
SWAP_H2D, SWAP_D2H, SWAP_D2D = 0, 1, 2

_DT = {torch.float32: DT_F32, torch.float16: DT_F16, torch.bfloat16: DT_BF16, torch.uint8: DT_U8}


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with hipError {rc}")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU: the MI355X path has no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    return t.data_ptr()


@dataclass
class InputMetadata:
    """attention_rs::InputMetadata as the reference fills it (inputs.rs:351-367 prefill, :552-568 decode)."""
    is_prefill: bool
    slot_mapping: torch.Tensor                       # i64 [T]
    block_tables: Optional[torch.Tensor] = None      # u32 [n, max_blocks] (stored as int32 bits)
    context_lens: Optional[torch.Tensor] = None      # u32 [n]
    cu_seqlens_q: Optional[torch.Tensor] = None
    cu_seqlens_k: Optional[torch.Tensor] = None
    max_seqlen_q: int = 0
    max_seqlen_k: int = 0
    max_context_len: int = 0
    sequence_ids: Optional[List[int]] = None
    seqlens: Optional[List[int]] = None
    is_mla: bool = False
    is_mtp_verify: bool = False
    use_cached_kv: Optional[bool] = None             # decided ONCE per step on the host (inputs.rs:132-143), not per layer call

    def cached_prefix(self) -> bool:
        """does any sequence of this prompt step have a cached prefix (seqlen_k = cached + chunk, inputs.rs:132-143)?  The
        answer is computed once per step and kept: comparing the device tensors in every layer's call is a host
        synchronisation per layer."""
        if self.use_cached_kv is None:
            uc = self.max_seqlen_k > self.max_seqlen_q
            if not uc and self.cu_seqlens_k is not None and self.cu_seqlens_q is not None:
                uc = not torch.equal(self.cu_seqlens_k, self.cu_seqlens_q)
            self.use_cached_kv = bool(uc)
        return self.use_cached_kv

    @staticmethod
    def from_oracle_meta(meta: dict, device, is_prefill=False):
        def t(a, dt):
            return torch.from_numpy(np.ascontiguousarray(a).astype(dt)).to(device)
        return InputMetadata(
            is_prefill=is_prefill,
            slot_mapping=t(meta["slot_mapping"], np.int64),
            block_tables=t(meta["block_tables"].astype(np.int64), np.int32),
            context_lens=t(meta["context_lens"].astype(np.int64), np.int32),
            max_context_len=int(meta["max_context_len"]),
            cu_seqlens_q=t(meta["cu_seqlens_q"].astype(np.int64), np.int32) if "cu_seqlens_q" in meta else None,
            cu_seqlens_k=t(meta["cu_seqlens_k"].astype(np.int64), np.int32) if "cu_seqlens_k" in meta else None,
            max_seqlen_q=int(meta.get("max_seqlen_q", 0)), max_seqlen_k=int(meta.get("max_seqlen_k", 0)),
            use_cached_kv=(bool(np.any(np.asarray(meta["cu_seqlens_k"]) != np.asarray(meta["cu_seqlens_q"])))
                           if ("cu_seqlens_k" in meta and "cu_seqlens_q" in meta) else None))


# ----------------------------------------------------------------------------------------------- cache ops
def kv_layout_of(key_cache: torch.Tensor) -> int:
    if key_cache.dim() == 4:
        return KV_FLASH          # [NB, bs, Hkv, D]           cache_engine.rs:326-341
    if key_cache.dim() == 5:
        return KV_PAGED          # [NB, Hkv, D/x, bs, x]      cache_engine.rs:298-312
    raise ValueError("unrecognised KV cache rank")


def copy_blocks(key_caches: List[torch.Tensor], value_caches: List[torch.Tensor],
                block_mapping: Dict[int, List[int]]):
    """src/backend/cache.rs:15-165: for every layer copy block src -> each dst of K and of V."""
    if len(key_caches) == 0:
        return
    kc0, vc0 = key_caches[0], value_caches[0]
    if not kc0.is_cuda:
        raise RuntimeError("copy_blocks: not implemented for CPU (cache.rs:251-258); caches must be on the GPU")
    if kc0.device != vc0.device:
        raise RuntimeError("`key` and `value` caches have different devices")
    if kc0.dtype != vc0.dtype:
        raise RuntimeError("Key and value caches have different types")
    fn = {torch.bfloat16: lib.copy_blocks_bf16, torch.float16: lib.copy_blocks_f16,
          torch.float32: lib.copy_blocks_f32, torch.uint8: lib.copy_blocks_u8}.get(kc0.dtype)
    if fn is None:
        raise RuntimeError("only f32, f16, bf16 (and u8/fp8) input data type supported!")
    kptrs = (ctypes.c_uint64 * len(key_caches))(*[_dev(k, "key cache") for k in key_caches])
    vptrs = (ctypes.c_uint64 * len(value_caches))(*[_dev(v, "value cache") for v in value_caches])
    pairs = []
    for src, dsts in block_mapping.items():
        for d in dsts:
            pairs += [int(src), int(d)]
    if not pairs:
        return
    bm = (ctypes.c_int64 * len(pairs))(*pairs)
    numel_per_block = int(np.prod(kc0.shape[1:]))
    fn(ctypes.cast(kptrs, ctypes.c_void_p), ctypes.cast(vptrs, ctypes.c_void_p),
       ctypes.cast(bm, ctypes.c_void_p), len(key_caches), len(pairs) // 2, numel_per_block, _stream())


def swap_blocks(src: torch.Tensor, dst: torch.Tensor, mapping: Dict[int, int]):
    """attention_rs::cache::swap_blocks -- cache_engine.rs:527-535 (either side may be a CPU tensor)."""
    if not mapping:
        return 0
    if not src.is_contiguous() or not dst.is_contiguous():
        raise RuntimeError("swap_blocks needs contiguous tensors")
    if src.is_cuda and dst.is_cuda:
        kind = SWAP_D2D
    elif src.is_cuda:
        kind = SWAP_D2H
    elif dst.is_cuda:
        kind = SWAP_H2D
    else:
        raise RuntimeError("swap_blocks: at least one side must be on the GPU")
    bytes_per_block = (src.numel() // src.shape[0]) * src.element_size()
    flat = []
    for s, d in mapping.items():
        flat += [int(s), int(d)]
    arr = (ctypes.c_int64 * len(flat))(*flat)
    _check(lib.mi355_swap_blocks(src.data_ptr(), dst.data_ptr(), ctypes.cast(arr, ctypes.c_void_p),
                                 len(flat) // 2, bytes_per_block, kind, _stream()), "swap_blocks")
    return bytes_per_block * len(mapping)


def reshape_and_cache(k, v, key_cache, value_cache, slot_mapping):
    layout = kv_layout_of(key_cache)
    T, Hkv, D = k.shape
    bs = key_cache.shape[1] if layout == KV_FLASH else key_cache.shape[3]
    if k.dtype != key_cache.dtype:
        raise RuntimeError("k/v dtype must equal the cache dtype")
    _check(lib.mi355_reshape_and_cache(_dev(k), _dev(v), _dev(key_cache), _dev(value_cache),
                                       _dev(slot_mapping), T, Hkv, D, bs, k.element_size(), layout, _stream()),
           "reshape_and_cache")


# ----------------------------------------------------------------------------------------------- attention
V1_MAX_CONTEXT = 256          # above this the context is partitioned (v2), vLLM-style
TARGET_WORKGROUPS = 2048      # one wave per partition: ~8 waves per CU


def choose_partition(num_seqs, num_kv_heads, max_context_len):
    """Partition size (tokens) for v2 so that the grid has >> 256 workgroups; multiple of 64."""
    if max_context_len <= V1_MAX_CONTEXT:
        return 0
    per_seq = max(1, -(-TARGET_WORKGROUPS // max(1, num_seqs * num_kv_heads)))
    ps = -(-max_context_len // per_seq)
    ps = max(32, ((ps + 31) // 32) * 32)
    return ps if ps < max_context_len else 0


class PagedAttention:
    """attention_rs::PagedAttention -- constructed as in attention.rs:888-897, forward as :983-995."""

    def __init__(self, num_heads, head_dim, scale, num_kv_heads=None, sliding_window=None, device=None,
                 alibi_slopes=None, fp8_kvcache=False):
        if alibi_slopes is not None:
            raise NotImplementedError("alibi slopes: every call site of the reference passes None (attention.rs:573,895)")
        # sliding_window (attention.rs:570,893): the query at position i sees keys i - w + 1 .. i; served by the generic kernels
        # (mi355_paged_attention_window / mi355_prefill_attention_window), no fp8 cache
        self.sliding_window = int(sliding_window) if sliding_window else 0
        if self.sliding_window and fp8_kvcache:
            raise NotImplementedError("sliding window over an fp8 KV cache")
        self.fp8_kvcache = bool(fp8_kvcache)           # is_fp8_keys (attention.rs:574,896): U8 cache, e4m3fn, PAGED layout
        self.k_scale = self.v_scale = 1.0
        self.num_heads, self.head_dim, self.scale = num_heads, head_dim, float(scale)
        self.num_kv_heads = num_kv_heads or num_heads
        self._tmp = None

    def _workspace(self, B, P, device):
        need = (B, self.num_heads, P)
        if self._tmp is None or self._tmp[0] != need:
            self._tmp = (need,
                         torch.empty((B, self.num_heads, P, self.head_dim), dtype=torch.float32, device=device),
                         torch.empty((B, self.num_heads, P), dtype=torch.float32, device=device),
                         torch.empty((B, self.num_heads, P), dtype=torch.float32, device=device))
        return self._tmp[1:]

    def forward(self, q, k, v, mask, key_cache, value_cache, input_metadata: InputMetadata,
                softcapping: Optional[float] = None, partition_size: Optional[int] = None):
        """q [T,H,D], k/v [T,Hkv,D] 16-bit.  Writes k,v into the paged cache (K1) then attends (K2/K3)."""
        if key_cache is None or value_cache is None:
            raise RuntimeError("PagedAttention.forward needs the paged KV cache")
        if self.fp8_kvcache:
            return self._forward_fp8(q, k, v, key_cache, value_cache, input_metadata, softcapping, partition_size)
        reshape_and_cache(k, v, key_cache, value_cache, input_metadata.slot_mapping)
        if input_metadata.is_prefill:
            return self.prefill(q, k, v, key_cache, value_cache, input_metadata, softcapping)
        return self.decode(q, key_cache, value_cache, input_metadata, softcapping, partition_size)

    def _forward_fp8(self, q, k, v, key_cache, value_cache, meta, softcapping, partition_size):
        """fp8 (e4m3fn) KV cache, PAGED layout with x = 16: K [NB,Hkv,D/16,bs,16], V [NB,Hkv,D,bs] uint8."""
        if key_cache.dtype != torch.uint8 or key_cache.dim() != 5 or q.dtype != torch.bfloat16:
            raise RuntimeError("fp8 KV cache: uint8 caches in the paged layout and bf16 activations")
        T, H, D = q.shape
        bs = key_cache.shape[3]
        sc = float(softcapping) if softcapping else 0.0
        _check(lib.mi355_reshape_and_cache_fp8(_dev(k), _dev(v), _dev(key_cache), _dev(value_cache),
                                               _dev(meta.slot_mapping), T, self.num_kv_heads, D, bs, KV_PAGED,
                                               self.k_scale, self.v_scale, _stream()), "reshape_and_cache_fp8")
        out = torch.empty_like(q)
        bt, cl = meta.block_tables, meta.context_lens
        if meta.is_prefill:
            n = meta.cu_seqlens_q.shape[0] - 1
            _check(lib.mi355_prefill_attention_fp8(_dev(out), _dev(q), _dev(key_cache), _dev(value_cache), _dev(bt),
                                                   _dev(cl), _dev(meta.cu_seqlens_q), n, meta.max_seqlen_q, H,
                                                   self.num_kv_heads, D, bs, bt.shape[1], self.scale, sc,
                                                   self.k_scale, self.v_scale, DT_BF16, _stream()), "prefill_attention_fp8")
            return out
        ps = choose_partition(T, self.num_kv_heads, meta.max_context_len) if partition_size is None else partition_size
        if ps:
            if not (ps in (256, 512) and ps % bs == 0):             # 256 / 512: chunks walked inside the wave (pa_mfma_chunk)
                ps = 32 if ps <= 32 else (64 if ps <= 64 else 128)
            P = -(-meta.max_context_len // ps)
            tmp, mx, sm = self._workspace(T, P, q.device)
            _check(lib.mi355_paged_attention_fp8(_dev(out), _dev(sm), _dev(mx), _dev(tmp), _dev(q), _dev(key_cache),
                                                 _dev(value_cache), _dev(bt), _dev(cl), T, H, self.num_kv_heads, D, bs,
                                                 bt.shape[1], meta.max_context_len, ps, self.scale, sc, self.k_scale,
                                                 self.v_scale, _stream()), "paged_attention_fp8")
        else:
            _check(lib.mi355_paged_attention_fp8(_dev(out), None, None, None, _dev(q), _dev(key_cache),
                                                 _dev(value_cache), _dev(bt), _dev(cl), T, H, self.num_kv_heads, D, bs,
                                                 bt.shape[1], meta.max_context_len, 0, self.scale, sc, self.k_scale,
                                                 self.v_scale, _stream()), "paged_attention_fp8")
        return out

    def prefill(self, q, k, v, key_cache, value_cache, meta: InputMetadata, softcapping=None):
        """K4.  Cached prefix (cu_seqlens_k != cu_seqlens_q, `use_cached_kv` inputs.rs:133-143) -> keys come from
        the paged cache; otherwise from the chunk's own k, v."""
        T, H, D = q.shape
        out = torch.empty_like(q)
        sc = float(softcapping) if softcapping else 0.0
        n = meta.cu_seqlens_q.shape[0] - 1
        # per sequence, as the reference decides it (seqlen_k = cached + chunk whenever num_cached > 0, inputs.rs:132-143):
        # ANY sequence with a cached prefix sends the whole batch through the cache path.  (max_seqlen_k > max_seqlen_q
        # misses a mixed batch such as 16 cached + 16 new beside 64 fresh tokens.)
        use_cached = meta.cached_prefix()
        if self.sliding_window:
            if use_cached:
                layout = kv_layout_of(key_cache)
                bs = key_cache.shape[1] if layout == KV_FLASH else key_cache.shape[3]
                _check(lib.mi355_prefill_attention_window(_dev(out), _dev(q), None, None, _dev(key_cache), _dev(value_cache),
                                                          _dev(meta.block_tables), _dev(meta.context_lens), _dev(meta.cu_seqlens_q), n,
                                                          meta.max_seqlen_q, H, self.num_kv_heads, D, bs, meta.block_tables.shape[1],
                                                          self.scale, sc, layout, _DT[q.dtype], self.sliding_window, _stream()),
                       "prefill_attention_window")
            else:
                _check(lib.mi355_prefill_attention_window(_dev(out), _dev(q), _dev(k), _dev(v), None, None, None, None,
                                                          _dev(meta.cu_seqlens_q), n, meta.max_seqlen_q, H, self.num_kv_heads, D, 0, 0,
                                                          self.scale, sc, 0, _DT[q.dtype], self.sliding_window, _stream()),
                       "prefill_attention_window")
            return out
        if use_cached:
            layout = kv_layout_of(key_cache)
            bs = key_cache.shape[1] if layout == KV_FLASH else key_cache.shape[3]
            _check(lib.mi355_prefill_attention(_dev(out), _dev(q), None, None, _dev(key_cache), _dev(value_cache),
                                               _dev(meta.block_tables), _dev(meta.context_lens),
                                               _dev(meta.cu_seqlens_q), n, meta.max_seqlen_q, H, self.num_kv_heads,
                                               D, bs, meta.block_tables.shape[1], self.scale, sc, layout,
                                               _DT[q.dtype], _stream()), "prefill_attention")
        else:
            _check(lib.mi355_prefill_attention(_dev(out), _dev(q), _dev(k), _dev(v), None, None, None, None,
                                               _dev(meta.cu_seqlens_q), n, meta.max_seqlen_q, H, self.num_kv_heads,
                                               D, 0, 0, self.scale, sc, 0, _DT[q.dtype], _stream()),
                   "prefill_attention")
        return out

    def decode(self, q, key_cache, value_cache, meta: InputMetadata, softcapping=None, partition_size=None):
        layout = kv_layout_of(key_cache)
        B, H, D = q.shape
        bs = key_cache.shape[1] if layout == KV_FLASH else key_cache.shape[3]
        out = torch.empty_like(q)
        dt = _DT[q.dtype]
        bt, cl = meta.block_tables, meta.context_lens
        sc = float(softcapping) if softcapping else 0.0
        if self.sliding_window:
            _check(lib.mi355_paged_attention_window(_dev(out), _dev(q), _dev(key_cache), _dev(value_cache), _dev(bt), _dev(cl), B, H,
                                                    self.num_kv_heads, D, bs, bt.shape[1], meta.max_context_len, self.scale, sc, layout, dt,
                                                    self.sliding_window, _stream()), "paged_attention_window")
            return out
        ps = choose_partition(B, self.num_kv_heads, meta.max_context_len) if partition_size is None else partition_size
        if layout == KV_PAGED and ps == 0 and meta.max_context_len > 8192:
            ps = 4096
        if (partition_size is None and ps and layout == KV_PAGED and q.dtype == torch.bfloat16 and D in (64, 128)
                and H // self.num_kv_heads <= 16 and bs % 16 == 0):
            # the MFMA kernel's shapes: 32-token partitions, several per workgroup -- the choice of the C++ step drivers
            # (host_model.cpp / dense_model.cpp).  Any other size would fall to the generic kernel, 40x slower at 4 k contexts
            # (tools/exp_attn_loop.py); an explicit partition_size is passed through (256 / 512: the looped chunks)
            ps = 32
        if ps == 0:
            _check(lib.mi355_paged_attention_v1(_dev(out), _dev(q), _dev(key_cache), _dev(value_cache), _dev(bt),
                                                _dev(cl), B, H, self.num_kv_heads, D, bs, bt.shape[1],
                                                meta.max_context_len, self.scale, sc, layout, dt, _stream()),
                   "paged_attention_v1")
        else:
            P = -(-meta.max_context_len // ps)
            tmp, mx, sm = self._workspace(B, P, q.device)
            _check(lib.mi355_paged_attention_v2(_dev(out), _dev(sm), _dev(mx), _dev(tmp), _dev(q), _dev(key_cache),
                                                _dev(value_cache), _dev(bt), _dev(cl), B, H, self.num_kv_heads, D,
                                                bs, bt.shape[1], meta.max_context_len, ps, self.scale, sc, layout,
                                                dt, _stream()),
                   "paged_attention_v2")
        return out


class FusedRope:
    """attention_rs::fused_rope::FusedRope (rotary_emb.rs:58-70); q,k rotated IN PLACE."""

    @staticmethod
    def apply_inplace(q, k, cos, sin, positions, is_rope_i):
        FusedRope.apply_inplace_partial(q, k, cos, sin, positions, is_rope_i, q.shape[-1])

    @staticmethod
    def apply_inplace_partial(q, k, cos, sin, positions, is_rope_i, rotary_dim):
        T, H, D = q.shape
        if cos.dtype != torch.float32 or sin.dtype != torch.float32:
            raise RuntimeError("cos/sin tables are F32 (llama.rs:222, quantized_llama.rs:313-318)")
        _check(lib.mi355_rope_inplace(_dev(q), _dev(k), _dev(cos), _dev(sin), _dev(positions), T, H, k.shape[1], D,
                                      int(rotary_dim), 1 if is_rope_i else 0, _DT[q.dtype], _stream()), "rope")


# ----------------------------------------------------------------------------------------------- small ops
def rms_norm(x, w, eps):
    out = torch.empty_like(x)
    _check(lib.mi355_rms_norm(_dev(out), _dev(x), _dev(w), x.shape[0], x.shape[1], float(eps), _DT[x.dtype],
                              _DT[w.dtype], _stream()), "rms_norm")
    return out


def silu_mul(gate, up):
    out = torch.empty_like(gate)
    _check(lib.mi355_silu_mul(_dev(out), _dev(gate), _dev(up), gate.numel(), _DT[gate.dtype], _stream()), "silu_mul")
    return out


def add(a, b):
    out = torch.empty_like(a)
    _check(lib.mi355_add_f32(_dev(out), _dev(a), _dev(b), a.numel(), _stream()), "add")
    return out


def cast(x, dtype):
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    _check(lib.mi355_cast(_dev(out), _dev(x), x.numel(), _DT[x.dtype], _DT[dtype], _stream()), "cast")
    return out


def embedding(table, ids_u32):
    out = torch.empty((ids_u32.shape[0], table.shape[1]), dtype=torch.float32, device=table.device)
    _check(lib.mi355_embedding_f32(_dev(out), _dev(table), _dev(ids_u32), ids_u32.shape[0], table.shape[1],
                                   _stream()), "embedding")
    return out


def argmax(logits):
    out = torch.empty((logits.shape[0],), dtype=torch.int32, device=logits.device)
    _check(lib.mi355_argmax_f32(_dev(out), _dev(logits), logits.shape[0], logits.shape[1], _stream()), "argmax")
    return out


# ----------------------------------------------------------------------------------------------- quantised matmul
def repack_qweight(blocks_native: np.ndarray, ggml_type: int, n_rows: int, k: int) -> np.ndarray:
    """Load-time re-tiling (host).  blocks_native: uint8 [n_rows, k/256, block_bytes]."""
    n = lib.mi355_qweight_repacked_size(ggml_type, n_rows, k)
    if n < 0:
        raise ValueError("unsupported ggml type / shape")
    src = np.ascontiguousarray(blocks_native, dtype=np.uint8)
    dst = np.empty(n, np.uint8)
    _check(lib.mi355_qweight_repack(dst.ctypes.data, src.ctypes.data, ggml_type, n_rows, k), "qweight_repack")
    return dst


class QMatMul:
    """candle_core::quantized::QMatMul over a GGUF Q4_K / Q6_K tensor [N, K] (attention.rs:875-881)."""

    def __init__(self, blocks_native: np.ndarray, ggml_type: int, device):
        self.ggml_type = ggml_type
        self.n = blocks_native.shape[0]
        self.native = torch.from_numpy(np.ascontiguousarray(blocks_native)).to(device)
        if ggml_type == 8:                                  # GGML_Q8_0: a re-quantised TP shard, native 34-byte blocks, no repack
            self.k = blocks_native.shape[1] * 32
            self.tiles = self.native
        else:
            self.k = blocks_native.shape[1] * 256
            self.tiles = torch.from_numpy(repack_qweight(blocks_native, ggml_type, self.n, self.k)).to(device)

    def forward(self, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        if x.dtype != torch.float32:
            raise RuntimeError("QMatMul::forward takes F32 activations (linear.rs:769-782)")
        T = x.shape[0]
        out = torch.empty((T, self.n), dtype=torch.float32, device=x.device)
        _check(lib.mi355_qmatmul(_dev(out), _dev(x), _dev(self.tiles), self.ggml_type, T, self.n, self.k,
                                 _dev(bias) if bias is not None else None, _stream()), "qmatmul")
        return out

    def forward_ref(self, x: torch.Tensor) -> torch.Tensor:
        """simple native-layout kernel (on-device cross-check)"""
        T = x.shape[0]
        out = torch.empty((T, self.n), dtype=torch.float32, device=x.device)
        _check(lib.mi355_qmatmul_ref(_dev(out), _dev(x), _dev(self.native), self.ggml_type, T, self.n, self.k,
                                     _stream()), "qmatmul_ref")
        return out

    def dequantize(self) -> torch.Tensor:
        out = torch.empty((self.n, self.k), dtype=torch.float32, device=self.native.device)
        _check(lib.mi355_dequantize(_dev(out), _dev(self.native), self.ggml_type, self.n * self.k, _stream()),
               "dequantize")
        return out


def qmatmul_fused(mats: List[QMatMul], x, *, epilogue, out=None, norm_weight=None, norm_eps=0.0, residual=None,
                  bias=None, rope=None):
    """[RMSNorm ->] several QMatMuls sharing x -> fused epilogue (see include/mi355_vllm.h).
    rope = dict(cos, sin, positions, slot_mapping, q_out, key_cache, value_cache, num_heads, num_kv_heads,
                head_dim, rotary_dim) for EPI_QKV_ROPE_CACHE."""
    d = QmmDesc()
    d.nseg = len(mats)
    for i, m in enumerate(mats):
        d.w_tiles[i] = _dev(m.tiles)
        d.ggml_type[i] = m.ggml_type
        d.n_rows[i] = m.n
    d.x = _dev(x)
    d.x_dtype = _DT[x.dtype]
    d.ldx = x.shape[1]
    d.k = mats[0].k
    d.num_tokens = x.shape[0]
    d.norm_weight = _dev(norm_weight) if norm_weight is not None else None
    d.norm_eps = float(norm_eps)
    d.epilogue = epilogue
    if out is not None:
        d.out = _dev(out)
        d.ldo = out.shape[1]
    d.residual = _dev(residual) if residual is not None else None
    d.bias = _dev(bias) if bias is not None else None
    if rope is not None:
        kc = rope["key_cache"]
        layout = kv_layout_of(kc)
        d.cos_table, d.sin_table = _dev(rope["cos"]), _dev(rope["sin"])
        d.positions, d.slot_mapping = _dev(rope["positions"]), _dev(rope["slot_mapping"])
        d.q_out, d.key_cache, d.value_cache = _dev(rope["q_out"]), _dev(kc), _dev(rope["value_cache"])
        d.num_heads, d.num_kv_heads, d.head_dim = rope["num_heads"], rope["num_kv_heads"], rope["head_dim"]
        d.rotary_dim = rope.get("rotary_dim", rope["head_dim"])
        d.block_size = kc.shape[1] if layout == KV_FLASH else kc.shape[3]
        d.kv_layout = layout
    _check(lib.mi355_qmatmul_fused(ctypes.byref(d), _stream()), "qmatmul_fused")
    return out


# ------------------------------------------------------------------------------------------------ safetensors path
ZERO_SYM8, ZERO_GPTQ_PLUS1, ZERO_AWQ_MARLIN = 0, 1, 2


def _dt16(t):
    if t.dtype not in (torch.bfloat16, torch.float16):
        raise RuntimeError("16-bit linear layers take bf16 / f16 activations (linear.rs:124-172, gptq.rs:229-238)")
    return _DT[t.dtype]


class Linear:
    """`Linear` of src/openai/models/linear.rs:64-172: y = x.W^T (+b), weight [out, in] as stored in the checkpoint.
    epilogue (decode fusions): EPI_STORE, EPI_RESID (y + residual), EPI_SILU_MUL (packed gate_up, mlp.rs:324-352)."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, tiled: bool = False):
        """tiled=True: keep the weight as the 16-row x 256-k tile image (mi355_dense_tile_repack; n % 16 == 0, k % 256 == 0), as the
        16-bit host layer does with its projections"""
        self.bias = bias
        self.n, self.k = weight.shape
        self.tiled = bool(tiled)
        if self.tiled:
            t = torch.empty_like(weight)
            _check(lib.mi355_dense_tile_repack(_dev(weight), _dev(t), self.n, self.k, self.k, _stream()), "dense_tile_repack")
            weight = t
        self.weight = weight

    def forward(self, x, *, epilogue=EPI_STORE, residual=None, out=None):
        dt = _dt16(x)
        if self.weight.dtype != x.dtype:
            raise RuntimeError("weight / activation dtype mismatch")
        x2 = x.reshape(-1, self.k)
        T = x2.shape[0]
        n_out = self.n // 2 if epilogue == EPI_SILU_MUL else self.n
        if out is None:
            out = torch.empty((T, n_out), dtype=x.dtype, device=x.device)
        fn = lib.mi355_linear_tiled if self.tiled else lib.mi355_linear
        _check(fn(_dev(out), _dev(x2), _dev(self.weight), _dev(self.bias) if self.bias is not None else None,
                  _dev(residual) if residual is not None else None, T, self.n, self.k, dt, epilogue, _stream()), "linear")
        return out


def marlin_weight_repack(qweight: torch.Tensor, bits: int, is_awq: bool) -> torch.Tensor:
    """`marlin_weight_repack` (src/backend/gptq.rs:264-359): u32 checkpoint tensor -> the layout the marlin entry
    points consume, returned with the reference's output shape [k/16, 2n]."""
    if qweight.dtype not in (torch.int32, torch.uint32):
        raise RuntimeError("qweight must be a u32 tensor")
    if bits != 4:
        raise RuntimeError("only 4-bit checkpoints are supported on this path")
    pack = 32 // bits
    d0, d1 = qweight.shape
    if is_awq:
        k, n = d0, d1 * pack
    else:
        k, n = d0 * pack, d1
    out = torch.empty((k // 16, 2 * n), dtype=qweight.dtype, device=qweight.device)
    if is_awq:
        lib.awq_repack(_dev(qweight), _dev(out), d0, d1, bits, _stream())
    else:
        lib.gptq_repack(_dev(qweight), _dev(out), d0, d1, _stream())
    return out


def gptq_matmul(x, qweight, scales, qzeros, g_idx, workspace, bits, group_size, is_awq):
    """`gptq_matmul` (src/backend/gptq.rs:242-262) -> GPTQMatMul::cuda_fwd_t (:26-204): same dispatch.
    workspace is not None  <=>  marlin format (qweight [k/16, 2n] from marlin_weight_repack, permuted scales);
    otherwise the exllama arm (f16 only, qweight [k/8, n])."""
    dt = _dt16(x)
    marlin = workspace is not None
    if marlin:
        k, n = qweight.shape[0] * 16, qweight.shape[1] // 2
    else:
        k, n = qweight.shape[0] * (32 // bits), qweight.shape[1]          # pack_factor = 32 / bits (gptq.rs:45-47)
    x2 = x.reshape(-1, k)
    m = x2.shape[0]
    out = torch.empty((m, n), dtype=x.dtype, device=x.device)
    zp = _dev(qzeros) if qzeros is not None else None
    gp = _dev(g_idx) if g_idx is not None else None
    if marlin:
        name = ("marlin_awq_4bit_" if is_awq else "marlin_4bit_") + ("f16" if dt == DT_F16 else "bf16")
        getattr(lib, name)(_dev(x2), _dev(qweight), _dev(scales), zp if is_awq else None, gp, _dev(out), m, k, n,
                           _dev(workspace), group_size, _stream())
    else:
        if dt != DT_F16:
            raise RuntimeError("GPTQMatMul is only supported for f16 non-marlin matmul (gptq.rs:196)")
        if qzeros is None or g_idx is None:
            raise RuntimeError("the exllama arm needs qzeros and g_idx")
        lib.gemm_half_q_half_alt(_dev(x2), _dev(qweight), zp, _dev(scales), gp, _dev(out), m, n, k, bits, _stream())
    err = lib.mi355_last_error()                     # the FFI symbols are `void`: the library records what it refused
    if err:
        lib.mi355_clear_error()
        raise RuntimeError(f"gptq_matmul refused by the device library (hipError {err}); the output was filled with NaN")
    return out.reshape(*x.shape[:-1], n)


def marlin_format_repack(B):
    """`checkpoint_format == "marlin"` (linear.rs:222-239,279-290): B [k/16, 2n] u32 in the Marlin tile order -> the
    [k/16, 2n]-shaped weight image marlin_weight_repack would have produced (same consumer: gptq_matmul with a workspace)."""
    k, n = B.shape[0] * 16, B.shape[1] // 2
    out = torch.empty_like(B)
    _check(lib.mi355_marlin_format_repack(_dev(B), _dev(out), k, n, _stream()), "marlin_format_repack")
    return out


def gptq_tile_unpack(qw_tiled, k, n):
    """the tiled weight image (marlin_weight_repack / marlin_format_repack output) back in checkpoint layout [k/8, n]"""
    out = torch.empty((k // 8, n), dtype=qw_tiled.dtype, device=qw_tiled.device)
    _check(lib.mi355_gptq_tile_unpack(_dev(qw_tiled), _dev(out), k, n, _stream()), "gptq_tile_unpack")
    return out


class GPTQLinear:
    """QLinear GPTQ arm (linear.rs:854-906) with the decode epilogues fused; qweight in checkpoint layout [k/8, n], re-ordered
    into the tiled image at construction when the shape allows (`tiled=None`: k % 256 == 0 and n % 16 == 0); scales natural or
    Marlin-permuted."""

    def __init__(self, qweight, scales, group_size, qzeros=None, zero_mode=ZERO_SYM8, scales_permuted=False, bias=None, tiled=None):
        self.scales, self.qzeros, self.bias = scales, qzeros, bias
        self.group_size, self.zero_mode, self.scales_permuted = group_size, zero_mode, scales_permuted
        self.k, self.n = qweight.numel() * 8 // scales.shape[-1], scales.shape[-1]
        if tiled is None:
            tiled = self.k % 256 == 0 and self.n % 16 == 0
        self.tiled = bool(tiled)
        if self.tiled:
            t = torch.empty_like(qweight)
            _check(lib.mi355_gptq_tile_repack(_dev(qweight), _dev(t), self.k, self.n, 0, _stream()), "gptq_tile_repack")
            qweight = t
        self.qweight = qweight

    def forward(self, x, *, epilogue=EPI_STORE, residual=None, out=None):
        dt = _dt16(x)
        x2 = x.reshape(-1, self.k)
        T = x2.shape[0]
        n_out = self.n // 2 if epilogue == EPI_SILU_MUL else self.n
        if out is None:
            out = torch.empty((T, n_out), dtype=x.dtype, device=x.device)
        fn = lib.mi355_gptq_linear_tiled if self.tiled else lib.mi355_gptq_linear
        _check(fn(_dev(out), _dev(x2), _dev(self.qweight), _dev(self.scales),
                  _dev(self.qzeros) if self.qzeros is not None else None, self.zero_mode, 1 if self.scales_permuted else 0,
                  _dev(self.bias) if self.bias is not None else None, _dev(residual) if residual is not None else None,
                  T, self.n, self.k, self.group_size, dt, epilogue, _stream()), "gptq_linear")
        return out


def rope_tables(rope_theta, rotary_dim, max_seq_len, scaling=None, max_position_embeddings=0):
    """`ScalingRotaryEmbedding::new` (layers/rotary_emb.rs:107-341): cos, sin f32 [n, rotary_dim/2] built by the
    library's host code.  scaling: None or a `_lib.RopeScaling`."""
    import ctypes
    scp = ctypes.byref(scaling) if scaling is not None else None
    n = lib.mi355_rope_table_len(scp, max_seq_len, max_position_embeddings)
    if n <= 0:
        raise ValueError("unknown rope scaling type")
    cos = np.empty((n, rotary_dim // 2), np.float32)
    sin = np.empty_like(cos)
    _check(lib.mi355_rope_tables(cos.ctypes.data, sin.ctypes.data, rotary_dim, n, float(rope_theta), scp, max_seq_len,
                                 max_position_embeddings), "rope_tables")
    return cos, sin
