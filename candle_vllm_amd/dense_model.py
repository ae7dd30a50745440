"""Python harness over the `mi355_dense_*` handle (16-bit safetensors llama-family host layer): mirrors
`Llama::forward` (src/openai/models/llama.rs:118-201).  No CPU fallback."""
import ctypes

import numpy as np
import torch

from ._lib import lib, DenseConfig
from .ops import _check, KV_FLASH, KV_PAGED  # noqa: F401

W_SLOTS = {"wq": 0, "wk": 1, "wv": 2, "wo": 3, "w1": 4, "w2": 5, "w3": 6, "attn_norm": 7, "ffn_norm": 8,
           "tok_embd": 9, "output_norm": 10, "output": 11, "bq": 12, "bk": 13, "bv": 14,
           "attn_norm_b": 15, "ffn_norm_b": 16, "output_norm_b": 17}
DT_BF16 = 2


def _bf16_bits(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


class DenseLlama:
    def __init__(self, cfg, max_batch=8, max_blocks_per_seq=64, kv_layout=KV_PAGED, rope_interleaved=False,
                 tp_rank=0, tp_world=1):
        """cfg: this rank's shard when tp_world > 1 (candle_vllm_amd.tp.shard_dense_config)"""
        self.cfg, self.kv_layout, self.tp_rank, self.tp_world, self.comm = cfg, kv_layout, tp_rank, tp_world, None
        c = DenseConfig(hidden=cfg.hidden, n_layers=cfg.n_layers, n_heads=cfg.n_heads, n_kv_heads=cfg.n_kv_heads,
                        head_dim=cfg.head_dim, intermediate=cfg.intermediate, vocab=cfg.vocab, max_seq=cfg.max_seq,
                        block_size=cfg.block_size, kv_layout=kv_layout, max_batch=max_batch,
                        max_blocks_per_seq=max_blocks_per_seq, rms_eps=cfg.rms_eps, rope_theta=cfg.rope_theta,
                        dtype=DT_BF16, rope_interleaved=int(rope_interleaved),
                        norm_type=1 if getattr(cfg, "layer_norm", False) else 0,
                        rotary_dim=int(getattr(cfg, "rotary_dim", 0) or 0),
                        kv_fp8=1 if getattr(cfg, "kv_fp8", False) else 0, tp_rank=tp_rank, tp_world=tp_world,
                        vocab_total=int(getattr(cfg, "vocab_total", 0) or 0) if tp_world > 1 else 0)
        self.h = lib.mi355_dense_create(ctypes.byref(c))
        if not self.h:
            raise RuntimeError("mi355_dense_create failed (bad config or no GPU memory)")

    def logits_width(self):
        """columns of a logits row: the lm_head's rows; with a communicator the gathered row narrowed to the real vocabulary"""
        if not (self.comm or getattr(self, 'comm_borrowed', None)):
            return self.cfg.vocab
        vt = int(getattr(self.cfg, "vocab_total", 0) or 0) if self.tp_world > 1 else 0
        return vt if vt > 0 else self.cfg.vocab * self.tp_world

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib.mi355_dense_destroy(self.h)
            self.h = None
        if getattr(self, "comm", None) and lib is not None:
            lib.mi355_comm_destroy(self.comm)
            self.comm = None

    def init_comm(self, id128=None):
        """RCCL communicator of the tensor-parallel group (one process per GPU): rank 0 makes the id, the launcher
        ships it (torch.distributed broadcast_object_list / the reference's pipe, pipeline.rs:805-812)."""
        buf = np.zeros(128, np.uint8)
        if id128 is None:
            _check(lib.mi355_comm_unique_id(buf.ctypes.data), "comm_unique_id")
        else:
            buf[:] = np.frombuffer(bytes(id128), np.uint8)
        self.comm = lib.mi355_comm_create(buf.ctypes.data, self.tp_rank, self.tp_world)
        if not self.comm:
            raise RuntimeError("mi355_comm_create failed (librccl missing or rendezvous error)")
        _check(lib.mi355_dense_set_comm(self.h, self.comm), "dense_set_comm")
        return bytes(buf)

    def set_comm(self, handle):
        """attach a communicator the caller owns (mi355_comm_create / tp.TorchDistComm().handle); not destroyed here"""
        _check(lib.mi355_dense_set_comm(self.h, handle), "dense_set_comm")
        self.comm_borrowed = handle

    def comm_capture_ok(self, stream):
        """can this stack capture the attached communicator's collectives in a hipGraph?  (local test, nothing goes on the wire.)  Tensor-
        parallel callers run it on EVERY rank, reduce the answers (min) over their process group and call set_graph(False) everywhere
        unless all ranks said yes: a rank that falls back alone would leave its peers inside a captured collective."""
        h = self.comm or getattr(self, "comm_borrowed", None)
        return bool(h) and lib.mi355_comm_capture_probe(h, stream) == 0

    def set_rope_tables(self, cos, sin):
        """replace the default RoPE tables: f32 [n >= max_seq, rotary_dim/2]"""
        cos, sin = np.ascontiguousarray(cos, np.float32), np.ascontiguousarray(sin, np.float32)
        _check(lib.mi355_dense_set_rope_tables(self.h, cos.ctypes.data, sin.ctypes.data, cos.shape[0]), "set_rope_tables")

    def set_weight(self, layer, name, values_f32):
        """values: f32 numpy, exactly representable in bf16 (checkpoint tensors)"""
        bits = _bf16_bits(values_f32)
        _check(lib.mi355_dense_set_weight(self.h, layer, W_SLOTS[name], bits.ctypes.data, bits.size), f"set_weight {name}")

    def set_gptq(self, layer, name, qweight_u32, scales_f32, group_size):
        """qweight [k/8, n] u32, scales [k/g, n] f32 values exactly representable in bf16"""
        qw = np.ascontiguousarray(qweight_u32, np.uint32)
        sc = _bf16_bits(scales_f32)
        k, n = qw.shape[0] * 8, qw.shape[1]
        _check(lib.mi355_dense_set_gptq(self.h, layer, W_SLOTS[name], qw.ctypes.data, sc.ctypes.data, n, k, group_size),
               f"set_gptq {name}")

    def load_oracle_weights(self, W):
        for name in ("tok_embd", "output_norm", "output", "output_norm_b"):
            if name in W:
                self.set_weight(-1, name, W[name])
        for l, lw in enumerate(W["layers"]):
            for name, v in lw.items():
                if isinstance(v, dict):                       # GPTQ projection {"qweight", "scales", "group"}
                    self.set_gptq(l, name, v["qweight"], v["scales"], v["group"])
                else:
                    self.set_weight(l, name, v)

    def load_synthetic(self, seed=0, std=0.02):
        """random bf16 weights generated on the GPU (bench)"""
        c = self.cfg
        g = torch.Generator(device="cuda").manual_seed(seed)

        def rnd(n, s=std, mean=0.0):
            return (torch.randn(n, generator=g, device="cuda") * s + mean).to(torch.bfloat16)

        def put(layer, name, t):
            _check(lib.mi355_dense_set_weight_dev(self.h, layer, W_SLOTS[name], t.data_ptr(), t.numel()), name)
            torch.cuda.synchronize()
        HD, KD = c.n_heads * c.head_dim, c.n_kv_heads * c.head_dim
        put(-1, "tok_embd", rnd(c.vocab * c.hidden, 0.5))
        put(-1, "output_norm", rnd(c.hidden, 0.02, 1.0))
        put(-1, "output", rnd(c.vocab * c.hidden))
        for l in range(c.n_layers):
            put(l, "attn_norm", rnd(c.hidden, 0.02, 1.0))
            put(l, "ffn_norm", rnd(c.hidden, 0.02, 1.0))
            put(l, "wq", rnd(HD * c.hidden)); put(l, "wk", rnd(KD * c.hidden)); put(l, "wv", rnd(KD * c.hidden))
            put(l, "wo", rnd(c.hidden * HD))
            put(l, "w1", rnd(c.intermediate * c.hidden)); put(l, "w3", rnd(c.intermediate * c.hidden))
            put(l, "w2", rnd(c.hidden * c.intermediate))

    def weight_bytes(self):
        c = self.cfg
        HD, KD = c.n_heads * c.head_dim, c.n_kv_heads * c.head_dim
        per_layer = (HD + 2 * KD) * c.hidden + c.hidden * HD + 3 * c.intermediate * c.hidden
        return 2 * (c.n_layers * per_layer + c.vocab * c.hidden)

    def alloc_kv_cache(self, num_blocks):
        _check(lib.mi355_dense_alloc_kv_cache(self.h, num_blocks), "alloc_kv_cache")
        self.num_blocks = num_blocks

    def kv_shape(self):
        c = self.cfg
        if self.kv_layout == KV_FLASH:
            s = (self.num_blocks, c.block_size, c.n_kv_heads, c.head_dim)
            return s, s
        return ((self.num_blocks, c.n_kv_heads, c.head_dim // 8, c.block_size, 8),
                (self.num_blocks, c.n_kv_heads, c.head_dim, c.block_size))

    def kv_download(self, layer):
        out = []
        for which in (0, 1):
            shape = self.kv_shape()[which]
            n = int(np.prod(shape))
            host = np.empty(n, np.uint16)
            ptr = lib.mi355_dense_kv_ptr(self.h, layer, which)
            t = torch.empty(n, dtype=torch.int16, device="cuda")
            torch.cuda.synchronize()
            ctypes.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(ptr),
                                                                 ctypes.c_size_t(n * 2), 3)
            host[:] = t.cpu().numpy().view(np.uint16)
            out.append(host.reshape(shape))
        return out

    def kv_upload(self, layer, k_bits, v_bits):
        for which, bits in ((0, k_bits), (1, v_bits)):
            t = torch.from_numpy(np.ascontiguousarray(bits).view(np.int16).reshape(-1)).cuda()
            ptr = lib.mi355_dense_kv_ptr(self.h, layer, which)
            torch.cuda.synchronize()
            ctypes.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(ctypes.c_void_p(ptr), ctypes.c_void_p(t.data_ptr()),
                                                                 ctypes.c_size_t(t.numel() * 2), 3)
            torch.cuda.synchronize()

    def forward(self, meta, is_prefill=False, stream=None, sync=True):
        """meta: dict from oracle.ops.prepare_decode / prepare_prompt or BlockEngine.prepare_*  -> logits f32"""
        dev = "cuda"
        n = len(meta["context_lens"])
        T = len(meta["input_ids"])
        tok = torch.from_numpy(np.asarray(meta["input_ids"]).astype(np.int64).astype(np.int32)).to(dev)
        pos = torch.from_numpy(np.asarray(meta["positions"]).astype(np.int64)).to(dev)
        slots = torch.from_numpy(np.asarray(meta["slot_mapping"]).astype(np.int64)).to(dev)
        bt = torch.from_numpy(np.asarray(meta["block_tables"]).astype(np.int64).astype(np.int32)).contiguous().to(dev)
        ctx = torch.from_numpy(np.asarray(meta["context_lens"]).astype(np.int64).astype(np.int32)).to(dev)
        cu = None
        if is_prefill:
            cu = torch.from_numpy(np.asarray(meta["cu_seqlens_q"]).astype(np.int64).astype(np.int32)).to(dev)
        logits = torch.empty((n, self.logits_width()),
                             dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        _check(lib.mi355_dense_forward(self.h, tok.data_ptr(), pos.data_ptr(), slots.data_ptr(), bt.data_ptr(),
                                       ctx.data_ptr(), cu.data_ptr() if cu is not None else None, n, T,
                                       int(meta.get("max_seqlen_q", 0)), bt.shape[1], int(meta["max_context_len"]),
                                       logits.data_ptr(), st), "dense_forward")
        if sync:
            torch.cuda.synchronize()
        return logits

    # ---- greedy loop on static device buffers (hipGraph replay): mi355_dense_decode_* = the GGUF layer's decode_begin / step / read
    def finalize(self):
        """freeze the weights (tile re-ordering happens here, not inside the first step)"""
        _check(lib.mi355_dense_finalize(self.h), "dense_finalize")

    def set_graph(self, enable):
        _check(lib.mi355_dense_set_graph(self.h, 1 if enable else 0), "dense_set_graph")

    def decode_begin(self, tokens, seq_lens, block_tables, ctx_cap, stream=0):
        tokens = np.ascontiguousarray(tokens, np.uint32)
        seq_lens = np.ascontiguousarray(seq_lens, np.uint32)
        bt = np.ascontiguousarray(block_tables, np.uint32)
        _check(lib.mi355_dense_decode_begin(self.h, tokens.ctypes.data, seq_lens.ctypes.data, bt.ctypes.data, len(tokens), bt.shape[1],
                                            int(ctx_cap), stream), "dense_decode_begin")
        self._loop_batch = len(tokens)

    def decode_step(self, stream=0):
        _check(lib.mi355_dense_decode_step(self.h, stream), "dense_decode_step")

    def graph_stats(self):
        """(step graphs captured so far, steps of the greedy loop that ran eagerly): mi355_dense_graph_captures / _eager_steps"""
        return int(lib.mi355_dense_graph_captures(self.h)), int(lib.mi355_dense_eager_steps(self.h))

    def read_tokens(self, stream=0):
        out = np.zeros(self._loop_batch, np.uint32)
        _check(lib.mi355_dense_decode_read_tokens(self.h, out.ctypes.data, stream), "dense_decode_read_tokens")
        return out

    def loop_logits(self):
        """f32 [batch, vocab] logits of the loop's last step (a copy)"""
        n = self._loop_batch * self.logits_width()
        out = torch.empty(n, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        ctypes.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(lib.mi355_dense_logits_ptr(self.h)),
                                                             ctypes.c_size_t(n * 4), 3)
        torch.cuda.synchronize()
        return out.view(self._loop_batch, -1)

    def forward_device(self, tok, pos, slots, bt, ctx, max_context_len, logits, stream):
        """decode step over device tensors that stay resident (bench loop)"""
        _check(lib.mi355_dense_forward(self.h, tok.data_ptr(), pos.data_ptr(), slots.data_ptr(), bt.data_ptr(),
                                       ctx.data_ptr(), None, tok.shape[0], tok.shape[0], 0, bt.shape[1],
                                       int(max_context_len), logits.data_ptr(), stream), "dense_forward")
