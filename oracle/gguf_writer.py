"""TEST INFRASTRUCTURE ONLY -- minimal GGUF v3 writer (the published GGUF layout [EXT]; what candle's
`gguf_file::Content::read` parses, src/backend/gguf.rs:48-102): used to produce files for the C++ reader tests.
Tensors: name -> (ggml_type, numpy array in NATIVE block bytes or f32 values, dims slowest-first)."""
import struct

import numpy as np

T_U32, T_F32, T_STRING, T_ARRAY, T_U64 = 4, 6, 8, 9, 10


def _s(b):
    b = b.encode() if isinstance(b, str) else b
    return struct.pack("<Q", len(b)) + b


def write_gguf(path, metadata, tensors, alignment=32, version=3):
    """metadata: dict key -> int (u32) | float (f32) | str | list[str] (array of strings);
    tensors: list of (name, ggml_type, raw_bytes: np.uint8 array, dims slowest-first)"""
    out = bytearray()
    out += struct.pack("<IIQQ", 0x46554747, version, len(tensors), len(metadata))
    for k, v in metadata.items():
        out += _s(k)
        if isinstance(v, bool):
            out += struct.pack("<IB", 7, int(v))
        elif isinstance(v, int):
            out += struct.pack("<II", T_U32, v)
        elif isinstance(v, float):
            out += struct.pack("<If", T_F32, v)
        elif isinstance(v, str):
            out += struct.pack("<I", T_STRING) + _s(v)
        elif isinstance(v, list):
            out += struct.pack("<IIQ", T_ARRAY, T_STRING, len(v))
            for e in v:
                out += _s(e)
        else:
            raise TypeError(k)
    offsets, off = [], 0
    for name, t, raw, dims in tensors:
        offsets.append(off)
        off += (len(raw) + alignment - 1) // alignment * alignment
    for (name, t, raw, dims), o in zip(tensors, offsets):
        out += _s(name)
        out += struct.pack("<I", len(dims))
        for d in reversed(dims):                                   # GGUF stores the fastest dim first
            out += struct.pack("<Q", d)
        out += struct.pack("<IQ", t, o)
    out += b"\0" * ((-len(out)) % alignment)
    for (name, t, raw, dims), o in zip(tensors, offsets):
        out += raw.tobytes()
        out += b"\0" * ((-len(raw)) % alignment)
    with open(path, "wb") as f:
        f.write(out)


def llama_to_gguf(path, cfg, W):
    """oracle.llama weights -> a llama-architecture GGUF file with the tensor names GGUFLLaMa reads
    (quantized_llama.rs:262-371)."""
    from . import kquants as kq
    md = {"general.architecture": "llama", "general.name": "synthetic", "general.alignment": 32,
          "llama.context_length": cfg.max_seq, "llama.embedding_length": cfg.hidden, "llama.block_count": cfg.n_layers,
          "llama.feed_forward_length": cfg.intermediate, "llama.attention.head_count": cfg.n_heads,
          "llama.attention.head_count_kv": cfg.n_kv_heads, "llama.attention.key_length": cfg.head_dim,
          "llama.attention.layer_norm_rms_epsilon": float(cfg.rms_eps), "llama.rope.freq_base": float(cfg.rope_theta),
          "tokenizer.ggml.tokens": ["<a>", "<b>", "<c>"]}
    if any("experts" in lw for lw in W["layers"]):
        md["llama.expert_count"] = len(W["layers"][0]["experts"])
        md["llama.expert_used_count"] = int(getattr(cfg, "n_expert_used", 2) or 2)
    ts = []

    def q(name, tw, rows, cols):
        t, blocks = tw
        ts.append((name, t, np.ascontiguousarray(blocks, np.uint8).reshape(-1), [rows, cols]))

    def f32(name, a):
        a = np.ascontiguousarray(a, np.float32)
        ts.append((name, 0, a.view(np.uint8).reshape(-1), list(a.shape)))
    emb = W["tok_embd"]
    ts.append(("token_embd.weight", kq.GGML_Q6_K, np.ascontiguousarray(kq.quantize(emb, kq.GGML_Q6_K), np.uint8).reshape(-1),
               list(emb.shape)))
    f32("output_norm.weight", W["output_norm"])
    q("output.weight", W["output"], cfg.vocab, cfg.hidden)
    H, Hkv, D, hid, I = cfg.n_heads, cfg.n_kv_heads, cfg.head_dim, cfg.hidden, cfg.intermediate
    for l, lw in enumerate(W["layers"]):
        p = f"blk.{l}."
        q(p + "attn_q.weight", lw["wq"], H * D, hid)
        q(p + "attn_k.weight", lw["wk"], Hkv * D, hid)
        q(p + "attn_v.weight", lw["wv"], Hkv * D, hid)
        q(p + "attn_output.weight", lw["wo"], hid, H * D)
        if "experts" in lw:                                            # quantized_llama.rs:347-365
            f32(p + "ffn_gate_inp.weight", lw["gate_inp"])
            for e, ex in enumerate(lw["experts"]):
                q(p + f"ffn_gate.{e}.weight", ex["w1"], I, hid)
                q(p + f"ffn_down.{e}.weight", ex["w2"], hid, I)
                q(p + f"ffn_up.{e}.weight", ex["w3"], I, hid)
        else:
            q(p + "ffn_gate.weight", lw["w1"], I, hid)
            q(p + "ffn_down.weight", lw["w2"], hid, I)
            q(p + "ffn_up.weight", lw["w3"], I, hid)
        f32(p + "attn_norm.weight", lw["attn_norm"])
        f32(p + "ffn_norm.weight", lw["ffn_norm"])
    write_gguf(path, md, ts)
    return md, ts
