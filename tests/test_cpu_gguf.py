"""GGUF reader (C++ behind the C ABI) against files produced by the test writer: header, metadata, tensor
directory (names, shapes slowest-first, types, sizes) and the mmap'ed tensor bytes, bit-exact."""
import ctypes
import os

import numpy as np
import pytest

from oracle import gguf_writer as GW
from oracle import kquants as kq
from oracle import llama


def test_reader_roundtrip(lib, tmp_path):
    cfg = llama.LlamaConfig.tiny()
    W = llama.make_weights(cfg, seed=1234)
    path = os.path.join(tmp_path, "tiny.gguf")
    md, ts = GW.llama_to_gguf(path, cfg, W)
    g = lib.mi355_gguf_open(path.encode())
    assert g
    try:
        assert lib.mi355_gguf_version(g) == 3
        assert lib.mi355_gguf_n_tensors(g) == len(ts)
        buf = ctypes.create_string_buffer(64)
        assert lib.mi355_gguf_get_str(g, b"general.architecture", buf, 64) == 5 and buf.value == b"llama"
        u = ctypes.c_uint64(0)
        for key, want in (("llama.block_count", cfg.n_layers), ("llama.attention.head_count_kv", cfg.n_kv_heads),
                          ("llama.embedding_length", cfg.hidden)):
            assert lib.mi355_gguf_get_u64(g, key.encode(), ctypes.addressof(u)) == 1 and u.value == want
        f = ctypes.c_double(0)
        assert lib.mi355_gguf_get_f64(g, b"llama.rope.freq_base", ctypes.addressof(f)) == 1
        assert abs(f.value - cfg.rope_theta) < 1e-3
        assert lib.mi355_gguf_get_u64(g, b"no.such.key", ctypes.addressof(u)) == 0
        for name, t, raw, dims in ts:
            i = lib.mi355_gguf_find(g, name.encode())
            assert i >= 0, name
            d4 = (ctypes.c_int64 * 4)()
            nd, ty, nb = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_uint64(0)
            assert lib.mi355_gguf_tensor_info(g, i, None, 0, d4, ctypes.addressof(nd), ctypes.addressof(ty),
                                              ctypes.addressof(nb)) == 0
            assert nd.value == len(dims) and list(d4)[:len(dims)] == list(dims) and ty.value == t
            assert nb.value == raw.size
            ptr = lib.mi355_gguf_tensor_data(g, i)
            got = np.ctypeslib.as_array((ctypes.c_uint8 * raw.size).from_address(ptr))
            assert np.array_equal(got, raw), name
        assert lib.mi355_gguf_find(g, b"blk.99.attn_q.weight") == -1
    finally:
        lib.mi355_gguf_close(g)


def test_reader_rejects_garbage(lib, tmp_path):
    p = os.path.join(tmp_path, "bad.gguf")
    with open(p, "wb") as f:
        f.write(b"GGML" + b"\0" * 64)
    assert not lib.mi355_gguf_open(p.encode())
    assert not lib.mi355_gguf_open(os.path.join(tmp_path, "missing.gguf").encode())
    # truncated data section
    cfg = llama.LlamaConfig.tiny()
    W = llama.make_weights(cfg, seed=1)
    good = os.path.join(tmp_path, "t.gguf")
    GW.llama_to_gguf(good, cfg, W)
    data = open(good, "rb").read()
    with open(p, "wb") as f:
        f.write(data[: len(data) // 2])
    assert not lib.mi355_gguf_open(p.encode())


_FILE_NAMES = {"wq": "attn_q", "wk": "attn_k", "wv": "attn_v", "wo": "attn_output", "w1": "ffn_gate", "w2": "ffn_down",
               "w3": "ffn_up"}
_DIM = {"wq": 0, "wk": 0, "wv": 0, "wo": 1, "w1": 0, "w2": 1, "w3": 0}


def _shard(lib, g, name, dim, rank, world):
    i = lib.mi355_gguf_find(g, name.encode())
    assert i >= 0, name
    n = lib.mi355_gguf_tensor_shard(g, i, dim, rank, world, None, 0)
    if n < 0:
        return n
    buf = np.empty(n, np.uint8)
    assert lib.mi355_gguf_tensor_shard(g, i, dim, rank, world, buf.ctypes.data, n) == n
    assert lib.mi355_gguf_tensor_shard(g, i, dim, rank, world, buf.ctypes.data, n - 1) == -3
    return buf


@pytest.mark.parametrize("world,n_kv", [(2, 2), (4, 2), (2, 4)])
def test_tensor_shard_equals_python_shard_plan(lib, tmp_path, world, n_kv):
    """raw byte-range TP shards of the file's tensors (get_sharded, quantized_var_builder.rs:135-183) == the shard plan
    of candle_vllm_amd/tp.py applied to the same blocks (the plan the 2-rank oracle test runs), incl. replicated kv
    heads when Hkv < W; the dequantised shard == the slice of the dequantised tensor"""
    from candle_vllm_amd import tp
    cfg = llama.LlamaConfig.tiny(hidden=1024, n_heads=4, n_kv_heads=n_kv, head_dim=256, intermediate=1024, vocab=512)
    W = llama.make_weights(cfg, seed=77)
    path = os.path.join(tmp_path, "tp.gguf")
    GW.llama_to_gguf(path, cfg, W)
    g = lib.mi355_gguf_open(path.encode())
    assert g
    try:
        for rank in range(world):
            lW = tp.shard_weights(W, cfg, rank, world)
            _, kv_rank, kv_world = tp.kv_head_shard(cfg.n_kv_heads, rank, world)
            got = _shard(lib, g, "output.weight", 0, rank, world)
            assert np.array_equal(got, np.ascontiguousarray(lW["output"][1]).reshape(-1))
            for l in range(cfg.n_layers):
                for key, fname in _FILE_NAMES.items():
                    r, w = (kv_rank, kv_world) if key in ("wk", "wv") else (rank, world)
                    got = _shard(lib, g, f"blk.{l}.{fname}.weight", _DIM[key], r, w)
                    t, want = lW["layers"][l][key]
                    assert np.array_equal(got, np.ascontiguousarray(want).reshape(-1)), (rank, l, key)
                    # value check through the dequantiser: shard of the matrix == matrix of the shard
                    full = kq.dequantize(W["layers"][l][key][1], t)
                    part = kq.dequantize(got.reshape(want.shape), t)
                    rows, cols = full.shape
                    sl = (slice(r * rows // w, (r + 1) * rows // w), slice(None)) if _DIM[key] == 0 else \
                         (slice(None), slice(r * cols // w, (r + 1) * cols // w))
                    assert np.array_equal(part, full[sl]), (rank, l, key)
    finally:
        lib.mi355_gguf_close(g)


def test_tensor_shard_error_codes(lib, tmp_path):
    cfg = llama.LlamaConfig.tiny(hidden=256, n_heads=2, n_kv_heads=2, head_dim=128, intermediate=768, vocab=384)
    W = llama.make_weights(cfg, seed=3)
    path = os.path.join(tmp_path, "e.gguf")
    GW.llama_to_gguf(path, cfg, W)
    g = lib.mi355_gguf_open(path.encode())
    assert g
    try:
        i_down = lib.mi355_gguf_find(g, b"blk.0.ffn_down.weight")        # [256, 768]: 3 k-blocks per row
        i_q = lib.mi355_gguf_find(g, b"blk.0.attn_q.weight")
        i_norm = lib.mi355_gguf_find(g, b"blk.0.attn_norm.weight")       # F32 vector [256]
        assert lib.mi355_gguf_tensor_shard(g, i_down, 1, 0, 2, None, 0) == -2     # 384 columns would cut a 256-block
        assert lib.mi355_gguf_tensor_shard(g, i_down, 1, 0, 3, None, 0) == 256 * 144 or \
            lib.mi355_gguf_tensor_shard(g, i_down, 1, 0, 3, None, 0) == 256 * 210
        assert lib.mi355_gguf_tensor_shard(g, i_q, 0, 0, 3, None, 0) == -1        # 256 rows / 3
        assert lib.mi355_gguf_tensor_shard(g, i_q, 0, 2, 2, None, 0) == -1        # rank out of range
        assert lib.mi355_gguf_tensor_shard(g, i_q, 2, 0, 2, None, 0) == -1        # no such dimension
        assert lib.mi355_gguf_tensor_shard(g, 9999, 0, 0, 2, None, 0) == -1
        assert lib.mi355_gguf_tensor_shard(g, i_norm, 0, 1, 2, None, 0) == 128 * 4
        half = np.empty(128, np.float32)
        assert lib.mi355_gguf_tensor_shard(g, i_norm, 0, 1, 2, half.ctypes.data, 512) == 512
        assert np.array_equal(half, np.asarray(W["layers"][0]["attn_norm"], np.float32)[128:])
        # world 1 = the whole tensor
        nb = ctypes.c_uint64(0)
        lib.mi355_gguf_tensor_info(g, i_q, None, 0, None, None, None, ctypes.addressof(nb))
        assert lib.mi355_gguf_tensor_shard(g, i_q, 0, 0, 1, None, 0) == nb.value
    finally:
        lib.mi355_gguf_close(g)


def test_load_gguf_tp_rejects_unshardable_files_before_touching_the_gpu(lib, tmp_path):
    """the shard-plan checks of `mi355_llama_load_gguf_tp` run on the host, ahead of any device call: heads that do
    not divide (attention.rs:553-554), kv heads that neither divide nor replicate (distributed.rs:744-760), a
    vocabulary `pad_vocab_size` would pad (needs the re-quantising fallback: hipErrorNotSupported = 801)"""
    from candle_vllm_amd import tp
    from candle_vllm_amd._lib import LlamaConfig
    h, c = ctypes.c_void_p(0), LlamaConfig()

    def load(path, rank, world):
        return lib.mi355_llama_load_gguf_tp(path.encode(), 1, 8, 16, 1, 0, rank, world, ctypes.addressof(h), ctypes.addressof(c))

    cfg = llama.LlamaConfig.tiny(hidden=256, n_heads=2, n_kv_heads=2, head_dim=128, intermediate=512, vocab=336)
    p1 = os.path.join(tmp_path, "v336.gguf")
    GW.llama_to_gguf(p1, cfg, llama.make_weights(cfg, seed=1))
    assert tp.pad_vocab_size(336, 2) != 336
    assert load(p1, 0, 2) == 801 and not h.value
    assert load(p1, 2, 2) == 1 and load(p1, 0, 0) == 1            # rank / world out of range
    assert load(p1, 0, 3) == 1                                    # 2 heads over 3 ranks
    cfg = llama.LlamaConfig.tiny(hidden=768, n_heads=6, n_kv_heads=3, head_dim=128, intermediate=512, vocab=384)
    p2 = os.path.join(tmp_path, "kv3.gguf")
    GW.llama_to_gguf(p2, cfg, llama.make_weights(cfg, seed=2))
    assert load(p2, 0, 2) == 1                                    # 3 kv heads over 2 ranks: neither split nor replicated
    with pytest.raises(ValueError):
        tp.kv_head_shard(3, 0, 2)
