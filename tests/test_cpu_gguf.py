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
