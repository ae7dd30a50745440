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


def test_requantised_shard_equals_the_restated_fallback_bit_for_bit(lib, tmp_path):
    """a dim-1 shard that cuts a k-quant block: `get_sharded_no_shape`'s fallback (quantized_var_builder.rs:234-269) --
    dequantize_f16 -> narrow -> Q8_0 -- as the loader does it on the host (mi355_gguf_tensor_shard_q8_0) against the numpy
    restatement, byte for byte; Q4_K and Q6_K sources; and the shard plan of candle_vllm_amd/tp.py uses the same bytes"""
    from candle_vllm_amd import tp
    cfg = llama.LlamaConfig.tiny(hidden=256, n_heads=6, n_kv_heads=2, head_dim=128, intermediate=768, vocab=384)
    W = llama.make_weights(cfg, seed=5)                            # wo [256, 768], w2 [256, 768]: 384 columns per rank of 2
    path = os.path.join(tmp_path, "q8.gguf")
    GW.llama_to_gguf(path, cfg, W)
    g = lib.mi355_gguf_open(path.encode())
    assert g
    try:
        seen = set()
        for l in range(cfg.n_layers):
            for name, key in (("attn_output", "wo"), ("ffn_down", "w2")):
                i = lib.mi355_gguf_find(g, f"blk.{l}.{name}.weight".encode())
                t, blocks = W["layers"][l][key]
                seen.add(t)
                assert lib.mi355_gguf_tensor_shard(g, i, 1, 0, 2, None, 0) == -2
                for rank in (0, 1):
                    n = lib.mi355_gguf_tensor_shard_q8_0(g, i, rank, 2, None, 0)
                    assert n == 256 * (384 // 32) * 34
                    out = np.empty(n, np.uint8)
                    assert lib.mi355_gguf_tensor_shard_q8_0(g, i, rank, 2, out.ctypes.data, n) == n
                    want = kq.requantize_shard_q8_0(blocks, t, rank, 2)
                    assert np.array_equal(out.reshape(want.shape), want), (l, name, rank)
                    st, sb = tp._cols((t, blocks), rank, 2, kq.requantize_shard_q8_0)
                    assert st == kq.GGML_Q8_0 and np.array_equal(sb, want)
                assert lib.mi355_gguf_tensor_shard_q8_0(g, i, 0, 2, out.ctypes.data, n - 1) == -3
        assert seen == {kq.GGML_Q4_K, kq.GGML_Q6_K}                 # the Q4_K_M mixture puts both types on these tensors
        i_q = lib.mi355_gguf_find(g, b"blk.0.attn_q.weight")
        assert lib.mi355_gguf_tensor_shard_q8_0(g, i_q, 2, 2, None, 0) == -1       # rank out of range
        i_norm = lib.mi355_gguf_find(g, b"blk.0.attn_norm.weight")
        assert lib.mi355_gguf_tensor_shard_q8_0(g, i_norm, 0, 2, None, 0) == -1    # not a quantised matrix
    finally:
        lib.mi355_gguf_close(g)
    # ggml's rounding: half away from zero with the UNROUNDED scale
    x = np.zeros((1, 32), np.float32)
    x[0, 0], x[0, 1], x[0, 2] = 127.0, 0.5, -2.5
    q = kq.quantize_q8_0_ggml(x)[0, 0, 2:].view(np.int8)
    assert q[0] == 127 and q[1] == 1 and q[2] == -3


def test_load_gguf_tp_rejects_unshardable_files_before_touching_the_gpu(lib, tmp_path):
    """the shard-plan checks of `mi355_llama_load_gguf_tp` run on the host, ahead of any device call: heads that do
    not divide (attention.rs:553-554), kv heads that neither divide nor replicate (distributed.rs:744-760).  (A
    vocabulary `pad_vocab_size` pads is NOT refused: its lm_head shards carry zero rows -- the reader test below.)"""
    from candle_vllm_amd import tp
    from candle_vllm_amd._lib import LlamaConfig
    h, c = ctypes.c_void_p(0), LlamaConfig()

    def load(path, rank, world):
        return lib.mi355_llama_load_gguf_tp(path.encode(), 1, 8, 16, 1, 0, rank, world, ctypes.addressof(h), ctypes.addressof(c))

    cfg = llama.LlamaConfig.tiny(hidden=256, n_heads=2, n_kv_heads=2, head_dim=128, intermediate=512, vocab=336)
    p1 = os.path.join(tmp_path, "v336.gguf")
    GW.llama_to_gguf(p1, cfg, llama.make_weights(cfg, seed=1))
    assert tp.pad_vocab_size(336, 2) != 336
    assert load(p1, 2, 2) == 1 and load(p1, 0, 0) == 1            # rank / world out of range
    assert load(p1, 0, 3) == 1                                    # 2 heads over 3 ranks
    cfg = llama.LlamaConfig.tiny(hidden=768, n_heads=6, n_kv_heads=3, head_dim=128, intermediate=512, vocab=384)
    p2 = os.path.join(tmp_path, "kv3.gguf")
    GW.llama_to_gguf(p2, cfg, llama.make_weights(cfg, seed=2))
    assert load(p2, 0, 2) == 1                                    # 3 kv heads over 2 ranks: neither split nor replicated
    with pytest.raises(ValueError):
        tp.kv_head_shard(3, 0, 2)


def test_reader_rows_padded_gives_the_padded_vocabulary_shards(lib, tmp_path):
    """`mi355_gguf_tensor_rows_padded`: the vocab-parallel lm_head shard of a vocabulary `pad_vocab_size` pads (336 -> 384 over two
    ranks): the file's own blocks for the real rows, all-zero blocks beyond them -- the same shards candle_vllm_amd/tp.py cuts"""
    from candle_vllm_amd import tp
    cfg = llama.LlamaConfig.tiny(hidden=256, n_heads=2, n_kv_heads=2, head_dim=128, intermediate=512, vocab=336)
    W = llama.make_weights(cfg, seed=1)
    path = os.path.join(tmp_path, "v336.gguf")
    GW.llama_to_gguf(path, cfg, W)
    g = lib.mi355_gguf_open(path.encode())
    assert g
    try:
        i = lib.mi355_gguf_find(g, b"output.weight")
        assert i >= 0
        local = tp.pad_vocab_size(336, 2) // 2
        assert local == 192
        for rank in range(2):
            want = tp._rows_padded(W["output"], rank, 2, 336)[1]
            n = lib.mi355_gguf_tensor_rows_padded(g, i, rank * local, local, None, 0)
            assert n == want.nbytes
            buf = np.full(n, 0xAB, np.uint8)
            assert lib.mi355_gguf_tensor_rows_padded(g, i, rank * local, local, buf.ctypes.data, n) == n
            assert np.array_equal(buf, want.reshape(-1))
            assert lib.mi355_gguf_tensor_rows_padded(g, i, rank * local, local, buf.ctypes.data, n - 1) == -3
        assert not np.frombuffer(tp._rows_padded(W["output"], 1, 2, 336)[1][336 - 192:].tobytes(), np.uint8).any()
        # a window wholly beyond the tensor: zero blocks only; bad arguments
        n = lib.mi355_gguf_tensor_rows_padded(g, i, 1000, 4, None, 0)
        buf = np.full(n, 0xAB, np.uint8)
        assert lib.mi355_gguf_tensor_rows_padded(g, i, 1000, 4, buf.ctypes.data, n) == n and not buf.any()
        assert lib.mi355_gguf_tensor_rows_padded(g, i, -1, 4, None, 0) == -1
        assert lib.mi355_gguf_tensor_rows_padded(g, i, 0, 0, None, 0) == -1
        assert lib.mi355_gguf_tensor_rows_padded(g, lib.mi355_gguf_n_tensors(g), 0, 4, None, 0) == -1
        j = lib.mi355_gguf_find(g, b"output_norm.weight")                      # 1-D tensor
        assert lib.mi355_gguf_tensor_rows_padded(g, j, 0, 4, None, 0) == -1
    finally:
        lib.mi355_gguf_close(g)


def _walk(lib, g):
    """touch everything the reader exposes for an opened file (under the sanitizer run this is the bounds check)"""
    n = lib.mi355_gguf_n_tensors(g)
    d4 = (ctypes.c_int64 * 4)()
    nd, ty, nb = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_uint64(0)
    name = ctypes.create_string_buffer(256)
    total = 0
    for i in range(n):
        assert lib.mi355_gguf_tensor_info(g, i, name, 256, d4, ctypes.addressof(nd), ctypes.addressof(ty), ctypes.addressof(nb)) == 0
        ptr = lib.mi355_gguf_tensor_data(g, i)
        if nb.value:
            raw = np.ctypeslib.as_array((ctypes.c_uint8 * nb.value).from_address(ptr))
            total += int(raw[0]) + int(raw[-1])                     # first and last byte of the claimed range are mapped
        for dim in (0, 1):
            for world in (1, 2):
                k = lib.mi355_gguf_tensor_shard(g, i, dim, world - 1, world, None, 0)
                if k > 0:
                    buf = np.empty(k, np.uint8)
                    assert lib.mi355_gguf_tensor_shard(g, i, dim, world - 1, world, buf.ctypes.data, k) == k
    return total


def test_reader_survives_corrupted_and_hostile_files(lib, tmp_path):
    """every length, count, dimension and offset in a GGUF file is untrusted: random byte corruption, truncation at
    every region, and hand-made hostile headers (string length 2^64-1, dims whose product wraps, offsets past the end,
    zero / absurd alignment) must be refused or read within bounds -- never crash, never read outside the mapping"""
    import struct
    cfg = llama.LlamaConfig.tiny()
    W = llama.make_weights(cfg, seed=5)
    good = os.path.join(tmp_path, "good.gguf")
    GW.llama_to_gguf(good, cfg, W)
    data = bytearray(open(good, "rb").read())
    g = lib.mi355_gguf_open(good.encode())
    assert g
    _walk(lib, g)
    lib.mi355_gguf_close(g)
    header_end = data.find(b"blk.0.attn_q.weight")          # somewhere inside the tensor directory
    assert header_end > 0
    rng = np.random.default_rng(2024)
    p = os.path.join(tmp_path, "fuzz.gguf")
    opened = 0
    for trial in range(300):
        d = bytearray(data)
        kind = trial % 3
        if kind == 0:                                        # flip bytes in the header / directory
            for _ in range(int(rng.integers(1, 6))):
                d[int(rng.integers(0, header_end + 4096))] = int(rng.integers(0, 256))
        elif kind == 1:                                      # overwrite an aligned u64 with an extreme value
            off = int(rng.integers(1, (header_end + 4096) // 8)) * 8
            extremes = [0, 1, 2**31, 2**32, 2**63, 2**64 - 1, len(data), len(data) + 1]
            d[off:off + 8] = struct.pack("<Q", extremes[int(rng.integers(0, len(extremes)))])
        else:                                                # truncate
            d = d[: int(rng.integers(0, len(d)))]
        open(p, "wb").write(bytes(d))
        g = lib.mi355_gguf_open(p.encode())
        if g:
            opened += 1
            _walk(lib, g)
            lib.mi355_gguf_close(g)
    assert opened < 300                                      # most corruptions must be caught

    def header(n_tensors, n_kv, body):
        return struct.pack("<IIQQ", 0x46554747, 3, n_tensors, n_kv) + body

    def s(b):
        return struct.pack("<Q", len(b)) + b
    hostile = {
        "string length 2^64-1": header(0, 1, struct.pack("<Q", 2**64 - 1) + b"abc"),
        "kv count 2^40": header(0, 2**40, b""),
        "tensor count past the file": header(5, 0, b""),
        "dims whose product wraps": header(1, 0, s(b"t") + struct.pack("<IQQIQ", 2, 2**32, 2**32, 0, 0) + b"\0" * 64),
        "bytes-per-block product wraps": header(1, 0, s(b"t") + struct.pack("<IQIQ", 1, 2**62, 0, 0) + b"\0" * 64),
        "offset past the end": header(1, 0, s(b"t") + struct.pack("<IQIQ", 1, 8, 0, 2**40) + b"\0" * 64),
        "offset + size wraps": header(1, 0, s(b"t") + struct.pack("<IQIQ", 1, 8, 0, 2**64 - 8) + b"\0" * 64),
        "five dimensions": header(1, 0, s(b"t") + struct.pack("<I", 5) + b"\0" * 128),
        "unknown ggml type": header(1, 0, s(b"t") + struct.pack("<IQIQ", 1, 32, 999, 0) + b"\0" * 64),
        "zero alignment": header(0, 1, s(b"general.alignment") + struct.pack("<II", 4, 0)),
        "alignment 2^32": header(0, 1, s(b"general.alignment") + struct.pack("<IQ", 10, 2**32)),
        "array of 2^60 bytes": header(0, 1, s(b"a") + struct.pack("<IIQ", 9, 0, 2**60) + b"\1" * 16),
        "nested array": header(0, 1, s(b"a") + struct.pack("<IIQIQ", 9, 9, 1, 0, 1) + b"\1"),
        "unknown value type": header(0, 1, s(b"a") + struct.pack("<I", 77) + b"\0" * 16),
        "too short": b"GGUF\x03",
        "empty": b"",
    }
    for what, blob in hostile.items():
        open(p, "wb").write(blob)
        assert not lib.mi355_gguf_open(p.encode()), what
    # accessors on a null handle answer instead of crashing
    u = ctypes.c_uint64(0)
    assert lib.mi355_gguf_find(None, b"x") == -1 and lib.mi355_gguf_get_u64(None, b"k", ctypes.addressof(u)) == 0
    assert lib.mi355_gguf_n_tensors(None) == -1 and lib.mi355_gguf_tensor_shard(None, 0, 0, 0, 1, None, 0) == -1
    assert not lib.mi355_gguf_tensor_data(None, 0) and not lib.mi355_gguf_open(None)
    lib.mi355_gguf_close(None)


def _check(lib, path, rank=0, world=1):
    from candle_vllm_amd._lib import LlamaConfig
    c = LlamaConfig()
    rc = lib.mi355_llama_check_gguf(path.encode(), rank, world, ctypes.addressof(c))
    return rc, c


def test_check_gguf_accepts_good_files_and_reports_the_global_config(lib, tmp_path):
    """host-only dry run of `GGUFLLaMa::from_gguf` (quantized_llama.rs:203-420): what the loader would build"""
    cfg = llama.LlamaConfig.tiny(hidden=1024, n_heads=4, n_kv_heads=2, head_dim=256, intermediate=1024, vocab=512)
    p = os.path.join(tmp_path, "ok.gguf")
    GW.llama_to_gguf(p, cfg, llama.make_weights(cfg, seed=3))
    for rank, world in ((0, 1), (0, 2), (1, 2), (3, 4)):           # world 4: 2 kv heads replicated on pairs of ranks
        rc, c = _check(lib, p, rank, world)
        assert rc == 0, (rank, world, rc)
        assert (c.hidden, c.n_layers, c.n_heads, c.n_kv_heads, c.head_dim, c.intermediate, c.vocab) == \
            (cfg.hidden, cfg.n_layers, cfg.n_heads, cfg.n_kv_heads, cfg.head_dim, cfg.intermediate, cfg.vocab)
        assert (c.tp_rank, c.tp_world, c.n_expert) == (rank, world, 0)
        assert abs(c.rope_theta - cfg.rope_theta) < 1e-3 * cfg.rope_theta and abs(c.rms_eps - cfg.rms_eps) < 1e-9
    assert _check(lib, p, 0, 8)[0] == 1                            # 4 heads over 8 ranks
    odd = llama.LlamaConfig.tiny(hidden=256, n_heads=2, n_kv_heads=2, head_dim=128, intermediate=512, vocab=333)
    p3 = os.path.join(tmp_path, "odd.gguf")
    GW.llama_to_gguf(p3, odd, llama.make_weights(odd, seed=8))
    rc, c = _check(lib, p3)
    assert rc == 0 and c.vocab == 333                              # a vocabulary that is not a multiple of the 16-row tile
    moe = llama.LlamaConfig.tiny()
    moe.n_expert, moe.n_expert_used = 4, 2
    p2 = os.path.join(tmp_path, "moe.gguf")
    GW.llama_to_gguf(p2, moe, llama.make_moe_weights(moe, 4, seed=4))
    rc, c = _check(lib, p2)
    assert rc == 0 and (c.n_expert, c.n_expert_used, c.intermediate) == (4, 2, moe.intermediate)


def test_check_gguf_refuses_inconsistent_files(lib, tmp_path):
    """a file whose tensors disagree with its own metadata must be refused on the host (the kernels trust the shapes)"""
    cfg = llama.LlamaConfig.tiny(hidden=512, n_heads=4, n_kv_heads=2, head_dim=128, intermediate=768, vocab=512)
    W = llama.make_weights(cfg, seed=6)
    good = os.path.join(tmp_path, "g.gguf")
    md, ts = GW.llama_to_gguf(good, cfg, W)
    assert _check(lib, good)[0] == 0
    assert _check(lib, good, 0, 2)[0] == 0                         # ffn_down: 768 / 2 = 384 columns cut a 256-block: the Q8_0 fallback
    cfg5 = llama.LlamaConfig.tiny(hidden=512, n_heads=4, n_kv_heads=2, head_dim=128, intermediate=1280, vocab=512)
    five = os.path.join(tmp_path, "five.gguf")
    GW.llama_to_gguf(five, cfg5, llama.make_weights(cfg5, seed=6))
    assert _check(lib, five, 0, 5)[0] != 0                         # 4 heads over 5 ranks
    p = os.path.join(tmp_path, "bad.gguf")

    def variant(md2=None, drop=None, retype=None, reshape=None):
        m = dict(md)
        m.update(md2 or {})
        t2 = []
        for name, t, raw, dims in ts:
            if name == drop:
                continue
            if name == retype:
                t = 8                                              # Q8_0: a valid GGUF type this build does not serve
                raw = raw[: (dims[0] * dims[1] // 32) * 34] if len(raw) >= (dims[0] * dims[1] // 32) * 34 else \
                    np.zeros((dims[0] * dims[1] // 32) * 34, np.uint8)
            if reshape and name == reshape[0]:
                dims = reshape[1]
            t2.append((name, t, raw, dims))
        GW.write_gguf(p, m, t2)
        return _check(lib, p)[0]
    assert variant(md2={"llama.attention.head_count": 8}) == 1                     # attn_q has 4 * 128 rows, not 8 * 128
    assert variant(md2={"llama.attention.head_count_kv": 4}) == 1
    assert variant(md2={"llama.embedding_length": 1024}) == 1
    assert variant(md2={"llama.block_count": cfg.n_layers + 1}) == 1               # a layer's tensors are missing
    assert variant(md2={"llama.attention.head_count": 0}) == 1                     # would divide by zero
    assert variant(md2={"llama.block_count": 2**31 - 1}) == 1
    assert variant(drop="blk.1.ffn_up.weight") == 1
    assert variant(drop="output_norm.weight") == 1
    assert variant(retype="blk.0.attn_v.weight") == 801
    assert variant(md2={"llama.rope.dimension_count": cfg.head_dim}) == 0         # full rotary, stated explicitly
    assert variant(md2={"llama.rope.dimension_count": cfg.head_dim // 2}) == 801  # partial rotary: not served by the GGUF step
    # a Qwen2-style file (qkv bias) or a q/k-norm file: the GGUF step would ignore those tensors -> refused
    for extra in ("blk.1.attn_q.bias", "blk.0.attn_k_norm.weight"):
        GW.write_gguf(p, md, ts + [(extra, 0, np.zeros(cfg.head_dim * 4, np.uint8), [cfg.head_dim])])
        assert _check(lib, p)[0] == 801, extra
    q = next(x for x in ts if x[0] == "blk.0.attn_output.weight")
    assert variant(reshape=("blk.0.attn_output.weight", [q[3][0] // 2, q[3][1] * 2])) == 1    # same bytes, wrong shape
    assert lib.mi355_llama_check_gguf(None, 0, 1, None) == 1
    assert lib.mi355_llama_check_gguf(os.path.join(tmp_path, "none.gguf").encode(), 0, 1, None) != 0
