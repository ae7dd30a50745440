"""N > 1 path on CPU: 2 processes over gloo run the tensor-parallel decode step of the ORACLE with the shard plan
of candle_vllm_amd/tp.py and real all-reduce / all-gather collectives; the result must equal the unsharded
oracle.  (The HIP path uses the same shard plan and RCCL; multi-GPU hardware runs are the driver's.)"""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist                      # noqa: E402
import torch.multiprocessing as mp                    # noqa: E402

from oracle import llama                              # noqa: E402
from oracle import ops as O                           # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class GlooComm:
    def all_reduce(self, a):
        t = torch.from_numpy(np.ascontiguousarray(a, np.float32).copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    def all_gather(self, a):
        t = torch.from_numpy(np.ascontiguousarray(a, np.float32))
        outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(outs, t)
        return [o.numpy() for o in outs]


def _case(moe=False, vocab=512):
    cfg = llama.LlamaConfig.tiny(hidden=512, n_heads=4, n_kv_heads=2, head_dim=128, intermediate=1024, vocab=vocab)
    if moe:
        cfg.n_expert, cfg.n_expert_used = 4, 2
        W = llama.make_moe_weights(cfg, 4, seed=99)
    else:
        W = llama.make_weights(cfg, seed=99)
    rng = np.random.default_rng(5)
    seqs = [{"tokens": [int(t) for t in rng.integers(0, cfg.vocab, 21)], "block_table": [2, 5]},
            {"tokens": [int(t) for t in rng.integers(0, cfg.vocab, 9)], "block_table": [1]}]
    return cfg, W, seqs


def _worker(rank, world, port, q, moe=False, vocab=512):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from candle_vllm_amd import tp
    cfg, W, seqs = _case(moe, vocab)
    lcfg = tp.shard_config(cfg, rank, world)
    lW = tp.shard_weights(W, cfg, rank, world)
    m = llama.OracleLlama(lcfg, lW, comm=GlooComm())
    cache = m.new_cache(8)                              # local kv heads only (cache_engine.rs:307,320,338)
    pre = m.forward(O.prepare_prompt(seqs, cfg.block_size), cache, is_prefill=True)
    for s, row in zip(seqs, pre):
        s["tokens"].append(int(row.argmax()))
    dec = m.forward(O.prepare_decode(seqs, cfg.block_size), cache)
    if rank == 0:
        q.put((pre, dec))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("moe,vocab", [(False, 512), (True, 512), (False, 500)])
def test_tp2_oracle_equals_unsharded(moe, vocab):
    """vocab = 500: not a fixed point of pad_vocab_size (-> 512): rank 1's lm_head shard carries 12 zero rows and the gathered
    logits are narrowed back to 500 columns (VocabParallelLinear, distributed.rs:1596-1616,1657-1660)"""
    import importlib.util
    if importlib.util.find_spec("candle_vllm_amd") is None:
        pytest.skip("package not importable")
    # importing candle_vllm_amd needs the built library (symbol check); build on demand
    import __graft_entry__ as ge
    ge.build()
    cfg, W, seqs = _case(moe, vocab)
    ref_m = llama.OracleLlama(cfg, W)
    cache = ref_m.new_cache(8)
    pre = ref_m.forward(O.prepare_prompt(seqs, cfg.block_size), cache, is_prefill=True)
    for s, row in zip(seqs, pre):
        s["tokens"].append(int(row.argmax()))
    dec = ref_m.forward(O.prepare_decode(seqs, cfg.block_size), cache)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, moe, vocab)) for r in range(2)]
    for p in procs:
        p.start()
    got_pre, got_dec = q.get(timeout=120)
    assert got_pre.shape == pre.shape and got_dec.shape == dec.shape and pre.shape[-1] == vocab
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.abs(got_pre - pre).max() < 1e-5 * np.abs(pre).max()
    assert np.abs(got_dec - dec).max() < 1e-5 * np.abs(dec).max()


def test_shard_plan_shapes_and_replication():
    from candle_vllm_amd import tp
    cfg = llama.LlamaConfig.llama3_8b()
    assert tp.kv_head_shard(8, 3, 8) == (1, 3, 8) and tp.kv_head_shard(4, 5, 8) == (1, 2, 4)
    l = tp.shard_config(cfg, 3, 8)
    assert (l.n_heads, l.n_kv_heads, l.intermediate, l.vocab) == (4, 1, 1792, 16032)
    assert l.intermediate % 256 == 0 and l.vocab % 16 == 0
    with pytest.raises(ValueError):
        tp.shard_config(cfg, 0, 3)


# ---- 16-bit safetensors / GPTQ path (BASELINE config 4: Qwen2 GPTQ, TP=2) -------------------------------------------
class GlooComm16(GlooComm):
    """collectives in the model dtype: gloo has no bf16 sum, so sum in f32 -- exact for two bf16 addends, the
    caller rounds (OracleDenseLlama._row_lin)"""


def _dense_case(gptq):
    from oracle import dense_llama as DL
    cfg = DL.DenseConfig(hidden=256, n_layers=2, n_heads=4, n_kv_heads=2, head_dim=64, intermediate=512, vocab=512,
                         rope_theta=10000.0, max_seq=256, block_size=16, qkv_bias=True)
    W = DL.make_weights(cfg, seed=31)
    if gptq:
        W = DL.quantize_gptq(W, group=128)
    rng = np.random.default_rng(8)
    seqs = [{"tokens": [int(t) for t in rng.integers(0, cfg.vocab, 19)], "block_table": [3, 1]},
            {"tokens": [int(t) for t in rng.integers(0, cfg.vocab, 7)], "block_table": [2]}]
    return cfg, W, seqs


def _dense_run(m, cfg, seqs):
    cache = m.new_cache(8)
    pre = m.forward(O.prepare_prompt(seqs, cfg.block_size), cache, is_prefill=True)
    for s, row in zip(seqs, pre):
        s["tokens"].append(int(row.argmax()))
    return pre, m.forward(O.prepare_decode(seqs, cfg.block_size), cache)


def _dense_worker(rank, world, port, q, gptq):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from candle_vllm_amd import tp
    from oracle import dense_llama as DL
    cfg, W, seqs = _dense_case(gptq)
    m = DL.OracleDenseLlama(tp.shard_dense_config(cfg, rank, world), tp.shard_dense_weights(W, cfg, rank, world),
                            flash_layout=False, comm=GlooComm16())
    out = _dense_run(m, cfg, seqs)
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("gptq", [False, True])
def test_tp2_dense_oracle_close_to_unsharded(gptq):
    """the row-parallel all-reduce rounds each rank's partial product before the sum (distributed.rs:696-711), so
    TP=2 is not bit-identical to TP=1 in 16-bit arithmetic; it must agree to bf16 accumulation noise and pick the
    same greedy tokens."""
    import __graft_entry__ as ge
    ge.build()
    from oracle import dense_llama as DL
    cfg, W, seqs = _dense_case(gptq)
    pre, dec = _dense_run(DL.OracleDenseLlama(cfg, W, flash_layout=False), cfg, seqs)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dense_worker, args=(r, 2, port, q, gptq)) for r in range(2)]
    for p in procs:
        p.start()
    got_pre, got_dec = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got_pre.shape == pre.shape and got_dec.shape == dec.shape
    assert np.abs(got_pre - pre).max() < 3e-2 * np.abs(pre).max()
    assert np.abs(got_dec - dec).max() < 3e-2 * np.abs(dec).max()
    assert (got_pre.argmax(-1) == pre.argmax(-1)).all()


def test_dense_shard_plan_gptq_shapes():
    from candle_vllm_amd import tp
    from oracle import dense_llama as DL
    cfg, W, _ = _dense_case(True)
    l0 = tp.shard_dense_weights(W, cfg, 0, 2)["layers"][0]
    assert l0["wq"]["qweight"].shape == (256 // 8, 128) and l0["wq"]["scales"].shape == (2, 128)      # out shard
    assert l0["wo"]["qweight"].shape == (128 // 8, 256) and l0["wo"]["scales"].shape == (1, 256)      # in shard
    assert l0["w2"]["qweight"].shape == (256 // 8, 256) and l0["bk"].shape == (64,)
    # the shards tile the global tensor
    l1 = tp.shard_dense_weights(W, cfg, 1, 2)["layers"][0]
    assert np.array_equal(np.concatenate([l0["wo"]["qweight"], l1["wo"]["qweight"]], 0), W["layers"][0]["wo"]["qweight"])
    assert np.array_equal(np.concatenate([l0["w1"]["qweight"], l1["w1"]["qweight"]], 1), W["layers"][0]["w1"]["qweight"])
    with pytest.raises(ValueError):
        tp.shard_dense_weights(W, cfg, 0, 4)          # 256/4 = 64 rows < group 128


def test_padded_vocabulary_shards_carry_zero_rows():
    """the lm_head shard of a vocabulary pad_vocab_size pads: rows [rank * local, (rank + 1) * local) of the padded matrix, the
    original blocks verbatim, all-zero blocks (= rows of 0.0 in every k-quant) beyond the real vocabulary"""
    from candle_vllm_amd import tp
    cfg = llama.LlamaConfig.tiny(hidden=512, n_heads=4, n_kv_heads=2, head_dim=128, intermediate=1024, vocab=785)   # -> 832 / 2 ranks
    W = llama.make_weights(cfg, seed=3)
    t, blocks = W["output"]
    assert tp.pad_vocab_size(cfg.vocab, 2) == 832
    got = [tp.shard_weights(W, cfg, r, 2)["output"] for r in range(2)]
    cat = np.concatenate([b for _, b in got], axis=0)
    assert all(tt == t and b.shape[0] == 416 for tt, b in got) and cat.shape[0] == 832
    assert np.array_equal(cat[:785], blocks) and not cat[785:].any()
    lc = tp.shard_config(cfg, 1, 2)
    assert (lc.vocab, lc.vocab_total) == (416, 785)
    # a rank wholly beyond the real vocabulary holds zero rows only
    assert tp.pad_vocab_size(100, 8) == 128
    small = (t, blocks[:100])
    last = tp._rows_padded(small, 7, 8, 100)[1]
    assert last.shape[0] == 16 and not last.any()
    six = tp._rows_padded(small, 6, 8, 100)[1]
    assert np.array_equal(six[:4], blocks[96:100]) and not six[4:].any()    # rows 96..99 are real


def test_pad_vocab_size_formula():
    from candle_vllm_amd import tp
    # BASELINE vocabularies are fixed points for their TP degrees (no padded rows to add)
    assert tp.pad_vocab_size(128256, 8) == 128256 and tp.pad_vocab_size(152064, 2) == 152064 and tp.pad_vocab_size(32000, 8) == 32000
    assert tp.pad_vocab_size(50257, 4) == 50304 and tp.pad_vocab_size(50304, 8) == 50304       # distributed.rs:1446-1452
    assert tp.pad_vocab_size(100, 3) == 192


# ---- the C-level communicator with HOST-supplied collectives (mi355_comm_create_external), 2 ranks over gloo -------------
def _ext_comm_worker(rank, world, port, q):
    try:
        import ctypes
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from candle_vllm_amd import tp
        from candle_vllm_amd._lib import lib

        class HostComm(tp.TorchDistComm):              # same callbacks, payload already in host memory (no GPU here)
            def _d2h(self, ptr, nbytes):
                return np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(ptr)).copy()

            def _h2d(self, ptr, host):
                ctypes.memmove(ptr, host.ctypes.data, host.nbytes)
        comm = HostComm()
        out = {}
        # all-reduce (sum, in place) in the three wire dtypes
        x = (np.arange(1000, dtype=np.float32) * 0.25 + rank * 1000.0)
        buf = x.copy()
        assert lib.mi355_comm_all_reduce(comm.handle, buf.ctypes.data, buf.size, 0, 0) == 0
        out["f32"] = buf
        h = (np.arange(512) * 0.5 + rank).astype(np.float16)
        assert lib.mi355_comm_all_reduce(comm.handle, h.ctypes.data, h.size, 1, 0) == 0
        out["f16"] = h.astype(np.float32)
        b = ((np.arange(256, dtype=np.float32) + rank * 0.5).view(np.uint32) >> 16).astype(np.uint16)   # exact bf16 values
        assert lib.mi355_comm_all_reduce(comm.handle, b.ctypes.data, b.size, 2, 0) == 0
        out["bf16"] = (b.astype(np.uint32) << 16).view(np.float32)
        # all-gather: [W, count] rank-major
        send = np.full(7, float(rank + 1), np.float32)
        recv = np.zeros(7 * world, np.float32)
        assert lib.mi355_comm_all_gather(comm.handle, send.ctypes.data, recv.ctypes.data, send.size, 0, 0) == 0
        out["gather"] = recv
        # bad dtype code is refused before the callback runs; a failing callback surfaces as a non-zero status
        assert lib.mi355_comm_all_reduce(comm.handle, buf.ctypes.data, buf.size, 9, 0) != 0
        comm._dist = None                              # makes the callback raise inside -> it returns 999, never unwinds
        assert lib.mi355_comm_all_reduce(comm.handle, buf.ctypes.data, buf.size, 0, 0) != 0
        q.put((rank, "ok", out))
        dist.barrier()
        lib.mi355_comm_destroy(comm.handle)
        dist.destroy_process_group()
    except BaseException as e:
        q.put((rank, "error", repr(e)))
        raise


def test_external_communicator_callbacks_two_ranks():
    """`mi355_comm_create_external` + `mi355_comm_all_reduce / all_gather` (the entry points a host that owns its own
    communicator binds): argument passing, dtype codes, in-place sum, rank-major gather, error propagation -- through the
    same Python callbacks `test_gpu_tp2.py` uses, with the payload in host memory."""
    import __graft_entry__ as ge
    ge.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ext_comm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(2):
            r = q.get(timeout=180)
            assert r[1] == "ok", r
            res[r[0]] = r[2]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    for rank in (0, 1):
        o = res[rank]
        assert np.array_equal(o["f32"], np.arange(1000, dtype=np.float32) * 0.5 + 1000.0)
        assert np.array_equal(o["f16"], (np.arange(512) * 1.0 + 1.0).astype(np.float16).astype(np.float32))
        want = (np.arange(256, dtype=np.float32) * 2 + 0.5)
        u = want.view(np.uint32)
        want_bf = (((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) << 16).view(np.float32)
        assert np.array_equal(o["bf16"], want_bf)
        assert np.array_equal(o["gather"], np.repeat(np.array([1.0, 2.0], np.float32), 7))
