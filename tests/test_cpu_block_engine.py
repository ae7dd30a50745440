"""The reference's own unit tests for the block manager, restated over the C ABI (these are the pinned fixtures of
SURVEY.md section 8c for row f1): src/scheduler/block_engine.rs:1476-1752, prefix_cache.rs:386-599,
sequence.rs:479-537 -- same scenarios, same asserted values.  Plus: input preparation from the engine state is
bit-identical to the numpy restatement of pipelines/inputs.rs, swap round trips, copy-on-write."""
import numpy as np
import pytest

from oracle import ops as O


@pytest.fixture(scope="module")
def be(lib):
    from candle_vllm_amd import block_engine
    return block_engine


# ----------------------------------------------------------------------------- block_engine.rs tests
def test_allocate_with_prefix_cache_reuses_blocks(be):                         # block_engine.rs:1537-1584
    eng = be.BlockEngine(4, 8, 8, True, 4)
    seq1 = eng.new_sequence(1, [1, 2, 3, 4, 5, 6, 7, 8])
    free_before = eng.get_num_free_blocks()
    eng.allocate([seq1])
    free_after_alloc = eng.get_num_free_blocks()
    assert free_after_alloc < free_before
    cached_ids = eng.block_table(seq1)[:2]
    eng.cache_sequence(seq1)
    eng.free_sequence(seq1)
    assert eng.get_num_free_blocks() == free_after_alloc + 1
    seq2 = eng.new_sequence(2, list(range(1, 13)))
    eng.allocate([seq2])
    assert seq2.get_num_cached_tokens() == 8
    table = eng.block_table(seq2)
    assert table[0] == cached_ids[0] and table[1] == cached_ids[1]


def test_prefix_cache_eviction_does_not_free_active_sequence_blocks(be):       # :1586-1633
    eng = be.BlockEngine(4, 8, 8, True, 4)
    seq1 = eng.new_sequence(1, [1, 2, 3, 4, 5, 6, 7, 8])
    eng.allocate([seq1])
    eng.cache_sequence(seq1)
    eng.free_sequence(seq1)
    seq2 = eng.new_sequence(2, list(range(1, 13)))
    eng.allocate([seq2])
    active = eng.block_table(seq2)[:2]
    assert eng.evict_prefix_cache_blocks(2) == 2
    free_ids = eng.free_block_ids()
    for b in active:
        assert b not in free_ids


def test_append_token_slot_repairs_table_after_skipped_boundary_allocation(be):  # :1635-1676
    eng = be.BlockEngine(4, 4, 4, False, 0)
    seq = eng.new_sequence(1, [1, 2, 3, 4])
    eng.allocate([seq])
    eng._pop_back_block(seq)
    seq.add_token(5)
    assert seq.get_logical_token_blocks() == 2
    assert len(eng.block_table(seq)) == 1
    assert eng.can_append_token_to_seq([seq])
    assert eng.append_token_slot_to_seq(seq) is None
    assert len(eng.block_table(seq)) == 2


def test_allocate_for_prefill_reserves_and_extends_by_chunk(be):               # :1678-1713
    bs = 4
    eng = be.BlockEngine(bs, 4, 4, False, 0)
    seq = eng.new_sequence(1, list(range(1, 11)))
    assert eng.can_allocate_for_prefill([seq], bs) == be.OK
    eng.allocate_for_prefill([seq], bs)
    assert len(eng.block_table(seq)) == 1
    seq.set_num_cached_tokens(4)
    assert eng.prefill_chunk_blocks_required([seq], bs) == 1
    assert eng.can_append_prefill_chunk_to_seq_group([seq], bs)
    eng.append_prefill_chunk_slots_to_seq_group([seq], bs)
    assert len(eng.block_table(seq)) == 2
    seq.set_num_cached_tokens(8)
    eng.append_prefill_chunk_slots_to_seq_group([seq], bs)
    assert len(eng.block_table(seq)) == 3


def test_rebuild_sequence_with_cached_prefix_shrinks_cached_tokens(be):        # :1715-1751
    eng = be.BlockEngine(4, 8, 8, True, 8)
    seq = eng.new_sequence(1, list(range(1, 13)))
    eng.allocate([seq])
    seq.set_num_cached_tokens(8)
    orig = eng.block_table(seq)
    assert eng.rebuild_sequence_with_cached_prefix(seq, 4)
    rebuilt = eng.block_table(seq)
    assert len(rebuilt) == len(orig)
    assert rebuilt[0] == orig[0]
    assert seq.get_num_cached_tokens() == 4
    assert seq.has_prefix_hash()


# ----------------------------------------------------------------------------- prefix_cache.rs tests
def test_prefix_cache_matches_full_blocks(be):                                 # prefix_cache.rs:401-423
    c = be.PrefixCache(4, True, 8)
    assert c.insert_prefix([1, 2, 3, 4, 5, 6, 7, 8], [0, 1]) == []
    m, blocks = c.match_prefix(list(range(1, 13)))
    assert m == 2 and blocks == [0, 1]


def test_prefix_cache_evicts_leaf_blocks(be):                                  # :425-453
    c = be.PrefixCache(4, True, 1)
    toks = [1, 2, 3, 4, 5, 6, 7, 8]
    assert c.insert_prefix(toks, [5, 6]) == []        # just-inserted blocks are protected
    assert c.cached_blocks() == 2
    assert c.evict_blocks(1) == [6]
    assert c.match_prefix(toks)[0] == 1


def test_prefix_cache_insert_trims_older_leaves_before_new_prefix(be):         # :455-482
    c = be.PrefixCache(4, True, 2)
    old, new = [1, 2, 3, 4, 5, 6, 7, 8], [9, 10, 11, 12, 13, 14, 15, 16]
    assert c.insert_prefix(old, [1, 2]) == []
    assert c.insert_prefix(new, [3, 4]) == [2, 1]
    assert c.match_prefix(old)[0] == 0
    assert c.match_prefix(new)[0] == 2


def test_lru_compacts_after_repeated_touches(be):                              # :484-507
    c = be.PrefixCache(4, True, 64)
    c.insert_prefix([1, 2, 3, 4], [0])
    for _ in range(500):
        c.match_prefix([1, 2, 3, 4])
    assert c.lru_len() < 500


def test_insert_does_not_evict_just_inserted_blocks(be):                       # :509-536
    c = be.PrefixCache(4, True, 3)
    c.insert_prefix([10, 20, 30, 40], [10])
    new = list(range(1, 13))
    ev = c.insert_prefix(new, [0, 1, 2])
    assert not ({0, 1, 2} & set(ev))
    assert c.match_prefix(new)[0] == 3


def test_evict_blocks_respects_protected_set(be):                              # :538-565
    c = be.PrefixCache(4, True, 100)
    a, b = [1, 2, 3, 4], [5, 6, 7, 8]
    c.insert_prefix(a, [0])
    c.insert_prefix(b, [1])
    assert c.cached_blocks() == 2
    assert c.evict_blocks(2, protect_tokens=a) == [1]
    assert c.match_prefix(a)[0] == 1


def test_seed_block_affects_only_target_block_hash(be):                        # :567-598
    c = be.PrefixCache(4, True, 100)
    toks = list(range(1, 13))
    h0 = c.hash_for_blocks(toks, 3)
    hs = [c.hash_for_blocks(toks, 3, 42, i) for i in range(3)]
    assert h0 != hs[0] and hs[0] != hs[1] and hs[1] != hs[2]
    assert c.hash_for_blocks(toks, 3, 42, 1) == hs[1]
    assert c.hash_for_blocks(toks, 3, 99, 1) != hs[1]


# ----------------------------------------------------------------------------- sequence.rs tests
def test_prefill_chunk_tokens_warmup_boundaries(be):                           # sequence.rs:488-535
    eng = be.BlockEngine(64, 4, 4, False, 0)
    s = eng.new_sequence(0, list(range(20000)))
    s.set_mamba_prefix_warmup_tokens(5824)
    assert s.prefill_chunk_tokens(8192) == 5824
    s.set_num_cached_tokens(5824)
    assert s.prefill_chunk_tokens(8192) == 8192
    s2 = eng.new_sequence(1, list(range(50000)))
    s2.set_mamba_prefix_warmup_tokens(20000)
    assert s2.prefill_chunk_tokens(8192) == 8192
    s2.set_num_cached_tokens(8192)
    assert s2.prefill_chunk_tokens(8192) == 8192
    s2.set_num_cached_tokens(16384)
    assert s2.prefill_chunk_tokens(8192) == 3616
    s2.set_num_cached_tokens(20000)
    assert s2.prefill_chunk_tokens(8192) == 8192
    s3 = eng.new_sequence(2, list(range(10000)))
    s3.set_num_cached_tokens(4096)
    s3.set_mamba_prefix_warmup_tokens(4096)
    assert s3.prefill_chunk_tokens(8192) == 5904
    s3.set_mamba_prefix_warmup_tokens(12000)
    assert s3.prefill_chunk_tokens(8192) == 5904
    s4 = eng.new_sequence(3, list(range(10000)))
    s4.set_num_cached_tokens(1000)
    s4.set_mamba_prefix_warmup_tokens(5000)
    assert s4.prefill_chunk_tokens(0) == 9000


# ----------------------------------------------------------------------------- beyond the reference's tests
def test_fifo_allocator_and_alloc_status(be):
    eng = be.BlockEngine(4, 4, 2, False, 0)
    a = eng.new_sequence(1, [1, 2, 3, 4, 5])            # 2 logical blocks
    assert eng.can_allocate([a]) == be.OK
    eng.allocate([a])
    assert eng.block_table(a) == [0, 1]                 # FIFO ids
    big = eng.new_sequence(2, list(range(40)))          # 11 blocks > 4 total
    assert eng.can_allocate([big]) == be.IMPOSSIBLE
    mid = eng.new_sequence(3, list(range(9)))           # 3 blocks, 2 free
    assert eng.can_allocate([mid]) == be.LATER
    eng.free_sequence(a)
    assert eng.free_block_ids() == [2, 3, 0, 1]         # freed blocks go to the back
    assert eng.can_allocate([mid]) == be.OK
    eng.allocate([mid])
    assert eng.block_table(mid) == [2, 3, 0]


def test_copy_on_write_when_last_block_is_shared(be):
    eng = be.BlockEngine(4, 8, 0, False, 0)
    s1, s2 = eng.new_sequence(1, [1, 2, 3, 4, 5]), eng.new_sequence(2, [1, 2, 3, 4, 5])
    eng.allocate([s1, s2])                              # beam-style group: ONE table shared by both sequences,
    # sized by the group's total logical blocks (block_engine.rs:396-413 -- a reference quirk we keep)
    assert eng.block_table(s1) == eng.block_table(s2) == [0, 1, 2, 3]
    assert eng.refcount(3) == 2
    s1.add_token(6)
    cow = eng.append_token_slot_to_seq(s1)
    assert cow == (3, 4)                                # shared last block copied into a fresh one (:1199-1210)
    assert eng.block_table(s1) == [0, 1, 2, 4] and eng.block_table(s2) == [0, 1, 2, 3]
    assert eng.refcount(3) == 1 and eng.refcount(4) == 1 and eng.refcount(0) == 2
    s2.add_token(7)
    assert eng.append_token_slot_to_seq(s2) is None     # now exclusive


def test_swap_out_in_roundtrip_and_rollback(be):
    eng = be.BlockEngine(4, 6, 6, False, 0)
    s = eng.new_sequence(1, list(range(9)))             # 3 blocks
    eng.allocate([s])
    assert eng.block_table(s) == [0, 1, 2]
    assert eng.can_swap_out_seq_group([s])
    m = eng.swap_out(7, [s])
    assert m == {0: 0, 1: 1, 2: 2}
    assert eng.block_table(s) == [-1, -2, -3]           # CPU codes
    assert eng.get_num_free_blocks() == 6 and eng.get_num_free_cpu_blocks() == 3
    eng.rollback_swap_out(7)                            # the copy failed: tables and refcounts restored
    assert eng.block_table(s) == [0, 1, 2]
    assert eng.get_num_free_blocks() == 3 and eng.get_num_free_cpu_blocks() == 6
    m = eng.swap_out(8, [s])
    eng.finalize_swap_out(8)
    assert eng.can_swap_in_seq_group([s])
    back = eng.swap_in(9, [s])
    assert sorted(back.keys()) == sorted(m.values())
    assert all(b >= 0 for b in eng.block_table(s))
    eng.finalize_swap_in(9)
    assert eng.get_num_free_cpu_blocks() == 6 and eng.get_num_free_blocks() == 3


def test_input_preparation_matches_the_numpy_restatement(be):
    """a1 / a2 from the engine state vs oracle.ops.prepare_decode / prepare_prompt (bit-exact integer arrays)."""
    rng = np.random.default_rng(3)
    bs = 16
    eng = be.BlockEngine(bs, 64, 0, False, 0)
    seqs = [eng.new_sequence(i, rng.integers(0, 1000, n).tolist()) for i, n in enumerate([5, 16, 33, 47])]
    for s in seqs:
        eng.allocate([s])

    def snapshot():
        return [{"tokens": all_tokens[s.id][:s.get_len()], "block_table": eng.block_table(s)} for s in seqs]

    # the engine owns the token lists; mirror them here from the same seed
    rng = np.random.default_rng(3)
    all_tokens = {i: rng.integers(0, 1000, n).tolist() for i, n in enumerate([5, 16, 33, 47])}
    # --- prompt step, whole prompts
    got = eng.prepare_prompt(seqs)
    ref = O.prepare_prompt(snapshot(), bs)
    for k in ("input_ids", "positions", "slot_mapping", "context_lens", "block_tables", "cu_seqlens_q", "cu_seqlens_k"):
        assert np.array_equal(np.asarray(got[k], np.int64), np.asarray(ref[k], np.int64)), k
    # --- a few decode steps with appends
    for step in range(20):
        for s in seqs:
            t = int(rng.integers(0, 1000))
            s.add_token(t)
            all_tokens[s.id].append(t)
            assert eng.append_token_slot_to_seq(s) is None
        got = eng.prepare_decode(seqs)
        ref = O.prepare_decode(snapshot(), bs)
        for k in ("input_ids", "positions", "slot_mapping", "context_lens", "block_tables"):
            assert np.array_equal(np.asarray(got[k], np.int64), np.asarray(ref[k], np.int64)), (step, k)
    # --- chunked prompt step with a cached prefix
    eng2 = be.BlockEngine(bs, 64, 0, False, 0)
    p = rng.integers(0, 1000, 70).tolist()
    s = eng2.new_sequence(0, p)
    eng2.allocate([s])
    s.set_num_cached_tokens(32)
    got = eng2.prepare_prompt([s], chunk=24)
    ref = O.prepare_prompt([{"tokens": p[:56], "block_table": eng2.block_table(s)}], bs, num_cached_tokens=[32])
    for k in ("input_ids", "positions", "slot_mapping", "context_lens", "cu_seqlens_q", "cu_seqlens_k"):
        assert np.array_equal(np.asarray(got[k], np.int64), np.asarray(ref[k], np.int64)), k


def test_input_preparation_at_baseline_sizes_ragged_and_empty(be):
    """a1 / a2 at BASELINE's full sizes: 32 sequences, ragged contexts U[256,4096] plus one of 5000 tokens (79 blocks of
    64 -- the table width of SURVEY 8a), decode steps that cross a block boundary, a 1-sequence batch, and the empty
    batch; bit-exact against oracle.ops (inputs.rs:376-454)."""
    rng = np.random.default_rng(77)
    bs = 64
    lens = rng.integers(256, 4097, 32).tolist()
    lens[7], lens[19], lens[30] = 5000, 4095, 64                   # widest table; one token short of a boundary; exactly one block
    eng = be.BlockEngine(bs, sum(-(-(n + 8) // bs) for n in lens) + 4, 0, False, 0)
    toks = {i: rng.integers(0, 128256, n).tolist() for i, n in enumerate(lens)}
    seqs = [eng.new_sequence(i, toks[i]) for i in range(32)]
    order = rng.permutation(32).tolist()                             # allocation order != batch order: tables interleave
    for i in order:
        eng.allocate([seqs[i]])

    def snap(group):
        return [{"tokens": toks[s.id][:s.get_len()], "block_table": eng.block_table(s)} for s in group]
    got = eng.prepare_prompt(seqs)
    ref = O.prepare_prompt(snap(seqs), bs)
    for k in ("input_ids", "positions", "slot_mapping", "context_lens", "block_tables", "cu_seqlens_q", "cu_seqlens_k"):
        assert np.array_equal(np.asarray(got[k], np.int64), np.asarray(ref[k], np.int64)), k
    assert int(got["cu_seqlens_q"][-1]) == sum(lens) and np.asarray(got["block_tables"]).shape == (32, 79)
    for step in range(3):                                            # 4095 -> 4096 -> 4097 crosses a block boundary
        for s in seqs:
            t = int(rng.integers(0, 128256))
            s.add_token(t)
            toks[s.id].append(t)
            assert eng.append_token_slot_to_seq(s) is None
        got = eng.prepare_decode(seqs)
        ref = O.prepare_decode(snap(seqs), bs)
        for k in ("input_ids", "positions", "slot_mapping", "context_lens", "block_tables"):
            assert np.array_equal(np.asarray(got[k], np.int64), np.asarray(ref[k], np.int64)), (step, k)
        assert len(set(np.asarray(got["slot_mapping"]).tolist())) == 32
    one = [seqs[7]]
    got, ref = eng.prepare_decode(one), O.prepare_decode(snap(one), bs)
    for k in ("input_ids", "positions", "slot_mapping", "context_lens", "block_tables"):
        assert np.array_equal(np.asarray(got[k], np.int64), np.asarray(ref[k], np.int64)), k
    assert np.asarray(got["block_tables"]).shape == (1, 79)
    # the empty batch: the reference unwraps max() of an empty list here (inputs.rs:441-444) and the oracle mirrors that;
    # the C entry point answers "0 sequences, 0 table columns" instead of failing
    empty = eng.prepare_decode([])
    assert len(empty["input_ids"]) == 0 and len(empty["slot_mapping"]) == 0 and empty["block_tables"].size == 0
    with pytest.raises(ValueError):
        O.prepare_decode([], bs)
    for s in seqs:
        eng.free_sequence(s)
    assert eng.get_num_free_blocks() == eng.get_num_blocks()


def test_cache_budget_matches_the_reference_tests_and_the_oracle(lib):
    """a24: `compute_kvcache_budget_bytes` against the reference's own unit tests (src/lib.rs:773-785: 700 * 0.9 -> 630,
    0 -> 0) and `get_cache_config` (src/lib.rs:128-284) against the oracle restatement and SURVEY 8a's worked number
    (Llama-3-8B bf16, block 64: 8 MiB per block across 32 layers x (K+V))"""
    import ctypes
    assert lib.mi355_kvcache_budget_bytes(700, 0.9) == 630               # test_compute_kvcache_budget_bytes
    assert lib.mi355_kvcache_budget_bytes(0, 0.9) == 0                   # ..._clamps_to_zero
    assert lib.mi355_kvcache_budget_bytes(1000, 1.0) == 1000
    for bad in (0.0, -0.5, 1.5, float("nan")):                           # "kv_fraction must be in (0, 1]"
        assert lib.mi355_kvcache_budget_bytes(700, bad) == -1
    # f32 fraction widened to f64, then round-half-away: 0.9f32 = 0.89999997..., 5 * 0.9f32 = 4.4999998 -> 4
    assert lib.mi355_kvcache_budget_bytes(5, 0.9) == 4 and lib.mi355_kvcache_budget_bytes(15, 0.5) == 8

    def cfg(mem_gpu, mem_cpu, bs, hkv, d, layers, dsize, shards, swap):
        g, c = ctypes.c_int64(-1), ctypes.c_int64(-1)
        rc = lib.mi355_get_cache_config(mem_gpu, mem_cpu, bs, hkv, d, layers, dsize, shards, swap, ctypes.addressof(g), ctypes.addressof(c))
        return rc, g.value, c.value
    assert cfg(16384, 0, 64, 8, 128, 32, 2, 1, 1) == (0, 2048, 1024)    # 16 GiB / 8 MiB; CPU default = half
    assert cfg(16384, 4096, 64, 8, 128, 32, 2, 1, 1) == (0, 2048, 512)  # explicit CPU budget
    assert cfg(16384, 4096, 64, 8, 128, 32, 2, 1, 0) == (0, 2048, 0)    # swap off: no CPU blocks
    assert cfg(16384, 0, 64, 8, 128, 32, 1, 1, 1)[1] == 4096            # fp8 cache: 1-byte elements
    assert cfg(16384, 0, 64, 8, 128, 32, 2, 8, 1)[1] == 16384           # TP 8: one kv head per rank
    assert cfg(16384, 0, 64, 8, 128, 32, 2, 16, 1)[1] == 16384          # fewer heads than shards: still one head
    assert cfg(100, 0, 64, 8, 128, 0, 2, 1, 1)[1] == cfg(100, 0, 64, 8, 128, 1, 2, 1, 1)[1]   # layers.max(1)
    rng = np.random.default_rng(9)
    for _ in range(200):
        mem, bs = int(rng.integers(1, 200000)), int(rng.choice([16, 32, 64]))
        hkv, d, layers = int(rng.choice([1, 2, 4, 8, 32])), int(rng.choice([64, 80, 128, 256])), int(rng.integers(1, 81))
        dsize, shards = int(rng.choice([1, 2, 4])), int(rng.choice([1, 2, 4, 8]))
        local = 1 if hkv < shards else hkv // shards
        assert cfg(mem, 0, bs, hkv, d, layers, dsize, shards, 1)[1] == O.num_gpu_blocks(mem, dsize, bs, local, d, layers)
    assert cfg(1, 0, 0, 8, 128, 32, 2, 1, 1)[0] == 1 and cfg(-1, 0, 64, 8, 128, 32, 2, 1, 1)[0] == 1
