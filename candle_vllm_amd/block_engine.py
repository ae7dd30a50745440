"""Host mirror of the reference's scheduler-side block bookkeeping over the C ABI (include/mi355_vllm.h section 5):
`BlockEngine` (src/scheduler/block_engine.rs), `PrefixCache` (prefix_cache.rs), `Sequence` (sequence.rs) -- same
method names and meaning, so tests read like the reference's own (block_engine.rs:1476-1752 etc.)."""
import ctypes

import numpy as np

from ._lib import lib

OK, LATER, IMPOSSIBLE = 0, 1, 2          # AllocStatus


def _ids(seq_ids):
    a = np.ascontiguousarray(seq_ids, np.int64)
    return a, a.ctypes.data, len(a)


def _tok(tokens):
    a = np.ascontiguousarray(tokens, np.uint32)
    return a, a.ctypes.data, len(a)


class Sequence:
    """`_Sequence::new(&tokens, seq_id, block_size, None)`; state lives in the engine."""

    def __init__(self, engine, seq_id, tokens):
        self.engine, self.id = engine, int(seq_id)
        a, p, n = _tok(tokens)
        if lib.mi355_be_seq_create(engine.h, self.id, p, n) != 0:
            raise ValueError("duplicate sequence id")

    def add_token(self, token):
        lib.mi355_be_seq_add_token(self.engine.h, self.id, int(token))

    def get_len(self):
        return lib.mi355_be_seq_len(self.engine.h, self.id)

    def get_logical_token_blocks(self):
        return lib.mi355_be_seq_logical_blocks(self.engine.h, self.id)

    def get_num_cached_tokens(self):
        return lib.mi355_be_seq_get_cached_tokens(self.engine.h, self.id)

    def set_num_cached_tokens(self, n):
        lib.mi355_be_seq_set_cached_tokens(self.engine.h, self.id, int(n))

    def set_mamba_prefix_warmup_tokens(self, n):
        lib.mi355_be_seq_set_warmup_tokens(self.engine.h, self.id, -1 if n is None else int(n))

    def prefill_chunk_tokens(self, chunk):
        return lib.mi355_be_seq_prefill_chunk_tokens(self.engine.h, self.id, int(chunk))

    def has_prefix_hash(self):
        return lib.mi355_be_seq_has_prefix_hash(self.engine.h, self.id) == 1


class BlockEngine:
    """`BlockEngine::new(block_size, num_gpu_blocks, num_cpu_blocks, _, PrefixCacheConfig, false)`."""

    def __init__(self, block_size, num_gpu_blocks, num_cpu_blocks, prefix_cache_enabled=False, max_cached_blocks=0):
        self.block_size = block_size
        self.h = lib.mi355_be_create(block_size, num_gpu_blocks, num_cpu_blocks, int(prefix_cache_enabled),
                                     max_cached_blocks)
        if not self.h:
            raise ValueError("bad block engine arguments")

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib.mi355_be_destroy(self.h)
            self.h = None

    def new_sequence(self, seq_id, tokens):
        return Sequence(self, seq_id, tokens)

    # -- counters
    def get_num_free_blocks(self):
        return lib.mi355_be_num_free_blocks(self.h)

    def get_num_free_cpu_blocks(self):
        return lib.mi355_be_num_free_cpu_blocks(self.h)

    def get_num_blocks(self):
        return lib.mi355_be_num_blocks(self.h)

    def prefix_cache_blocks(self):
        return lib.mi355_be_prefix_cache_blocks(self.h)

    def free_block_ids(self):
        out = np.zeros(self.get_num_blocks() + 1, np.int32)
        n = lib.mi355_be_free_block_ids(self.h, out.ctypes.data, len(out))
        return out[:n].tolist()

    def block_table(self, seq):
        out = np.zeros(4096, np.int32)
        n = lib.mi355_be_block_table(self.h, seq.id, out.ctypes.data, len(out))
        if n < 0:
            raise KeyError(seq.id)
        return out[:n].tolist()

    def refcount(self, block):
        return lib.mi355_be_block_refcount(self.h, int(block))

    # -- allocation (groups are lists of Sequence)
    def can_allocate(self, group, chunk=0):
        a, p, n = _ids([s.id for s in group])
        return lib.mi355_be_can_allocate(self.h, p, n, chunk)

    can_allocate_for_prefill = can_allocate

    def allocate(self, group, chunk=0):
        a, p, n = _ids([s.id for s in group])
        rc = lib.mi355_be_allocate(self.h, p, n, chunk)
        if rc != 0:
            raise RuntimeError(f"allocate failed ({rc}): no free GPU blocks")

    allocate_for_prefill = allocate

    def can_append_token_to_seq(self, group):
        a, p, n = _ids([s.id for s in group])
        return lib.mi355_be_can_append_token(self.h, p, n) == 1

    def append_token_slot_to_seq(self, seq):
        src, dst = ctypes.c_int32(-1), ctypes.c_int32(-1)
        rc = lib.mi355_be_append_token_slot(self.h, seq.id, ctypes.addressof(src), ctypes.addressof(dst))
        if rc < 0:
            raise RuntimeError(f"append_token_slot failed ({rc})")
        return (src.value, dst.value) if rc == 1 else None

    def prefill_chunk_blocks_required(self, group, chunk):
        a, p, n = _ids([s.id for s in group])
        return lib.mi355_be_prefill_chunk_blocks_required(self.h, p, n, chunk)

    def can_append_prefill_chunk_to_seq_group(self, group, chunk):
        a, p, n = _ids([s.id for s in group])
        return lib.mi355_be_can_append_prefill_chunk(self.h, p, n, chunk) == 1

    def append_prefill_chunk_slots_to_seq_group(self, group, chunk):
        a, p, n = _ids([s.id for s in group])
        if lib.mi355_be_append_prefill_chunk_slots(self.h, p, n, chunk) != 0:
            raise RuntimeError("append_prefill_chunk_slots failed")

    def free_sequence(self, seq):
        lib.mi355_be_free_sequence(self.h, seq.id)

    def cache_sequence(self, seq):
        return lib.mi355_be_cache_sequence(self.h, seq.id)

    def evict_prefix_cache_blocks(self, n):
        return lib.mi355_be_evict_prefix_cache_blocks(self.h, n)

    def evict_prefix_cache_until_free(self, n):
        return lib.mi355_be_evict_prefix_cache_until_free(self.h, n)

    def query_prefix_cache_match_tokens(self, tokens):
        a, p, n = _tok(tokens)
        return lib.mi355_be_query_prefix_match_tokens(self.h, p, n)

    def fallback_sequence_to_full_prefill(self, seq):
        return lib.mi355_be_fallback_to_full_prefill(self.h, seq.id) == 1

    def rebuild_sequence_with_cached_prefix(self, seq, cached_tokens):
        return lib.mi355_be_rebuild_with_cached_prefix(self.h, seq.id, cached_tokens) == 1

    def _pop_back_block(self, seq):
        lib.mi355_be_pop_back_block(self.h, seq.id)

    # -- swap
    def can_swap_out_seq_group(self, group):
        a, p, n = _ids([s.id for s in group])
        return lib.mi355_be_can_swap_out(self.h, p, n) == 1

    def can_swap_in_seq_group(self, group):
        a, p, n = _ids([s.id for s in group])
        return lib.mi355_be_can_swap_in(self.h, p, n) == 1

    def _swap(self, fn, group_id, group):
        a, p, n = _ids([s.id for s in group])
        pairs = np.zeros(2 * 4096, np.int64)
        k = fn(self.h, group_id, p, n, pairs.ctypes.data, 4096)
        if k < 0:
            raise RuntimeError(f"swap failed ({k})")
        return {int(pairs[2 * i]): int(pairs[2 * i + 1]) for i in range(k)}

    def swap_out(self, group_id, group):
        """-> {gpu_block: cpu_block}"""
        return self._swap(lib.mi355_be_swap_out, group_id, group)

    def swap_in(self, group_id, group):
        """-> {cpu_block: gpu_block}"""
        return self._swap(lib.mi355_be_swap_in, group_id, group)

    def test_refuse_swaps(self, n_out=0, n_in=0):
        """test hook: the next n_out swap-outs / n_in swap-ins are refused before anything is touched"""
        lib.mi355_be_test_refuse_swaps(self.h, n_out, n_in)

    def finalize_swap_out(self, gid):
        lib.mi355_be_finalize_swap_out(self.h, gid)

    def rollback_swap_out(self, gid):
        lib.mi355_be_rollback_swap_out(self.h, gid)

    def finalize_swap_in(self, gid):
        lib.mi355_be_finalize_swap_in(self.h, gid)

    def rollback_swap_in(self, gid):
        lib.mi355_be_rollback_swap_in(self.h, gid)

    # -- input preparation (pipelines/inputs.rs)
    def prepare_decode(self, group):
        a, p, n = _ids([s.id for s in group])
        cap = 4096
        tok, pos = np.zeros(n, np.uint32), np.zeros(n, np.int64)
        slot, ctx = np.zeros(n, np.int64), np.zeros(n, np.uint32)
        bt = np.zeros(n * cap, np.uint32)
        mb = lib.mi355_be_prepare_decode(self.h, p, n, tok.ctypes.data, pos.ctypes.data, slot.ctypes.data,
                                         ctx.ctypes.data, bt.ctypes.data, cap)
        if mb < 0:
            raise RuntimeError(f"prepare_decode failed ({mb})")
        return {"input_ids": tok, "positions": pos, "slot_mapping": slot, "context_lens": ctx,
                "block_tables": bt[: n * mb].reshape(n, mb).copy(), "max_context_len": int(ctx.max()) if n else 0}

    def prepare_prompt(self, group, chunk=0, tok_cap=1 << 20):
        a, p, n = _ids([s.id for s in group])
        cap = 4096
        tok, pos, slot = np.zeros(tok_cap, np.uint32), np.zeros(tok_cap, np.int64), np.zeros(tok_cap, np.int64)
        ctx, cu_q, cu_k = np.zeros(n, np.uint32), np.zeros(n + 1, np.uint32), np.zeros(n + 1, np.uint32)
        bt = np.zeros(n * cap, np.uint32)
        mb = ctypes.c_int32(0)
        T = lib.mi355_be_prepare_prompt(self.h, p, n, chunk, tok.ctypes.data, pos.ctypes.data, slot.ctypes.data,
                                        ctx.ctypes.data, cu_q.ctypes.data, cu_k.ctypes.data, bt.ctypes.data, tok_cap,
                                        cap, ctypes.addressof(mb))
        if T < 0:
            raise RuntimeError(f"prepare_prompt failed ({T})")
        return {"input_ids": tok[:T].copy(), "positions": pos[:T].copy(), "slot_mapping": slot[:T].copy(),
                "context_lens": ctx, "block_tables": bt[: n * mb.value].reshape(n, mb.value).copy(),
                "cu_seqlens_q": cu_q, "cu_seqlens_k": cu_k, "max_seqlen_q": int(np.diff(cu_q.astype(np.int64)).max()),
                "max_seqlen_k": int(np.diff(cu_k.astype(np.int64)).max()), "max_context_len": int(ctx.max())}


class PrefixCache:
    """`PrefixCache::new(block_size, PrefixCacheConfig{enabled, max_cached_blocks})` driven with bare block ids, as
    the reference's unit tests do (prefix_cache.rs:392-399)."""

    def __init__(self, block_size, enabled, max_cached_blocks, num_block_ids=256):
        self.h = lib.mi355_pc_create(block_size, int(enabled), max_cached_blocks, num_block_ids)

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib.mi355_be_destroy(self.h)
            self.h = None

    def insert_prefix(self, tokens, blocks):
        a, p, n = _tok(tokens)
        b = np.ascontiguousarray(blocks, np.int32)
        ev = np.zeros(1024, np.int32)
        k = lib.mi355_pc_insert(self.h, p, n, b.ctypes.data, len(b), ev.ctypes.data, len(ev))
        return ev[:k].tolist()

    def match_prefix(self, tokens):
        """-> (matched_blocks, [block ids of the match])"""
        a, p, n = _tok(tokens)
        b = np.zeros(1024, np.int32)
        m = lib.mi355_pc_match(self.h, p, n, b.ctypes.data, len(b))
        return m, b[:m].tolist()

    def evict_blocks(self, num, protect_tokens=None):
        ev = np.zeros(1024, np.int32)
        if protect_tokens is None:
            k = lib.mi355_pc_evict(self.h, num, None, 0, ev.ctypes.data, len(ev))
        else:
            a, p, n = _tok(protect_tokens)
            k = lib.mi355_pc_evict(self.h, num, p, n, ev.ctypes.data, len(ev))
        return ev[:k].tolist()

    def cached_blocks(self):
        return lib.mi355_pc_cached_blocks(self.h)

    def lru_len(self):
        return lib.mi355_pc_lru_len(self.h)

    def hash_for_blocks(self, tokens, full_blocks, seed=None, seed_block=None):
        a, p, n = _tok(tokens)
        h = lib.mi355_pc_hash_for_blocks(self.h, p, n, full_blocks, int(seed is not None), int(seed or 0),
                                         -1 if seed_block is None else int(seed_block))
        return h or None


class _BorrowedEngine(BlockEngine):
    """the engine owned by a Scheduler (not destroyed by this wrapper)"""

    def __init__(self, handle, block_size):
        self.h, self.block_size = handle, block_size

    def __del__(self):
        self.h = None


class SchedulerOutput:
    def __init__(self, is_prompt, scheduled, ignored, swap_in, swap_out, copy, swap_in_groups, swap_out_groups):
        self.is_prompt = is_prompt                      # the step kind (`is_last_prefill` after the call)
        self.scheduled = scheduled                      # group ids
        self.ignored_seq_groups = ignored
        self.blocks_to_swap_in = swap_in                # {cpu_block: gpu_block}
        self.blocks_to_swap_out = swap_out              # {gpu_block: cpu_block}
        self.blocks_to_copy = copy                      # {src: [dst, ...]}
        self.swap_in_groups, self.swap_out_groups = swap_in_groups, swap_out_groups


class Scheduler:
    """`Scheduler` of src/scheduler/mod.rs (new :98-121, add_sequence :123, schedule :183-455, filter_prefill_finished
    :542-616, free_finished_sequence_groups :465-504, abort_sequences :618-657).  Groups are (group_id, [Sequence])."""
    WAITING, PENDING, RUNNING, SWAPPED, FINISHED, ABORTED, IGNORED = range(7)

    def __init__(self, block_size, num_gpu_blocks, num_cpu_blocks, max_num_parallel_reqs, max_num_batched_tokens,
                 prefill_chunk_size=0, prefix_cache_enabled=False, max_cached_blocks=0):
        self.h = lib.mi355_sched_create(block_size, num_gpu_blocks, num_cpu_blocks, int(prefix_cache_enabled),
                                        max_cached_blocks, max_num_parallel_reqs, max_num_batched_tokens,
                                        prefill_chunk_size)
        if not self.h:
            raise ValueError("bad scheduler arguments")
        self.block_engine = _BorrowedEngine(lib.mi355_sched_block_engine(self.h), block_size)
        self.groups = {}
        self._arrival = 0

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib.mi355_sched_destroy(self.h)
            self.h = None

    def add_sequence(self, group_id, seqs, arrival=None):
        a, p, n = _ids([s.id for s in seqs])
        if arrival is None:
            self._arrival += 1
            arrival = self._arrival
        if lib.mi355_sched_add_group(self.h, group_id, p, n, arrival) != 0:
            raise ValueError("bad group")
        self.groups[group_id] = list(seqs)

    def _result(self, which):
        out = np.zeros(1 << 15, np.int64)
        k = lib.mi355_sched_result(self.h, which, out.ctypes.data, len(out))
        return out[:k].tolist()

    def schedule(self, now_ms=0):
        is_prompt = lib.mi355_sched_schedule(self.h, int(now_ms)) == 1
        def pairs(w):
            v = self._result(w)
            return list(zip(v[0::2], v[1::2]))
        copy = {}
        for s, d in pairs(4):
            copy.setdefault(s, []).append(d)
        return SchedulerOutput(is_prompt, self._result(0), self._result(1), dict(pairs(2)), dict(pairs(3)), copy,
                               self._result(5), self._result(6))

    def take_pending_runner_releases(self):
        return self._result(7)

    def filter_prefill_finished(self, scheduled):
        a, p, n = _ids(scheduled)
        out = np.zeros(max(1, n), np.int64)
        k = lib.mi355_sched_filter_prefill_finished(self.h, p, n, out.ctypes.data, len(out))
        if k < 0:
            raise RuntimeError("filter_prefill_finished needs a prefill chunk size")
        return out[:k].tolist()

    def set_finished(self, group_id):
        lib.mi355_sched_set_group_finished(self.h, group_id)

    def free_finished_sequence_groups(self):
        out = np.zeros(4096, np.int64)
        k = lib.mi355_sched_free_finished(self.h, out.ctypes.data, len(out))
        return out[:k].tolist()

    def abort_sequences(self, seq_ids):
        a, p, n = _ids(seq_ids)
        return lib.mi355_sched_abort_sequences(self.h, p, n)

    def status(self, group_id):
        return lib.mi355_sched_group_status(self.h, group_id)

    def num_waiting(self):
        return lib.mi355_sched_queue_len(self.h, 0)

    def num_running(self):
        return lib.mi355_sched_queue_len(self.h, 1)

    def num_swapped(self):
        return lib.mi355_sched_queue_len(self.h, 2)

    def has_unfinished_sequences(self):
        return lib.mi355_sched_has_unfinished(self.h) == 1

    def rollback_swap_in_groups(self, gids):
        for g in gids:
            lib.mi355_sched_rollback_swap_in(self.h, g)

    def rollback_swap_out_groups(self, gids):
        for g in gids:
            lib.mi355_sched_rollback_swap_out(self.h, g)
