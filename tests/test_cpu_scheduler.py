"""Scheduling policy of src/scheduler/mod.rs restated over the C ABI: scenario tests of the observable policy
(the reference has no unit tests for `schedule` itself; each assert cites the rule it checks), and an end-to-end
simulation of BASELINE config 3's shape: 32 staggered sequences through prompt + decode steps to completion with
block-table invariants checked every step."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def be(lib):
    from candle_vllm_amd import block_engine
    return block_engine


def _mk(be, **kw):
    args = dict(block_size=4, num_gpu_blocks=16, num_cpu_blocks=16, max_num_parallel_reqs=8,
                max_num_batched_tokens=64, prefill_chunk_size=0)
    args.update(kw)
    return be.Scheduler(**args)


def test_prompt_and_decode_steps_alternate(be):
    s = _mk(be)
    a = s.block_engine.new_sequence(1, list(range(6)))
    s.add_sequence(10, [a])
    out = s.schedule()
    assert out.is_prompt and out.scheduled == [10]                 # waiting -> running on a prompt step (mod.rs:196-257)
    b = s.block_engine.new_sequence(2, list(range(5)))
    s.add_sequence(11, [b])
    out = s.schedule()
    assert not out.is_prompt and out.scheduled == [10]             # is_last_prefill && running: decode first (:197-199)
    out = s.schedule()
    assert out.is_prompt and out.scheduled == [11]                 # then the new prompt
    out = s.schedule()
    assert not out.is_prompt and sorted(out.scheduled) == [10, 11]
    assert out.scheduled == [11, 10]                               # FCFS sort puts the earliest arrival last (:777-782)


def test_prefill_token_budget_and_parallel_limit(be):
    s = _mk(be, max_num_batched_tokens=10, max_num_parallel_reqs=3, num_gpu_blocks=64)
    for i in range(5):
        s.add_sequence(i, [s.block_engine.new_sequence(i, list(range(4)))])
    out = s.schedule()
    assert out.scheduled == [0, 1]                                 # 4 + 4 <= 10 < 12 (:202-212)
    s.schedule()                                                   # decode
    out = s.schedule()
    assert out.is_prompt and out.scheduled == [2]                  # running.len() reaches max_num_parallel_reqs = 3 (:214-216)
    s.schedule()
    out = s.schedule()
    assert not out.is_prompt and s.num_waiting() == 2


def test_prompt_longer_than_the_pool_is_ignored(be):
    s = _mk(be, num_gpu_blocks=4, max_num_batched_tokens=1000)
    s.add_sequence(1, [s.block_engine.new_sequence(1, list(range(100)))])
    s.add_sequence(2, [s.block_engine.new_sequence(2, list(range(5)))])
    out = s.schedule()
    assert out.ignored_seq_groups == [1] and out.scheduled == [2]  # AllocStatus::Impossible (:233-241)
    assert s.status(1) == be.Scheduler.IGNORED


def test_chunked_prefill_requeues_until_the_prompt_is_done(be):
    s = _mk(be, prefill_chunk_size=8, num_gpu_blocks=32)
    q = s.block_engine.new_sequence(1, list(range(20)))
    s.add_sequence(1, [q])
    steps = []
    for _ in range(3):
        out = s.schedule()
        assert out.is_prompt and out.scheduled == [1]
        meta = s.block_engine.prepare_prompt([q], chunk=8)
        steps.append((int(meta["cu_seqlens_q"][1]), int(meta["context_lens"][0]), len(s.block_engine.block_table(q))))
        fin = s.filter_prefill_finished(out.scheduled)
        if fin:
            break
        assert s.num_waiting() == 1 and s.num_running() == 0       # pushed back to waiting as Pending (:566-580)
        assert s.status(1) == be.Scheduler.PENDING
    assert steps == [(8, 8, 2), (8, 16, 4), (4, 20, 5)]            # chunk tokens, context, blocks reserved per chunk
    assert fin == [1] and s.num_running() == 1


def test_out_of_blocks_preempts_by_recompute(be):
    """The running queue is sorted by arrival DESCENDING (sort ascending then reverse, mod.rs:777-782), visited from
    the front and preempted from the BACK (:296-311): with the pool exhausted it is the EARLIEST arrival that is
    recomputed -- the reference's own comment says "preempting the lowest (earliest) first".  Mirrored as is."""
    s = _mk(be, num_gpu_blocks=4, block_size=4)
    a = s.block_engine.new_sequence(1, list(range(7)))             # 2 blocks
    b = s.block_engine.new_sequence(2, list(range(7)))             # 2 blocks
    s.add_sequence(1, [a])
    s.add_sequence(2, [b])
    assert s.schedule().scheduled == [1, 2]
    for q in (a, b):
        q.add_token(99)                                            # 8 tokens -> 3 logical blocks each, pool is empty
    out = s.schedule()
    assert not out.is_prompt and out.scheduled == [2]
    assert s.status(1) == be.Scheduler.WAITING and s.num_waiting() == 1
    assert s.take_pending_runner_releases() == [1]                 # recompute: runner state released (:718-723)
    assert s.block_engine.get_num_free_blocks() == 1               # 2 freed, 1 taken by the survivor's new slot
    assert len(s.block_engine.block_table(b)) == 3


def test_preemption_swaps_when_the_prefix_cache_is_on(be):
    s = _mk(be, num_gpu_blocks=4, num_cpu_blocks=8, prefix_cache_enabled=True, max_cached_blocks=2)
    a = s.block_engine.new_sequence(1, list(range(7)))
    b = s.block_engine.new_sequence(2, list(range(100, 107)))
    s.add_sequence(1, [a])
    s.add_sequence(2, [b])
    s.schedule()
    for q in (a, b):
        q.add_token(5)                                             # 8 tokens -> a third block each, none free
    out = s.schedule(now_ms=1000)
    assert out.scheduled == [2] and out.swap_out_groups == [1]     # _preempt_by_swap of the back of the queue (:725-755)
    assert len(out.blocks_to_swap_out) == 2
    assert s.status(1) == be.Scheduler.SWAPPED and s.num_swapped() == 1
    assert all(c < 0 for c in s.block_engine.block_table(a))       # table now points at CPU blocks
    s.set_finished(2)
    assert s.free_finished_sequence_groups() == [2]
    out = s.schedule(now_ms=1100)
    assert out.swap_in_groups == []                                # 300 ms cooling period (:364-373)
    out = s.schedule(now_ms=1400)
    assert out.swap_in_groups == [1] and out.scheduled == [1]
    assert all(c >= 0 for c in s.block_engine.block_table(a))


def _table(eng, seq):
    try:
        return eng.block_table(seq)
    except KeyError:                                                       # no table: the sequence holds no blocks
        return []


def _two_groups_about_to_collide(be, n_seqs_of_victim=1):
    """two running groups on a 4-block pool (prefix cache on => preemption by swap), each about to need a third block"""
    s = _mk(be, num_gpu_blocks=4 if n_seqs_of_victim == 1 else 6, num_cpu_blocks=8, prefix_cache_enabled=True, max_cached_blocks=2)
    eng = s.block_engine
    victim = [eng.new_sequence(10 + i, list(range(50 * i, 50 * i + 7))) for i in range(n_seqs_of_victim)]
    other = eng.new_sequence(2, list(range(100, 107)))
    s.add_sequence(1, victim)
    s.add_sequence(2, [other])
    s.schedule()
    for q in victim + [other]:
        q.add_token(5)
    return s, eng, victim, other


def test_refused_swap_out_falls_back_to_recompute_for_a_single_sequence(be):
    """`mi355_be_swap_out` refusing AFTER `can_swap_out` said yes (block_engine.cpp preempt(): k < 0): nothing was moved, so the
    group must not be recorded as swapped out -- a single sequence is recomputed (mod.rs:700-723), exactly as if the swap had
    been impossible."""
    s, eng, (a,), b = _two_groups_about_to_collide(be)
    eng.test_refuse_swaps(n_out=1)
    out = s.schedule(now_ms=1000)
    assert out.scheduled == [2]
    assert out.swap_out_groups == [] and out.blocks_to_swap_out == {}      # no copy was requested
    assert s.status(1) == be.Scheduler.WAITING and s.num_swapped() == 0 and s.num_waiting() == 1
    assert s.take_pending_runner_releases() == [10]                        # recompute releases the runner state of the sequence
    assert _table(eng, a) == []                                            # its blocks went back to the pool
    assert eng.get_num_free_cpu_blocks() == 8                              # and the CPU pool was never touched
    # the hook is spent: the next collision swaps for real
    s.set_finished(2)
    s.free_finished_sequence_groups()
    out = s.schedule(now_ms=1400)
    assert out.is_prompt and out.scheduled == [1]                          # the recomputed group is prefilled again


def test_refused_swap_out_aborts_a_multi_sequence_group(be):
    """a group of several sequences cannot be recomputed (mod.rs:706-716): a refused swap aborts it and frees its blocks"""
    s, eng, victim, b = _two_groups_about_to_collide(be, n_seqs_of_victim=2)
    free_before = eng.get_num_free_blocks()
    eng.test_refuse_swaps(n_out=1)
    out = s.schedule(now_ms=1000)
    assert out.swap_out_groups == [] and s.num_swapped() == 0
    assert s.status(1) != be.Scheduler.SWAPPED
    if s.status(1) != be.Scheduler.RUNNING:                                # the victim was preempted: it must be gone, not half-swapped
        assert s.status(1) == be.Scheduler.ABORTED
        assert all(_table(eng, q) == [] for q in victim)
        assert eng.get_num_free_blocks() >= free_before
        assert eng.get_num_free_cpu_blocks() == 8


def test_refused_swap_in_keeps_the_group_swapped_and_first_in_line(be):
    """`mi355_be_swap_in` refusing (schedule(): k < 0): the group stays SWAPPED at the head of the queue with its CPU blocks, no
    copy is requested, and it comes back on the next step"""
    s, eng, (a,), b = _two_groups_about_to_collide(be)
    out = s.schedule(now_ms=1000)
    assert out.swap_out_groups == [1]
    eng.finalize_swap_out(1)
    cpu_table = eng.block_table(a)
    s.set_finished(2)
    s.free_finished_sequence_groups()
    eng.test_refuse_swaps(n_in=1)
    out = s.schedule(now_ms=2000)                                          # past the cooling period: would swap in, but is refused
    assert out.swap_in_groups == [] and out.blocks_to_swap_in == {} and out.scheduled == []
    assert s.status(1) == be.Scheduler.SWAPPED and s.num_swapped() == 1 and s.num_running() == 0
    assert eng.block_table(a) == cpu_table                                 # untouched
    out = s.schedule(now_ms=2400)
    assert out.swap_in_groups == [1] and out.scheduled == [1]
    assert all(c >= 0 for c in eng.block_table(a))


def test_abort_frees_blocks(be):
    s = _mk(be)
    a = s.block_engine.new_sequence(1, list(range(9)))
    s.add_sequence(1, [a])
    s.schedule()
    free = s.block_engine.get_num_free_blocks()
    assert s.abort_sequences([1]) == 1
    assert s.status(1) == be.Scheduler.ABORTED
    assert s.block_engine.get_num_free_blocks() == free + 3 and not s.has_unfinished_sequences()


def test_continuous_batching_simulation_32_sequences(be):
    """BASELINE config 3's shape, scaled down: 32 sequences, prompt lengths U[16,256], staggered arrival, decode to a
    random length; every step the scheduled block tables must be disjoint across sequences, slots must be unique,
    and at the end every block is back in the pool."""
    rng = np.random.default_rng(32)
    bs, nblk = 16, 160
    s = _mk(be, block_size=bs, num_gpu_blocks=nblk, num_cpu_blocks=64, max_num_parallel_reqs=32,
            max_num_batched_tokens=512, prefill_chunk_size=128)
    eng = s.block_engine
    seqs, target, done = {}, {}, set()
    arrivals = sorted(rng.integers(0, 40, 32).tolist())
    next_id, step, prompt_steps, decode_steps, preempted = 0, 0, 0, 0, 0
    while len(done) < 32 and step < 5000:
        while next_id < 32 and arrivals[next_id] <= step:
            n = int(rng.integers(16, 257))
            seqs[next_id] = eng.new_sequence(next_id, rng.integers(0, 1000, n).tolist())
            target[next_id] = n + int(rng.integers(4, 64))
            s.add_sequence(next_id, [seqs[next_id]])
            next_id += 1
        out = s.schedule(now_ms=step * 50)
        preempted += len(s.take_pending_runner_releases())
        group = [seqs[g] for g in out.scheduled]
        if out.is_prompt:
            prompt_steps += 1
            meta = eng.prepare_prompt(group, chunk=128)
            assert int(meta["cu_seqlens_q"][-1]) <= 512                       # per-step prefill token budget
            finished = s.filter_prefill_finished(out.scheduled)
            decoding = [seqs[g] for g in finished]
        else:
            decode_steps += 1
            decoding = group
            if group:
                meta = eng.prepare_decode(group)
                assert len(set(meta["slot_mapping"].tolist())) == len(group)   # one distinct slot per sequence
        tables = [eng.block_table(q) for q in group]
        flat = [b for t in tables for b in t]
        assert len(flat) == len(set(flat))                                     # no block shared between live sequences
        for q in decoding:                                                     # "sample" one token each
            if q.id in done:
                continue
            q.add_token(int(rng.integers(0, 1000)))
            if q.get_len() >= target[q.id]:
                s.set_finished(q.id)
                done.add(q.id)
        s.free_finished_sequence_groups()
        step += 1
    assert len(done) == 32, (len(done), step)
    assert eng.get_num_free_blocks() == nblk and not s.has_unfinished_sequences()
    assert prompt_steps > 0 and decode_steps > 0


@pytest.mark.parametrize("prefix_cache", [False, True])
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_scheduler_stress_under_block_pressure(be, seed, prefix_cache):
    """random arrivals, shared prompt prefixes, a pool far too small for the offered load (forces preemption by
    recompute without the prefix cache, by swap with it), random aborts: whatever the interleaving,
      * a decode step only schedules groups whose tables are entirely on the GPU, with one distinct slot each,
      * a physical block belongs to two live sequences only if its reference count says so,
      * every group reaches FINISHED / ABORTED / IGNORED (no starvation, no lost group),
      * at the end every GPU block is free or held by the prefix cache, and every CPU block is free."""
    rng = np.random.default_rng(1000 * seed + int(prefix_cache))
    bs, nblk, ncpu, N = 8, 24, 48, 40
    s = _mk(be, block_size=bs, num_gpu_blocks=nblk, num_cpu_blocks=ncpu, max_num_parallel_reqs=6,
            max_num_batched_tokens=96, prefill_chunk_size=32, prefix_cache_enabled=prefix_cache,
            max_cached_blocks=6 if prefix_cache else 0)
    eng = s.block_engine
    common = rng.integers(0, 50, 40).tolist()                         # shared system prompt: prefix-cache hits
    seqs, target, terminal = {}, {}, set()
    arrivals = sorted(rng.integers(0, 120, N).tolist())
    next_id, step, now = 0, 0, 0
    saw_swap = saw_recompute = 0
    TERMINAL = (be.Scheduler.FINISHED, be.Scheduler.ABORTED, be.Scheduler.IGNORED)
    while len(terminal) < N and step < 20000:
        while next_id < N and arrivals[next_id] <= step:
            n = int(rng.integers(4, 90))
            toks = (common[: int(rng.integers(0, 41))] + rng.integers(0, 50, n).tolist())[:n]
            if rng.random() < 0.05:
                toks = rng.integers(0, 50, bs * nblk + 5).tolist()     # can never fit: must be IGNORED, not starve the queue
            seqs[next_id] = eng.new_sequence(next_id, toks)
            target[next_id] = len(toks) + int(rng.integers(1, 40))
            s.add_sequence(next_id, [seqs[next_id]])
            next_id += 1
        now += 400                                                    # past the 300 ms swap-in cooling every step
        out = s.schedule(now_ms=now)
        for g in out.ignored_seq_groups:
            terminal.add(g)
        saw_recompute += len(s.take_pending_runner_releases())
        saw_swap += len(out.swap_out_groups)
        for g in out.swap_out_groups:                                 # execute_scheduler_ops: copies done, commit them
            eng.finalize_swap_out(g)
        for g in out.swap_in_groups:
            eng.finalize_swap_in(g)
        group = [seqs[g] for g in out.scheduled]
        if out.is_prompt:
            if group:
                meta = eng.prepare_prompt(group, chunk=32)
                assert int(meta["cu_seqlens_q"][-1]) <= 96
            decoding = [seqs[g] for g in s.filter_prefill_finished(out.scheduled)]
        else:
            decoding = group
            if group:
                meta = eng.prepare_decode(group)
                assert len(set(meta["slot_mapping"].tolist())) == len(group)
                assert (np.asarray(meta["slot_mapping"]) >= 0).all()
        owners = {}
        for q in group:
            t = eng.block_table(q)
            assert all(b >= 0 for b in t), "a scheduled group still has swapped-out blocks"
            for b in set(t):
                owners[b] = owners.get(b, 0) + 1
        for b, n in owners.items():
            assert eng.refcount(b) >= n, (b, n, eng.refcount(b))
        for q in decoding:
            if q.id in terminal:
                continue
            q.add_token(int(rng.integers(0, 50)))
            if q.get_len() >= target[q.id]:
                s.set_finished(q.id)
                terminal.add(q.id)
        live = [g for g in range(next_id) if g not in terminal and s.status(g) not in TERMINAL]
        if live and rng.random() < 0.02:                               # a client disconnects
            victim = int(rng.choice(live))
            assert s.abort_sequences([victim]) == 1
            terminal.add(victim)
        s.free_finished_sequence_groups()
        step += 1
    assert len(terminal) == N, (len(terminal), step, s.num_waiting(), s.num_running(), s.num_swapped())
    for g in range(N):
        assert s.status(g) in TERMINAL or s.status(g) < 0, (g, s.status(g))
    assert not s.has_unfinished_sequences()
    cached = eng.prefix_cache_blocks() if prefix_cache else 0
    assert eng.get_num_free_blocks() + cached == nblk, (eng.get_num_free_blocks(), cached)
    assert eng.get_num_free_cpu_blocks() == ncpu
    assert saw_recompute + saw_swap > 0                                # the pool really was under pressure
    if prefix_cache:
        assert saw_swap > 0


def test_active_sequence_limit_is_at_least_one(be):
    """the reference's own unit test (scheduler/mod.rs:59-62: `active_sequence_limit(0, None) == 1`): a parallel-request
    limit of 0 still admits one sequence per prompt step instead of starving the queue"""
    s = _mk(be, max_num_parallel_reqs=0)
    a = s.block_engine.new_sequence(1, list(range(5)))
    b = s.block_engine.new_sequence(2, list(range(5)))
    s.add_sequence(1, [a])
    s.add_sequence(2, [b])
    out = s.schedule()
    assert out.is_prompt and out.scheduled == [1] and s.num_waiting() == 1
