"""The whole hot path as the reference's engine loop drives it (llm_engine.rs execute_scheduled_batch): the C++ scheduler
picks a prompt or a decode step, the block manager builds the step inputs, the GPU model runs the step, greedy tokens are
appended, finished sequences are freed -- continuous batching with staggered arrivals and chunked prefill.  Every
sequence must end up with exactly the tokens the ORACLE generates for it ALONE: batching, chunking and block placement
must not change a result."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import llama                               # noqa: E402
from oracle import ops as O                            # noqa: E402

pytestmark = pytest.mark.gpu


def _oracle_alone(orc, cfg, prompt, n_new):
    nblk = -(-(len(prompt) + n_new + 1) // cfg.block_size)
    cache = orc.new_cache(nblk + 1)
    seq = {"tokens": list(prompt), "block_table": list(range(1, nblk + 1))}
    lg = orc.forward(O.prepare_prompt([seq], cfg.block_size), cache, is_prefill=True)
    out = []
    for _ in range(n_new):
        t = int(lg[0].argmax())
        out.append(t)
        seq["tokens"].append(t)
        lg = orc.forward(O.prepare_decode([seq], cfg.block_size), cache)
    return out


@pytest.mark.parametrize("chunk,layout,nblk", [(0, "paged", 48), (24, "paged", 48), (24, "flash", 48), (0, "paged", 14)])
def test_continuous_batching_engine_loop_matches_per_sequence_oracle(lib, chunk, layout, nblk):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    from candle_vllm_amd import model as M
    from candle_vllm_amd import block_engine as be
    cfg = llama.LlamaConfig.tiny()
    W = llama.make_weights(cfg, seed=2024)
    flash = layout == "flash"
    orc = llama.OracleLlama(cfg, W, flash_layout=flash)
    rng = np.random.default_rng(606)
    NSEQ = 7                                            # nblk = 14: the pool runs dry -> preemption by recompute (mod.rs:303-330)
    prompts = [[int(t) for t in rng.integers(0, cfg.vocab, int(n))] for n in rng.integers(5, 60, NSEQ)]
    n_new = [int(n) for n in rng.integers(3, 12, NSEQ)]
    arrivals = sorted(int(a) for a in rng.integers(0, 12, NSEQ))
    want = [_oracle_alone(orc, cfg, p, n) for p, n in zip(prompts, n_new)]

    sched = be.Scheduler(block_size=cfg.block_size, num_gpu_blocks=nblk, num_cpu_blocks=8, max_num_parallel_reqs=8,
                         max_num_batched_tokens=96, prefill_chunk_size=chunk)
    eng = sched.block_engine
    gm = M.GGUFLLaMa(cfg, max_batch=8, kv_layout=M.KV_FLASH if flash else M.KV_PAGED)
    gm.load_oracle_weights(W)
    gm.alloc_kv_cache(nblk)
    seqs, got, done = {}, {i: [] for i in range(NSEQ)}, set()
    next_id, step, prompt_steps, decode_steps, max_batch_seen, preempted = 0, 0, 0, 0, 0, 0
    while len(done) < NSEQ and step < 400:
        while next_id < NSEQ and arrivals[next_id] <= step:
            seqs[next_id] = eng.new_sequence(next_id, prompts[next_id])
            sched.add_sequence(next_id, [seqs[next_id]])
            next_id += 1
        out = sched.schedule(now_ms=step * 50)
        preempted += len(sched.take_pending_runner_releases())
        group = [seqs[g] for g in out.scheduled]
        if group:
            if out.is_prompt:
                prompt_steps += 1
                meta = eng.prepare_prompt(group, chunk=chunk)
                logits = gm.forward_prefill(meta).cpu().numpy()
                sampled = sched.filter_prefill_finished(out.scheduled) if chunk else list(out.scheduled)
            else:
                decode_steps += 1
                max_batch_seen = max(max_batch_seen, len(group))
                meta = eng.prepare_decode(group)
                logits = gm.forward_decode(meta).cpu().numpy()
                sampled = list(out.scheduled)
            for row, gid in enumerate(out.scheduled):
                if gid not in sampled or gid in done:
                    continue
                tok = int(logits[row].argmax())
                got[gid].append(tok)
                seqs[gid].add_token(tok)
                if len(got[gid]) >= n_new[gid]:
                    sched.set_finished(gid)
                    done.add(gid)
            sched.free_finished_sequence_groups()
        step += 1
    assert len(done) == NSEQ, (len(done), step)
    assert prompt_steps >= 2 and decode_steps >= 2 and max_batch_seen >= 3      # the batch really was mixed and ragged
    assert (preempted > 0) == (nblk < 20), preempted                            # the small pool did preempt, the large one did not
    assert eng.get_num_free_blocks() == nblk and not sched.has_unfinished_sequences()
    for i in range(NSEQ):
        assert got[i] == want[i], (i, got[i], want[i])


def test_engine_loop_with_swap_preemption(lib):
    """Prefix cache on -> preemption swaps a sequence's KV blocks to the host (mod.rs:725-755) and back after the
    cooling period; the engine executes the scheduler's block ops the way `execute_scheduler_ops` does
    (llm_engine.rs:1357-1400: swap_in, swap_out, copy, then finalize) with `mi355_swap_blocks` on every layer's K
    and V.  A swapped-and-restored sequence must still generate exactly the oracle's tokens."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    import ctypes
    from candle_vllm_amd import model as M
    from candle_vllm_amd import block_engine as be
    from candle_vllm_amd import ops as cvo
    cfg = llama.LlamaConfig.tiny()
    W = llama.make_weights(cfg, seed=2025)
    orc = llama.OracleLlama(cfg, W, flash_layout=False)
    rng = np.random.default_rng(707)
    NSEQ, nblk, ncpu = 5, 10, 24
    prompts = [[int(t) for t in rng.integers(0, cfg.vocab, int(n))] for n in rng.integers(20, 40, NSEQ)]
    n_new = [int(n) for n in rng.integers(14, 22, NSEQ)]
    want = [_oracle_alone(orc, cfg, p, n) for p, n in zip(prompts, n_new)]
    sched = be.Scheduler(block_size=cfg.block_size, num_gpu_blocks=nblk, num_cpu_blocks=ncpu, max_num_parallel_reqs=8,
                         max_num_batched_tokens=256, prefill_chunk_size=0, prefix_cache_enabled=True, max_cached_blocks=2)
    eng = sched.block_engine
    gm = M.GGUFLLaMa(cfg, max_batch=8, kv_layout=M.KV_PAGED)
    gm.load_oracle_weights(W)
    gm.alloc_kv_cache(nblk)
    block_bytes = cfg.n_kv_heads * cfg.head_dim * cfg.block_size * 2
    cpu_cache = torch.zeros((cfg.n_layers, 2, ncpu, block_bytes), dtype=torch.uint8).pin_memory()
    st = torch.cuda.current_stream().cuda_stream

    def swap(mapping, to_host):
        if not mapping:
            return
        flat = []
        for s, d in mapping.items():
            flat += [int(s), int(d)]
        arr = (ctypes.c_int64 * len(flat))(*flat)
        for l in range(cfg.n_layers):
            for which in (0, 1):                                   # K then V (cache_engine.rs:352-358)
                dev_ptr = M.lib.mi355_llama_kv_ptr(gm.h, l, which)
                host_ptr = cpu_cache[l, which].data_ptr()
                src, dst = (dev_ptr, host_ptr) if to_host else (host_ptr, dev_ptr)
                rc = M.lib.mi355_swap_blocks(src, dst, ctypes.cast(arr, ctypes.c_void_p), len(flat) // 2, block_bytes,
                                             cvo.SWAP_D2H if to_host else cvo.SWAP_H2D, st)
                assert rc == 0
        torch.cuda.synchronize()

    seqs = {i: eng.new_sequence(i, prompts[i]) for i in range(NSEQ)}
    for i in range(NSEQ):
        sched.add_sequence(i, [seqs[i]])
    got, done = {i: [] for i in range(NSEQ)}, set()
    swapped_out, swapped_in, step = 0, 0, 0
    while len(done) < NSEQ and step < 600:
        out = sched.schedule(now_ms=step * 200)                    # 200 ms per step: the 300 ms cooling passes in 2 steps
        sched.take_pending_runner_releases()
        swap(out.blocks_to_swap_in, to_host=False)                 # execute_scheduler_ops order: in, out, copy
        swap(out.blocks_to_swap_out, to_host=True)
        assert not out.blocks_to_copy
        for g in out.swap_in_groups:
            eng.finalize_swap_in(g)
        for g in out.swap_out_groups:
            eng.finalize_swap_out(g)
        swapped_out += len(out.swap_out_groups)
        swapped_in += len(out.swap_in_groups)
        group = [seqs[g] for g in out.scheduled]
        if group:
            if out.is_prompt:
                logits = gm.forward_prefill(eng.prepare_prompt(group)).cpu().numpy()
            else:
                logits = gm.forward_decode(eng.prepare_decode(group)).cpu().numpy()
            for row, gid in enumerate(out.scheduled):
                if gid in done:
                    continue
                tok = int(logits[row].argmax())
                got[gid].append(tok)
                seqs[gid].add_token(tok)
                if len(got[gid]) >= n_new[gid]:
                    sched.set_finished(gid)
                    done.add(gid)
            sched.free_finished_sequence_groups()
        step += 1
    assert len(done) == NSEQ, (len(done), step)
    assert swapped_out > 0 and swapped_in == swapped_out, (swapped_out, swapped_in)
    for i in range(NSEQ):
        assert got[i] == want[i], (i, got[i], want[i])


def test_engine_loop_prefix_cache_hit_reuses_kv_blocks(lib):
    """Prefix cache: a finished sequence leaves its full blocks in the cache (hash chain over token blocks); a later
    prompt with the same first 32 tokens is allocated ON those blocks, only its remaining tokens are computed (prefill
    with a cached prefix, K4 through the block table) -- and it must still produce the oracle's tokens."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    from candle_vllm_amd import model as M
    from candle_vllm_amd import block_engine as be
    cfg = llama.LlamaConfig.tiny()
    W = llama.make_weights(cfg, seed=2026)
    orc = llama.OracleLlama(cfg, W, flash_layout=False)
    rng = np.random.default_rng(808)
    shared = [int(t) for t in rng.integers(0, cfg.vocab, 2 * cfg.block_size)]
    prompts = [shared + [int(t) for t in rng.integers(0, cfg.vocab, 9)],
               shared + [int(t) for t in rng.integers(0, cfg.vocab, 13)]]
    n_new = [6, 7]
    want = [_oracle_alone(orc, cfg, p, n) for p, n in zip(prompts, n_new)]
    sched = be.Scheduler(block_size=cfg.block_size, num_gpu_blocks=24, num_cpu_blocks=8, max_num_parallel_reqs=4,
                         max_num_batched_tokens=256, prefill_chunk_size=0, prefix_cache_enabled=True, max_cached_blocks=8)
    eng = sched.block_engine
    gm = M.GGUFLLaMa(cfg, max_batch=4, kv_layout=M.KV_PAGED)
    gm.load_oracle_weights(W)
    gm.alloc_kv_cache(24)
    got = {}
    computed_prompt_tokens = []
    for gid in (0, 1):                                              # one after the other: the second arrives when the first is done
        seq = eng.new_sequence(gid, prompts[gid])
        sched.add_sequence(gid, [seq])
        got[gid] = []
        for step in range(64):
            out = sched.schedule(now_ms=1000 * gid + 50 * step)
            if not out.scheduled:
                continue
            if out.is_prompt:
                meta = eng.prepare_prompt([seq])
                computed_prompt_tokens.append(len(meta["input_ids"]))
                logits = gm.forward_prefill(meta).cpu().numpy()
            else:
                logits = gm.forward_decode(eng.prepare_decode([seq])).cpu().numpy()
            tok = int(logits[0].argmax())
            got[gid].append(tok)
            seq.add_token(tok)
            if len(got[gid]) >= n_new[gid]:
                sched.set_finished(gid)
                sched.free_finished_sequence_groups()
                break
    assert got[0] == want[0] and got[1] == want[1], (got, want)
    assert computed_prompt_tokens[0] == len(prompts[0])
    assert computed_prompt_tokens[1] == len(prompts[1]) - 2 * cfg.block_size     # the shared two blocks were NOT recomputed
    assert eng.prefix_cache_blocks() > 0


@pytest.mark.parametrize("chunk", [0, 24])
def test_engine_module_graph_replayed_decode_steps_match_per_sequence_oracle(lib, chunk):
    """candle_vllm_amd.engine.run_engine -- the loop bench_legs.py `engine_b32` times at Llama-3-8B size: decode steps go through the
    library's step driver (decode_begin copies the step's inputs into the static buffers, ONE hipGraph replay per step, one graph per batch
    size kept as the batch grows and shrinks), prompt steps are sampled on the device.  Staggered arrivals, chunked prefill; every request
    must end with exactly the tokens the oracle generates for it alone, and the usage record must follow llm_engine.rs:984-1002."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    from candle_vllm_amd import model as M
    from candle_vllm_amd import block_engine as be
    from candle_vllm_amd import engine as E
    cfg = llama.LlamaConfig.tiny()
    W = llama.make_weights(cfg, seed=2027)
    orc = llama.OracleLlama(cfg, W, flash_layout=False)
    rng = np.random.default_rng(909)
    NSEQ, nblk = 9, 64
    prompts = [[int(t) for t in rng.integers(0, cfg.vocab, int(n))] for n in rng.integers(5, 60, NSEQ)]
    n_new = [int(n) for n in rng.integers(4, 14, NSEQ)]
    arrivals = sorted(int(a) for a in rng.integers(0, 10, NSEQ))
    want = [_oracle_alone(orc, cfg, p, n) for p, n in zip(prompts, n_new)]
    sched = be.Scheduler(block_size=cfg.block_size, num_gpu_blocks=nblk, num_cpu_blocks=8, max_num_parallel_reqs=8,
                         max_num_batched_tokens=96, prefill_chunk_size=chunk)
    gm = M.GGUFLLaMa(cfg, max_batch=8, max_blocks_per_seq=8, kv_layout=M.KV_PAGED)
    gm.load_oracle_weights(W)
    gm.alloc_kv_cache(nblk)
    reqs = [E.Request(i, prompts[i], n_new[i], arrivals[i]) for i in range(NSEQ)]
    stream = torch.cuda.Stream()
    stats = E.run_engine(gm, sched, reqs, chunk=chunk, stream=stream.cuda_stream, graph=True)
    assert stats["finished"] == NSEQ and stats["max_batch"] >= 3 and stats["prompt_steps"] >= 2
    assert len(stats["batch_hist"]) >= 3                                       # several batch sizes = several cached graphs
    for i, r in enumerate(reqs):
        assert r.tokens == want[i], (i, r.tokens, want[i])
    u = E.usage_summary(reqs)
    assert u["requests"] == NSEQ and u["completion_tokens"] == sum(n_new) and u["decode_throughput"] > 0
    assert sched.block_engine.get_num_free_blocks() == nblk and not sched.has_unfinished_sequences()
