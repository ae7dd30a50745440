"""The reference's engine loop over the C ABI: `LLMEngine::generate_once` + `execute_scheduler_ops` + `execute_scheduled_batch`
(src/openai/pipelines/llm_engine.rs:1357-1400,1785-1791) -- scheduler -> block manager -> step inputs -> model -> greedy token ->
append / finish / free, with requests arriving while others decode (continuous batching).  The scheduler, the block manager and the
input preparation are the C++ of csrc/block_engine.cpp; a decode step is ONE hipGraph replay of the library's step driver
(mi355_llama_decode_begin copies the step's inputs into the static buffers -- what backend/graph.rs:592-661 does before it replays the
graph of the batch size -- then mi355_llama_decode_step, mi355_llama_decode_read_tokens); prompt steps run eagerly, sampled on the
device (mi355_argmax_f32).  Harness code (tests/test_gpu_engine.py, bench_legs.py `engine_b32`), torch only for device buffers.

Per-request times follow the reference's usage record: `prompt_time` = arrival -> the prompt step that produced the first token,
`completion_time` = from there to the last token; decode tok/s = mean over requests of completion_tokens / completion_time, throughput =
that x the number of requests (llm_engine.rs:984-1002)."""
import time

import numpy as np


class Request:
    def __init__(self, rid, prompt, max_new, arrival_step=0):
        self.id, self.prompt, self.max_new, self.arrival_step = rid, list(prompt), int(max_new), int(arrival_step)
        self.tokens = []                     # generated
        self.logits = []                     # traced runs: the f32 logits row behind every generated token
        self.t_arrive = self.t_first = self.t_done = None
        self.seq = None


def run_engine(gm, sched, requests, chunk=0, bt_width=None, ctx_cap=None, stream=0, graph=True, max_steps=100000, swap=None, trace_ids=()):
    """drive `requests` to completion; returns a dict of counters.  gm: model.GGUFLLaMa; sched: block_engine.Scheduler.
    bt_width / ctx_cap: the fixed block-table width and context bucket of the captured decode graphs (default: the model's limits).
    swap(mapping, to_host): optional executor of the scheduler's swap maps (cache_engine.rs:345-385).
    trace_ids: requests whose logits row is kept for every token they sample (Request.logits; parity runs, never timed runs)."""
    import torch
    from . import ops as cvo
    eng = sched.block_engine
    bt_width = bt_width or int(gm.c.max_blocks_per_seq)
    ctx_cap = ctx_cap or min(gm.cfg.max_seq, bt_width * gm.cfg.block_size)
    pending = sorted(requests, key=lambda r: (r.arrival_step, r.id))
    by_id = {r.id: r for r in requests}
    done, step, pi = set(), 0, 0
    stats = dict(prompt_steps=0, decode_steps=0, prompt_tokens=0, max_batch=0, preempted=0, batch_hist={}, idle_steps=0)
    gm.set_graph(bool(graph))
    t0 = time.perf_counter()
    while len(done) < len(requests) and step < max_steps:
        while pi < len(pending) and pending[pi].arrival_step <= step:
            r = pending[pi]
            r.seq = eng.new_sequence(r.id, r.prompt)
            sched.add_sequence(r.id, [r.seq])
            r.t_arrive = time.perf_counter()
            pi += 1
        out = sched.schedule(now_ms=int((time.perf_counter() - t0) * 1e3))
        stats["preempted"] += len(sched.take_pending_runner_releases())
        if swap is None and (len(out.blocks_to_swap_in) or len(out.blocks_to_swap_out) or len(out.swap_in_groups) or len(out.swap_out_groups)):
            # the scheduler decided to move KV blocks and nobody executes the copies: a group swapped back in would decode on blocks
            # that were never filled (the reference's execute_scheduler_ops always performs them, cache_engine.rs:527-535)
            raise RuntimeError("run_engine: the scheduler emitted swap operations but no `swap` callback was supplied "
                               "(pass swap=..., or give the scheduler no CPU blocks so that it preempts by recompute)")
        if swap is not None:                                       # execute_scheduler_ops order: in, out, (copy)
            swap(out.blocks_to_swap_in, False)
            swap(out.blocks_to_swap_out, True)
            for g in out.swap_in_groups:
                eng.finalize_swap_in(g)
            for g in out.swap_out_groups:
                eng.finalize_swap_out(g)
        group = [by_id[g].seq for g in out.scheduled]
        if not group:
            stats["idle_steps"] += 1
            step += 1
            continue
        if out.is_prompt:
            meta = eng.prepare_prompt(group, chunk=chunk)
            logits = gm.forward_prefill(meta)
            toks = cvo.argmax(logits).cpu().numpy()
            sampled = set(sched.filter_prefill_finished(out.scheduled)) if chunk else set(out.scheduled)
            rows = {row: logits[row].cpu().numpy() for row, gid in enumerate(out.scheduled) if gid in trace_ids}
            stats["prompt_steps"] += 1
            stats["prompt_tokens"] += len(meta["input_ids"])
        else:
            meta = eng.prepare_decode(group)
            B = len(group)
            bt = np.zeros((B, bt_width), np.uint32)
            w = meta["block_tables"].shape[1]
            bt[:, :w] = meta["block_tables"]
            gm.decode_begin(meta["input_ids"], meta["context_lens"], bt, ctx_cap=ctx_cap, stream=stream)
            gm.decode_step(stream)
            toks = gm.read_tokens(stream)
            sampled = set(out.scheduled)
            rows = {}
            if any(gid in trace_ids for gid in out.scheduled):
                lg = gm.logits_numpy(B)
                rows = {row: lg[row].copy() for row, gid in enumerate(out.scheduled) if gid in trace_ids}
            stats["decode_steps"] += 1
            stats["max_batch"] = max(stats["max_batch"], B)
            stats["batch_hist"][B] = stats["batch_hist"].get(B, 0) + 1
        now = time.perf_counter()
        for row, gid in enumerate(out.scheduled):
            r = by_id[gid]
            if gid not in sampled or gid in done:
                continue
            tok = int(toks[row])
            if not r.tokens:
                r.t_first = now
            r.tokens.append(tok)
            if row in rows:
                r.logits.append(rows[row])
            r.seq.add_token(tok)
            if len(r.tokens) >= r.max_new:
                r.t_done = now
                sched.set_finished(gid)
                done.add(gid)
        sched.free_finished_sequence_groups()
        step += 1
    stats["steps"] = step
    stats["wall_s"] = time.perf_counter() - t0
    stats["finished"] = len(done)
    return stats


def usage_summary(requests):
    """the reference's closing printout (llm_engine.rs:984-1002) over finished requests"""
    fin = [r for r in requests if r.t_done is not None]
    n = len(fin)
    if not n:
        return {"requests": 0}
    dec = [(len(r.tokens) - 1) / max(r.t_done - r.t_first, 1e-6) for r in fin]      # tokens after the first / the time they took
    pre = [len(r.prompt) / max(r.t_first - r.t_arrive, 1e-6) for r in fin]
    return {"requests": n, "completion_tokens": int(sum(len(r.tokens) for r in fin)), "prompt_tokens": int(sum(len(r.prompt) for r in fin)),
            "decode_tps_avg": float(np.mean(dec)), "decode_throughput": float(np.mean(dec) * n),
            "prompt_tps_avg": float(np.mean(pre)), "prompt_throughput": float(np.mean(pre) * n)}
