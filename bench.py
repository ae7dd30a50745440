#!/usr/bin/env python
"""bench.py -- Llama-3-8B Q4_K_M greedy decode on MI355X through the C-ABI decode path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
  metric/unit : BASELINE.json's metric -- decode tokens/s (greedy, batch 1, Llama-3-8B Q4_K GGUF shapes)
  a "step"    : one decode step of the whole model over the batch (1 token per sequence)
  workload    : BASELINE.json configs[1] -- "Llama-3-8B Q4_K GGUF, batch=1, 1xMI355X"; prompt context 4096 tokens
                already in the paged KV cache (README protocol "input 4k"), then K decode steps
  value       : tokens/s over the timed K steps, inputs/weights/KV resident in HBM, max over ranks
  N > 1       : tensor parallel over N GPUs (one process per GPU, RCCL all-reduce / all-gather), same batch ->
                "strong" scaling
Extra objects: `roofline` (dominant kernel, HIP-event timed inside this script), `cpu_baseline` (C port of the
reference CPU arithmetic on the host cores), `step` (whole-step algorithmic bytes and achieved GB/s).
Synthetic data: random-init weights of the named architecture in Q4_K_M mixture, random KV prefix.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling


# OpenMP threads that spin while idle burn a CPU quota and can slow the C oracle's cpu_baseline by an order of magnitude
# on hosts whose quota is below their visible CPU count; must be set before libgomp is loaded (torch, liboracle)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")


def llama3_8b():
    from candle_vllm_amd.model import ModelDims
    return ModelDims.llama3_8b()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--blocks", type=int, default=3, help="timed blocks of --steps steps each; the value is the median block")
    ap.add_argument("--settle-steps", type=int, default=160, help="N > 1 only: untimed steps before the first block (every rank must run the "
                                                                   "same number of steps, so the time criterion of N = 1 cannot be used)")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--ctx", type=int, default=4096, help="prompt tokens already in the KV cache")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--tp-eager", action="store_true", help="N > 1 only: run tensor-parallel steps eagerly instead of from a hipGraph")
    ap.add_argument("--all-reduce", choices=["auto", "rccl", "p2p"], default="auto",
                    help="N > 1 only: transport of the decode-sized all-reduces (<= 256 KiB).  auto (default): the in-stream one-shot "
                         "peer-to-peer kernel when every pair of ranks has peer access and its self-test passes on every rank, RCCL on "
                         "its side stream otherwise (the line's config.all_reduce says which); rccl / p2p force one")
    ap.add_argument("--p2p", action="store_true", help="same as --all-reduce p2p")
    ap.add_argument("--wire-bf16", action="store_true", help="N > 1 only: the reference's all-reduce numerics (bf16 partials on the wire)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=8)
    ap.add_argument("--no-batch32", action="store_true", help="skip the secondary batch-32 measurement")
    ap.add_argument("--no-experiments", action="store_true", help="skip the A/B of opt-in experiment kernels (separate process)")
    ap.add_argument("--b32-steps", type=int, default=24)
    ap.add_argument("--kv-layout", choices=["flash", "paged"], default="paged",
                    help="paged = vLLM layout (reference default without flash-attn features; MFMA attention), "
                         "flash = [NB,bs,Hkv,D]")
    ap.add_argument("--parity", choices=["off", "quick", "full"], default="full",
                    help="full-size check of the benchmarked geometry against the C oracle after the timed region "
                         "(quick: batch 1; full: + batch 32 ragged and a 2048-token prompt step)")
    ap.add_argument("--legs", default="all", help="secondary legs at BASELINE configs[2..4] shapes on one GPU (bench_legs.py): "
                                                   "all | none | comma list of bf16_b32,gptq_qwen2,mixtral_fp8")
    ap.add_argument("--layers", type=int, default=0, help="debug only: fewer layers (result marked invalid)")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal of the N > 1 command on a one-GPU box: every rank on cuda:0 (gloo control plane, the one-shot peer kernel "
                         "over IPC for the all-reduces, host-staged vocabulary gather); the line is marked invalid")
    return ap.parse_args()


def host_cpus():
    """what the host offers this process: sockets / physical cores / hardware threads from /proc/cpuinfo, the scheduler affinity, and the
    cgroup CPU quota (v2 cpu.max, v1 cfs_quota) -- a quota below the visible thread count is what keeps an OpenMP baseline from scaling"""
    d = {"hw_threads": os.cpu_count()}
    try:
        d["affinity"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    try:
        phys, cores = set(), set()
        pid = cid = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                pid = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                cid = ln.split(":")[1].strip()
            elif not ln.strip():
                if pid is not None:
                    phys.add(pid)
                    cores.add((pid, cid))
                pid = cid = None
        d["sockets"], d["physical_cores"] = len(phys) or None, len(cores) or None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                d["model"] = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = "unlimited" if q == "max" else round(int(q) / int(per), 2)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = "unlimited" if q < 0 else round(q / per, 2)
        except (OSError, ValueError):
            pass
    d["cgroup_cpu_quota"] = quota
    return d


def cpu_baseline(cfg, steps=8, ctx=4096):
    """C port of the reference's CPU arithmetic (Q8_K activations, integer dots in AVX-512 VNNI / AVX2, OpenMP over rows and
    attention heads) on the host cores: `steps` greedy decode steps of the same model at the SAME context as the GPU run
    (ctx tokens of random bf16 K/V in the paged pool), batch 1."""
    import ctypes
    from oracle import cref
    from candle_vllm_amd.model import q4km_type_for
    cref.build()
    names = ["wq", "wk", "wv", "wo", "w1", "w2", "w3"]
    types = [q4km_type_for(n, l, cfg.n_layers) for l in range(cfg.n_layers) for n in names]
    types.append(q4km_type_for("output", 0, cfg.n_layers))
    t0 = time.time()
    m = cref.CLlama(cfg, W=None, types=types, seed=1235)
    nblk = -(-(ctx + steps + 8) // cfg.block_size) + 1
    rng = np.random.default_rng(3)
    shape = (nblk, cfg.block_size, cfg.n_kv_heads, cfg.head_dim)
    kb = (rng.standard_normal(shape).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    vb = (rng.standard_normal(shape).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    cache = [(np.roll(kb, l, axis=0).copy(), np.roll(vb, 3 * l + 1, axis=0).copy()) for l in range(cfg.n_layers)]
    setup = time.time() - t0
    table = list(range(nblk))
    toks = [int(t) for t in rng.integers(0, cfg.vocab, ctx)]
    from oracle import ops as O
    # thread count: all hardware threads is not always the fastest (SMT siblings, NUMA, a cgroup quota below the visible CPU
    # count); one step per candidate, then `steps` steps at the best one
    L = cref.lib()
    L.orc_isa.restype = ctypes.c_char_p
    L.orc_set_attn_fast(1)                                    # the timed baseline may vectorise its QK dot; the parity oracle never does
    nmax = int(L.orc_num_threads())
    trial = {}

    def one_step():
        meta = O.prepare_decode([{"tokens": toks, "block_table": table}], cfg.block_size)
        t1 = time.time()
        lg = m.decode(meta, cache, o2=True)
        dt = time.time() - t1
        toks.append(int(lg[0].argmax()))
        return dt
    for n in sorted({nmax, max(1, nmax // 2), max(1, nmax // 4), max(1, nmax // 8), max(1, nmax // 16)}, reverse=True):
        L.orc_set_num_threads(n)
        trial[n] = one_step()
        if trial[n] > 20.0:                                   # keep the sample bounded on a slow host
            break
    nbest = min(trial, key=trial.get)
    L.orc_set_num_threads(nbest)
    ph = (ctypes.c_double * 3)()
    L.orc_llama_phase_times(ph, 1)
    times = [one_step() for _ in range(steps)]
    L.orc_llama_phase_times(ph, 1)
    mean = float(np.mean(times))
    L.orc_set_attn_fast(0)
    return {"value": round(1.0 / mean, 4), "unit": "tokens/s", "cores": nbest, "kind": "port",
            "host": host_cpus(),
            "phase_seconds_per_step": {"quantised_matvecs": round(ph[0] / steps, 4), "attention": round(ph[1] / steps, 4),
                                       "norms_rope_residuals": round(ph[2] / steps, 4),
                                       "outside_the_C_step (numpy argmax, ctypes)": round(mean - (ph[0] + ph[1] + ph[2]) / steps, 4)},
            "thread_scaling_s_per_step": {str(k): round(v, 4) for k, v in sorted(trial.items())},
            "ctx": ctx, "steps": steps, "isa": L.orc_isa().decode(), "host_threads": nmax, "best_step_tok_s": round(1.0 / min(times), 4),
            "sample": f"{steps} greedy decode steps, batch 1, ctx {ctx} (the GPU run's workload), full {cfg.n_layers}-layer Q4_K_M model; "
                      f"candle-CPU-style Q8_K integer dots (oracle/oracle.c, OpenMP, passive wait); mean step at the best of the "
                      f"thread counts tried {({k: round(v, 3) for k, v in trial.items()})} s/step; setup {setup:.1f}s"}


def cpu_baseline_config0(prompt=128, gen=64):
    """BASELINE configs[0] as SURVEY 8(d) words it: StableLM-3B bf16, batch 1, a 128-token prompt then 64 greedy tokens, on the
    host cores.  A port: the step's 16-bit mat-vecs (f32 accumulation, AVX2 + OpenMP: oracle/oracle.c orc_bf16_gemv) in the
    layer order of stable_lm.rs:158-212 with LayerNorm, partial rotary and attention in numpy over the tokens seen so far;
    synthetic bf16 weights (one random row block per shape, tiled; distinct memory per layer: the host streams the full 5.6 GB of
    weights every token).  The prompt goes through the same one-token step (the reference's CPU prompt path is a batched
    mat-mul: its prompt rate would be higher; the metric here is the DECODE rate over the 64 generated tokens)."""
    import ctypes
    from oracle import cref
    cref.build()
    d = dict(hidden=2560, n_layers=32, n_heads=32, head_dim=80, intermediate=6912, vocab=50304)      # StableLM-3B-4e1t shapes
    hid, H, D, I, V, NL = d["hidden"], d["n_heads"], d["head_dim"], d["intermediate"], d["vocab"], d["n_layers"]
    rng = np.random.default_rng(11)
    t0 = time.time()

    def wbits(n, k):                                               # bf16 bit patterns of N(0, 0.02)
        base = rng.standard_normal((min(n, 512), k)).astype(np.float32) * 0.02
        bits = (base.view(np.uint32) >> 16).astype(np.uint16)
        return np.ascontiguousarray(np.tile(bits, (-(-n // bits.shape[0]), 1))[:n])
    one = {"wq": wbits(H * D, hid), "wk": wbits(H * D, hid), "wv": wbits(H * D, hid), "wo": wbits(hid, H * D),
           "w1": wbits(I, hid), "w3": wbits(I, hid), "w2": wbits(hid, I)}
    layers = [{k: v.copy() for k, v in one.items()} for _ in range(NL)]      # distinct memory per layer: nothing stays in the L3
    out_w = wbits(V, hid)
    emb = (rng.standard_normal((1024, hid)) * 0.5).astype(np.float32)
    setup = time.time() - t0
    total = prompt + gen
    kc = [np.zeros((total + 8, H, D), np.float32) for _ in range(NL)]
    vc = [np.zeros((total + 8, H, D), np.float32) for _ in range(NL)]
    L = cref.lib()
    nmax = int(L.orc_num_threads())

    def ln(x):
        return (x - x.mean()) / np.sqrt(x.var() + 1e-5)

    def one_step(tok, n_ctx):
        x = emb[tok % 1024].copy()
        t1 = time.time()
        for l, w in enumerate(layers):
            h = ln(x)
            q, k, v = cref.bf16_gemv(w["wq"], h), cref.bf16_gemv(w["wk"], h), cref.bf16_gemv(w["wv"], h)
            kc[l][n_ctx], vc[l][n_ctx] = k.reshape(H, D), v.reshape(H, D)
            s_ = np.einsum("hd,thd->ht", q.reshape(H, D), kc[l][: n_ctx + 1]) / np.sqrt(D)
            p_ = np.exp(s_ - s_.max(-1, keepdims=True))
            p_ /= p_.sum(-1, keepdims=True)
            a_ = np.einsum("ht,thd->hd", p_, vc[l][: n_ctx + 1]).reshape(-1).astype(np.float32)
            x = x + cref.bf16_gemv(w["wo"], a_)
            h = ln(x)
            g, u = cref.bf16_gemv(w["w1"], h), cref.bf16_gemv(w["w3"], h)
            x = x + cref.bf16_gemv(w["w2"], (g / (1.0 + np.exp(-g)) * u).astype(np.float32))
        lg = cref.bf16_gemv(out_w, ln(x))
        return int(lg.argmax()), time.time() - t1
    # thread count from a short trial on the first prompt tokens (they are part of the prompt either way)
    trial, pos, tok = {}, 0, 17
    for n in sorted({nmax, max(1, nmax // 2), max(1, nmax // 4), max(1, nmax // 8), max(1, nmax // 16)}, reverse=True):
        L.orc_set_num_threads(n)
        _, trial[n] = one_step(int(rng.integers(0, V)), pos)
        pos += 1
        if trial[n] > 20.0:
            break
    nbest = min(trial, key=trial.get)
    L.orc_set_num_threads(nbest)
    t_prompt = 0.0
    while pos < prompt:
        tok, dt = one_step(int(rng.integers(0, V)), pos)
        t_prompt += dt
        pos += 1
    n_prompt_timed = prompt - len(trial)
    t_gen = 0.0
    for _ in range(gen):
        tok, dt = one_step(tok, pos)
        t_gen += dt
        pos += 1
    wb = 2.0 * (NL * (4 * H * D * hid + 3 * I * hid) + V * hid)
    return {"value": round(gen / t_gen, 3), "unit": "tokens/s", "cores": nbest, "kind": "port", "isa": "avx2 (bf16 -> f32 widening, fma)",
            "ctx": f"{prompt}-token prompt, then {gen} greedy tokens (contexts {prompt}..{total - 1})", "steps": gen,
            "config": "BASELINE configs[0]: StableLM-3B bf16 greedy decode, batch 1 (CPU plumbing case), SURVEY 8(d): 128-token prompt + 64 greedy tokens",
            "achieved_GBs": round(wb * gen / t_gen / 1e9, 1), "prompt_tokens_per_s_one_token_steps": round(n_prompt_timed / max(t_prompt, 1e-9), 3),
            "host_threads": nmax,
            "sample": f"full 32-layer StableLM-3B shapes, bf16 weights streamed once per token (f32 accumulation, AVX2 + OpenMP); mean over "
                      f"the {gen} generated tokens at the best of the thread counts tried {({k: round(v, 3) for k, v in trial.items()})} s/step; "
                      f"setup {setup:.1f}s"}


def bench_batch32(gm, cfg, args, perm, blocks_per_seq, stream, kv_per_tok):
    """Secondary measurement of BASELINE's metric at batch 32 (same weights, ragged contexts U[256,4096],
    shuffled block tables, hipGraph replay, greedy tokens read back every step)."""
    import torch
    B, K, Wm = 32, args.b32_steps, 4
    rng = np.random.default_rng(4321)
    seq_lens = rng.integers(256, 4097, B).astype(np.uint32)
    bt = perm[: B * blocks_per_seq].reshape(B, blocks_per_seq).astype(np.uint32)
    tokens = rng.integers(0, cfg.vocab, B).astype(np.uint32)
    st = stream.cuda_stream
    import bench_timing

    def reset():
        gm.decode_begin(tokens, seq_lens, bt, ctx_cap=int(seq_lens.max()) + K + Wm + 2, stream=st)

    def step():
        gm.decode_step(st)
        gm.read_tokens(st)
    tb = bench_timing.timed_blocks(step, torch.cuda.synchronize, K, warmup=Wm, blocks=3, reset=reset, stats=gm.graph_stats)
    dt = tb["median_s"]
    mean_ctx = float(seq_lens.mean()) + Wm + (K - 1) / 2.0
    step_bytes = gm.weight_bytes_global + B * (mean_ctx + 1) * kv_per_tok
    return {"value": round(B * K / dt, 1), "unit": "tokens/s", "batch": B, "steps": K, "ms_per_step": round(1e3 * dt / K, 3),
            "value_min": round(B * K / tb["max_s"], 1), "value_max": round(B * K / tb["min_s"], 1), "value_is": "median of 3 blocks",
            "blocks_ms_per_step": tb["blocks_ms_per_step"], "settle": tb["settle"],
            "graph_captures_in_timed_region": tb["graph_captures_in_timed_region"],
            "mean_ctx": round(mean_ctx, 1), "algorithmic_bytes": int(step_bytes),
            "achieved_GBs": round(step_bytes * K / dt / 1e9, 1),
            "roofline_tok_s_at_8TBs": round(B * HBM_PEAK_GBS * 1e9 / step_bytes, 1)}


def bench_prefill(gm, cfg, perm, blocks_per_seq, T=2048):
    """Secondary: one prompt step over a T-token prompt (the prompt-step GEMM path; MFMA-bound, not HBM-bound)."""
    import torch
    from candle_vllm_amd.block_engine import BlockEngine      # the product's own block manager builds the step inputs
    rng = np.random.default_rng(99)
    nblk = -(-T // cfg.block_size)
    eng = BlockEngine(cfg.block_size, nblk + 1, 0)
    seq = eng.new_sequence(0, rng.integers(0, cfg.vocab, T).tolist())
    eng.allocate([seq])
    meta = eng.prepare_prompt([seq])
    import bench_timing
    tc = bench_timing.timed_calls(lambda: gm.forward_prefill(meta), torch.cuda.synchronize)      # settle by time, then >= 3 calls: median
    dt = tc["median_s"]
    # useful flops of the weight GEMMs: 2 x parameters x tokens from the SHAPES (Q4_K and Q6_K tensors alike: one multiply-add per
    # weight and token), the lm_head on the LAST row only (host_model.cpp runs it on one token per sequence); causal attention counted
    # separately (QK^T and PV over the lower triangle: 2 x 2 x T^2/2 x H x D per layer)
    H, Hkv, D, hid, I = cfg.n_heads, cfg.n_kv_heads, cfg.head_dim, cfg.hidden, cfg.intermediate
    layer_params = 2 * hid * H * D + 2 * hid * Hkv * D + 3 * hid * I
    gemm_flops = 2.0 * (cfg.n_layers * layer_params * T + cfg.vocab * hid * 1)
    attn_flops = cfg.n_layers * 2.0 * 2.0 * (T * (T + 1) / 2.0) * H * D
    useful = gemm_flops / dt / 1e12
    return {"value": round(T / dt, 1), "unit": "prompt tokens/s", "tokens": T, "ms": round(dt * 1e3, 2),
            "value_min": round(T / tc["max_s"], 1), "value_max": round(T / tc["min_s"], 1), "value_is": "median of %d calls" % len(tc["calls_ms"]),
            "calls_ms": tc["calls_ms"], "settle": tc["settle"],
            "useful_TFLOPs": round(useful, 1), "frac_of_2.5PF_dense_f16": round(useful / 2500.0, 3),
            "with_causal_attention_TFLOPs": round((gemm_flops + attn_flops) / dt / 1e12, 1),
            "flops_counted": "2 x (32 layers x %.1f M projection weights x T + lm_head x 1 token); attention's %.1f TFLOP reported separately"
                             % (layer_params / 1e6, attn_flops / 1e12),
            "note": "hand-written quantised GEMM (csrc/qmm_prefill.inc): Q4_K/Q6_K unpacked in registers into f16 MFMA operands, "
                    "no weight image in HBM, no library GEMM; activations ONE f16 plane with a power-of-two scale per (token, "
                    "k-block) = 1 MFMA pass (round 2: hi + lo planes, 2 passes; tuning key 24 restores them); whole prompt step "
                    "incl. prefill attention and epilogues"}


def experiments_leg():
    """A/B of a product dispatch decision, in a SEPARATE process (a fault there cannot touch the judged numbers): the ragged batch-32 step
    with the balanced LDS-DMA attention stream (the step drivers' choice at >= 64 (sequence, kv head) pairs since round 4; tuning key 44
    = 1) against the one-partition MFMA waves it replaced (key 44 = 5), same model, alternated.  Reported only."""
    import subprocess
    env = dict(os.environ, B32_STEPS="16", B32_AB="44=1;44=5;44=1;44=5")
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exp_b32.py")], env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=240)
    except Exception as e:
        return {"error": repr(e)}
    rows = [ln.split() for ln in r.stdout.splitlines() if "tok/s" in ln]
    try:
        base = [float(x[1]) for x in rows if x[0] == "44=1"]
        exp = [float(x[1]) for x in rows if x[0] == "44=5"]
        return {"batch32_ragged_tok_s": {"product_default_lds_dma_attention_stream": base, "one_partition_mfma_waves": exp},
                "note": "A/B of the attention kernel choice at batch 32 (tuning key 44: 1 = default, 5 = never the stream), separate process, "
                        "same model, alternated; `batch32` above is the product",
                "rc": r.returncode}
    except Exception as e:
        return {"error": repr(e), "tail": r.stdout[-400:]}


def parity_leg(mode):
    """The benchmarked geometry (Llama-3-8B shapes, Q4_K_M, ctx 4096, block 64) against the C oracle built from the same
    weight bytes (tests/fullsize_parity.py; the oracle is the CHECKER here, never the thing measured).  PARITY UNPINNED: the
    oracle restates the reference's arithmetic, no reference-held vector exists for the float kernels."""
    import gc
    import torch
    from tests.fullsize_parity import Pair, ragged_batch32
    out = {"oracle": "O1 (unpinned)", "geometry": "Llama-3-8B Q4_K_M, ctx 4096 in paged KV (block 64), batch 1"
                                                   + (" / 32 ragged / 2048-token prompt step" if mode == "full" else "")}
    t0 = time.time()
    pb = Pair(fill_scale=1.0)                                   # the bench's synthetic weight scale
    g = pb.run_parts([4097], o2=0)
    out["launch_groups_b1"] = {k: (round(v, 7) if isinstance(v, float) else v) for k, v in g.items() if k not in ("units", "oracle")}
    out["launch_groups_units"] = g["units"]
    if mode == "full":
        g32 = pb.run_parts(ragged_batch32(np.random.default_rng(4321)), o2=2)
        out["launch_groups_b32"] = {k: (round(v, 7) if isinstance(v, float) else v) for k, v in g32.items() if k not in ("units", "oracle")}
    del pb
    gc.collect(); torch.cuda.empty_cache()
    pt = Pair(fill_scale=0.2)                                   # branch gain < 1, as in a trained checkpoint: end-to-end
    e = pt.run_decode([4097], steps=3, o2=0, graph=True)
    out.update({"max_rel_err": round(e["max_rel_err"], 6), "max_rel_err_vs_bf16_attention": round(e["max_rel_err_vs_bf16_attention"], 6),
                "tokens_equal": bool(e["tokens_equal"]), "near_tie_tokens": int(e["near_tie_tokens"]), "steps_compared": int(e["steps_compared"]),
                "reference_bf16_attention_spread": round(e["reference_bf16_attention_spread"], 6),
                "end_to_end": "3 greedy steps, batch 1, hipGraph replay, logits vs O1; `reference_bf16_attention_spread` = the "
                              "oracle with the reference's bf16 attention tensors (models/mod.rs:1288-1306) vs the f32-attention "
                              "oracle on the same step: the floor under any end-to-end comparison through 32 layers"})
    if mode == "full":
        e32 = pt.run_decode(ragged_batch32(np.random.default_rng(4321)), steps=2, o2=2, graph=True)
        out["batch32"] = {k: (round(v, 6) if isinstance(v, float) else v) for k, v in e32.items()}
        pr = pt.run_prompt(2048)
        out["prompt_step"] = {k: (round(v, 6) if isinstance(v, float) else v) for k, v in pr.items()}
    del pt
    gc.collect(); torch.cuda.empty_cache()
    out["seconds"] = round(time.time() - t0, 1)
    return out


def setup_comm(gm, M, dist, args, rank, world):
    """The tensor-parallel communicator of this run, decided by ALL ranks together -> (transport string, fallback communicator or None).
      1. local pre-flight (no collective inside): librccl loadable; not the same-device rehearsal (RCCL refuses two ranks on one GPU);
         not the forced failure of the rehearsal test (MI355_BENCH_FAIL_COMM_RANK = a rank number).  Minimum over the ranks.
      2. all ranks enter GGUFLLaMa.init_comm (RCCL + the `auto` choice of the decode-sized all-reduce); it raises on every rank or on none.
      3. otherwise, on EVERY rank together: host-supplied collectives staged through the host over a gloo group (tp.TorchDistComm;
         eager steps) -- and, unless the pre-flight failure was a forced one, the one-shot peer kernel over IPC for the all-reduces (C1 / C2)
         with its self-test; the vocabulary gather (C3) stays host-staged.  The run still measures the sharded model and says what carried it.
    The flags travel over a gloo control group of their own, so a failing device communicator cannot take the agreement down with it."""
    import torch
    ctl = dist.new_group(backend="gloo")
    agree = lambda ok: bool(M.comm_all_min(dist, ok, group=ctl))
    mode = "p2p" if args.p2p else args.all_reduce
    fail_rank = os.environ.get("MI355_BENCH_FAIL_COMM_RANK", "")       # the rehearsal test's switch; every rank sees the same environment
    why, forced = None, fail_rank != ""
    if fail_rank == str(rank):
        why = "forced failure of the device communicator on rank %s (MI355_BENCH_FAIL_COMM_RANK)" % fail_rank
    elif args.same_device:
        why = f"{world} ranks share one device: RCCL refuses duplicate GPUs"
    else:
        probe = np.zeros(128, np.uint8)
        if M.lib.mi355_comm_unique_id(probe.ctypes.data) != 0:
            why = "librccl could not be loaded / ncclGetUniqueId failed"
    if agree(why is None):
        try:
            transport, ok = gm.init_comm(dist, p2p={"auto": "auto", "rccl": False, "p2p": True}[mode], wire_bf16=args.wire_bf16), True
        except Exception as e:                                 # init_comm raises on every rank or on none
            transport, ok = repr(e)[:200], False
        if agree(ok):
            return transport, None
        why = "init_comm failed: " + transport
    from candle_vllm_amd import tp as _tp
    comm = _tp.TorchDistComm(ctl)
    comm.set_options(1, 1 if args.wire_bf16 else 0)
    peer = False
    if not forced and mode != "rccl":
        if agree(M.ranks_have_peer_access(dist, world, group=ctl) or args.same_device):
            peer = M.comm_attach_p2p(dist, comm.handle, rank, world, group=ctl)
            if not peer:
                M.lib.mi355_comm_p2p_enable(comm.handle, 0)
    gm.set_comm(comm.handle)
    reasons = [None] * world
    dist.all_gather_object(reasons, why, group=ctl)
    reason = next((r for r in reasons if r and r.startswith("forced")), None) or next((r for r in reasons if r), "unknown")
    transport = ("FALLBACK: " + ("one-shot peer kernel over IPC for the all-reduces (self-test passed on every rank), " if peer else "")
                 + "host-staged collectives over gloo" + (" for the vocabulary gather" if peer else "") + " -- " + reason[:200])
    return transport, comm


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: spawn the N ranks (one process per GPU) the way the reference spawns its own
        # (src/openai/communicator.rs:704-785; the unique id travels by torch.distributed here) -- the same entry script under
        # torch.distributed.run on the loopback address; rank 0 of the children prints the one JSON line
        import socket
        import subprocess
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(sys.argv[0])] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1")))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch `python bench.py --gpus N` (it spawns its ranks) or "
                         f"`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    import torch
    import torch.distributed as dist
    if args.same_device:                                      # rehearsal: every rank on cuda:0 (RCCL refuses two ranks on one GPU: gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.same_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from candle_vllm_amd import model as M
    for kv in filter(None, os.environ.get("MI355_TUNE", "").split(",")):    # experiments only: "key=value,..."
        k, v = kv.split("=")
        M.lib.mi355_set_tuning(int(k), int(v))

    cfg = llama3_8b()
    invalid = None
    if args.layers:
        cfg.n_layers = args.layers
        invalid = "debug run with fewer layers"
    if args.same_device and world > 1:
        invalid = ((invalid + "; ") if invalid else "") + f"{world} ranks on ONE device: a rehearsal of the multi-GPU control flow, not a scaling measurement"
    B, K, Wm = args.batch, args.steps, args.warmup
    do_b32 = (not args.no_batch32) and world == 1 and B == 1
    B32 = 32
    blocks_per_seq = -(-(max(args.ctx, 4096 if do_b32 else 0) + max(K, args.b32_steps) + Wm + 2) // cfg.block_size)
    num_blocks = (B32 if do_b32 else B) * blocks_per_seq + 8
    kv_layout = M.KV_PAGED if args.kv_layout == "paged" else M.KV_FLASH
    gm = M.GGUFLLaMa(cfg, max_batch=(B32 if do_b32 else B), max_blocks_per_seq=blocks_per_seq, kv_layout=kv_layout,
                     tp_rank=rank, tp_world=world)
    transport, comm_fallback = None, None
    if world > 1:
        transport, comm_fallback = setup_comm(gm, M, dist, args, rank, world)
    gm.load_synthetic(seed=1235, recipe="q4_k_m")
    gm.alloc_kv_cache(num_blocks)
    gm.kv_fill_random(seed=7)                                 # (TP: every rank draws the global cache and keeps its kv-head group)

    # block tables: physical blocks in shuffled order (stresses the gather), identical on every rank
    rng = np.random.default_rng(1235)
    perm = rng.permutation(num_blocks - 1) + 1
    bt = perm[: B * blocks_per_seq].reshape(B, blocks_per_seq).astype(np.uint32)
    tokens = rng.integers(0, cfg.vocab, B).astype(np.uint32)
    seq_lens = np.full(B, args.ctx + 1, np.uint32)            # prompt + the first generated token
    stream = torch.cuda.Stream()
    st = stream.cuda_stream
    # TP steps are captured too (RCCL on its side stream joins the capture as a fork / join; --tp-eager keeps them eager)
    graph_mode = (not args.no_graph) and not (world > 1 and (args.tp_eager or comm_fallback is not None))
    if world > 1 and graph_mode:
        # every rank tests LOCALLY whether this stack captures the communicator's all-reduce (capture + instantiate, nothing
        # is launched), then the ranks agree: one rank falling back to eager steps alone would leave its peers inside a
        # collective
        graph_mode = bool(M.comm_all_min(dist, 1 if gm.comm_capture_ok(st) else 0))
    gm.set_graph(graph_mode)
    ctx_cap = args.ctx + K + Wm + 2

    def reset():                                              # the greedy loop back at the benchmark's start state (same graph shape)
        gm.decode_begin(tokens, seq_lens, bt, ctx_cap=ctx_cap, stream=st)

    def step():
        gm.decode_step(st)
        gm.read_tokens(st)                                    # greedy sample -> host every step, as the engine does

    def sync():                                               # the contract's bracket: barrier + device synchronisation
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def reduce_max(dt):
        if world == 1:
            return dt
        t = torch.tensor([dt], dtype=torch.float64, device=M._ctl_device(dist))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # settle by time (fixed count under TP: every step holds collectives), then >= 3 blocks of: reset, W untimed warm-up steps, EXACTLY K
    # timed steps between barrier + synchronize, max over ranks; `value` = the median block (bench_timing.py)
    import bench_timing
    tb = bench_timing.timed_blocks(step, sync, K, warmup=Wm, blocks=args.blocks, reset=reset, stats=gm.graph_stats,
                                   fixed_settle_steps=(args.settle_steps if world > 1 else 0), reduce_max=reduce_max)
    dt = tb["median_s"]
    tok_s = B * K / dt
    # the first greedy tokens of the benchmark's start state (untimed, every rank): a sharded run and the one-GPU run of the same
    # synthetic model can be compared token by token (tests/test_gpu_tp2.py runs both)
    reset()
    head_tokens = []
    for _ in range(min(8, K + Wm)):
        gm.decode_step(st)
        head_tokens.append([int(t) for t in gm.read_tokens(st)])

    # ---- algorithmic bytes of one step: every weight byte once + live KV once (+ the write)
    mean_ctx = args.ctx + 1 + Wm + (K - 1) / 2.0
    kv_per_tok = 2 * cfg.n_layers * cfg.n_kv_heads * cfg.head_dim * 2
    step_bytes = gm.weight_bytes_global + B * (mean_ctx + 1) * kv_per_tok
    achieved = step_bytes * (K / dt) / 1e9

    out = {
        "metric": "decode tokens/s (greedy), Llama-3-8B Q4_K GGUF",
        "value": round(tok_s, 2), "unit": "tokens/s", "n_gpus": world, "steps": K, "warmup": Wm,
        "ms_per_step": round(1e3 * dt / K, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None,
        "dtype": "q4_k/q6_k weights, f32 activations, bf16 KV+attention", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: Llama-3-8B Q4_K_M GGUF shapes, greedy decode, "
                               f"batch={B}, prompt ctx {args.ctx} in paged KV (block 64), {K} decode steps",
                   "batch": B, "ctx_start": args.ctx + 1 + Wm, "ctx_end": args.ctx + Wm + K,
                   "parallelism": f"tp{world}", "graph": bool(graph_mode), "first_tokens": head_tokens,
                   **({"same_device": True} if args.same_device else {}),
                   "timing": {"blocks_ms_per_step": tb["blocks_ms_per_step"], "value_is": "median block",
                              "tokens_per_s_min": round(B * K / tb["max_s"], 2), "tokens_per_s_max": round(B * K / tb["min_s"], 2),
                              "settle": tb["settle"], "graph_captures_in_timed_region": tb.get("graph_captures_in_timed_region"),
                              "eager_steps_in_timed_region": tb.get("eager_steps_in_timed_region"),
                              "graph_captures_total": tb.get("graph_captures_total")},
                   **({"all_reduce": transport, "ranks": world,
                       "wire": ("bf16 (reference numerics)" if args.wire_bf16 else "f32")} if world > 1 else {}),
                   "kv_layout": ("paged K[NB,Hkv,D/8,64,8] V[NB,Hkv,D,64] bf16" if args.kv_layout == "paged"
                                 else "flash [NB,64,Hkv,128] bf16")},
        "step": {"algorithmic_bytes": int(step_bytes), "achieved_GBs": round(achieved, 1),
                 "frac_of_8TBs": round(achieved / HBM_PEAK_GBS, 4), "frac_of_6.29TBs_copy": round(achieved / 6290.0, 4),
                 "roofline_tok_s_at_8TBs": round(B * HBM_PEAK_GBS * 1e9 / step_bytes, 1)},
    }
    if invalid:
        out["invalid"] = invalid
    # every rank runs the per-launch-group timing: under TP the groups contain the all-reduce / all-gather, so a
    # rank-0-only call would wait for its peers forever
    roofline = gm.dominant_kernel_roofline(stream, HBM_PEAK_GBS)
    if rank == 0:
        out["roofline"] = roofline
        # `roofline.frac` describes the DOMINANT launch (the best-behaved one of the step); the whole step's time-weighted fraction
        # -- every algorithmic byte of the step over the step's time -- rides next to it, in `roofline` and in the contract's `config`
        out["roofline"]["whole_step_frac"] = out["step"]["frac_of_8TBs"]
        out["config"]["whole_step_frac_of_8TBs"] = out["step"]["frac_of_8TBs"]
        out["config"]["dominant_launch_frac_of_8TBs"] = roofline["frac"]
        if do_b32:
            out["batch32"] = bench_batch32(gm, cfg, args, perm, blocks_per_seq, stream, kv_per_tok)
            try:
                out["prefill"] = bench_prefill(gm, cfg, perm, blocks_per_seq)
            except Exception as e:                            # secondary number only
                out["prefill"] = {"error": repr(e)}
            # BASELINE's metric is "decode tokens/s @ batch=1,32 ...; achieved HBM GB/s vs peak": `value` is the batch-1 number, the
            # batch-32 number and the prompt-step number ride in `config` so that a record which keeps only the contract keys holds them
            out["config"]["batch32_tokens_per_s"] = out["batch32"]["value"]
            out["config"]["batch32_frac_of_8TBs"] = round(out["batch32"]["achieved_GBs"] / HBM_PEAK_GBS, 4)
            out["config"]["batch1_frac_of_8TBs"] = out["step"]["frac_of_8TBs"]
            if "value" in out["prefill"]:
                out["config"]["prompt_2048_tokens_per_s"] = out["prefill"]["value"]
                out["config"]["prompt_2048_frac_of_dense_f16_peak"] = out["prefill"]["frac_of_2.5PF_dense_f16"]
        if args.legs != "none" and world == 1 and not args.layers and B == 1:
            import gc
            del gm                                            # the legs build their own models
            gc.collect(); torch.cuda.empty_cache()
            import bench_legs
            names = list(bench_legs.LEGS) if args.legs == "all" else [n for n in args.legs.split(",") if n in bench_legs.LEGS]
            out["configs"] = bench_legs.run_legs(names, parity=args.parity != "off")
        if args.parity != "off" and world == 1 and not args.layers:
            try:
                out["parity"] = parity_leg(args.parity)
            except Exception as e:                            # the checker must never sink the measured number
                out["parity"] = {"error": repr(e)}
        if do_b32 and not args.no_experiments and args.legs != "none" and not args.layers:
            out["experiments"] = experiments_leg()
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_steps, args.ctx)
            except Exception as e:                            # the baseline must never sink the GPU number
                out["cpu_baseline"] = {"error": repr(e)}
            try:
                out["cpu_baseline_config0"] = cpu_baseline_config0()
            except Exception as e:
                out["cpu_baseline_config0"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
