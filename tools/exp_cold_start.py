#!/usr/bin/env python
"""What is the slow mode the driver's round-5 run caught on the Mixtral batch-32 leg (2229 tok/s against 4411-4454 everywhere else)?

That leg timed 8 steps after 2 warm-up steps, once, ~45 s after the GPU had last been busy (the previous leg's CPU-side oracle check).
This script reproduces the situation on purpose and prints every step's own duration (each step ends in the token read-back = a stream
synchronisation), so the transient is visible step by step:
  series A  right after the model was built: begin, then 60 steps (step 1 eager, step 2 captures + instantiates, the rest replay)
  series B  after IDLE_S seconds (default 45) of an idle GPU: 60 more replays of the SAME graph
  series C  a fresh begin (same shape: no capture) right after B
and the sclk / mclk rocm-smi reports before and after the idle phase.  LEG=mixtral_b32 (default) | llama_b1 | llama_b32."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def clocks():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=20)
        return " | ".join(ln.strip() for ln in r.stdout.splitlines() if "sclk" in ln or "mclk" in ln or "fclk" in ln)
    except Exception as e:
        return repr(e)


def cpu_stat():
    """cgroup CPU bandwidth counters: nr_throttled / throttled_usec grow when the process group used up its quota inside a period and was
    frozen until the period ended (CFS bandwidth control)"""
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            d = dict(ln.split() for ln in open(path).read().splitlines())
            return {k: int(v) for k, v in d.items() if "throttl" in k or k in ("nr_periods", "usage_usec")}
        except (OSError, ValueError):
            continue
    return {}


def series(tag, step, n):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        step()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{tag}: " + " ".join(f"{t:.2f}" for t in ts), flush=True)
    print(f"{tag}: first 8 mean {np.mean(ts[:8]):.3f} ms, last 8 mean {np.mean(ts[-8:]):.3f} ms, min {min(ts):.3f}", flush=True)
    return ts


def main():
    import torch
    import bench_legs
    from candle_vllm_amd import model as M
    leg = os.environ.get("LEG", "mixtral_b32")
    idle = float(os.environ.get("IDLE_S", "45"))
    n = int(os.environ.get("N_STEPS", "60"))
    rng = np.random.default_rng(5)
    if leg == "mixtral_b32":
        B = 32
        bps = -(-(4096 + 3 * n + 8) // 64)
        gm, cfg, _, rng, gen = bench_legs._build_mixtral(B, bps)
    else:
        B = 32 if leg == "llama_b32" else 1
        cfg = M.ModelDims.llama3_8b()
        bps = -(-(4096 + 3 * n + 8) // 64)
        gm = M.GGUFLLaMa(cfg, max_batch=B, max_blocks_per_seq=bps, kv_layout=M.KV_PAGED)
        gm.load_synthetic(seed=1235, recipe="q4_k_m")
    ctxs = [4097] if B == 1 else np.random.default_rng(4321).integers(256, 4097, B).tolist()
    nblk = [-(-(int(c) + 3 * n + 8) // 64) for c in ctxs]
    gm.alloc_kv_cache(sum(nblk) + 8)
    perm = rng.permutation(sum(nblk) + 7) + 1
    bt = np.zeros((B, bps), np.uint32)
    o = 0
    for i, k in enumerate(nblk):
        bt[i, :k] = perm[o:o + k]
        o += k
    stream = torch.cuda.Stream()
    st = stream.cuda_stream
    gm.set_graph(True)
    tok0 = rng.integers(0, cfg.vocab, B).astype(np.uint32)
    cap = 4096 + 3 * n + 8

    def begin():
        gm.decode_begin(tok0, np.asarray(ctxs, np.uint32), bt, ctx_cap=cap, stream=st)

    def step():
        gm.decode_step(st)
        gm.read_tokens(st)
    print("leg", leg, "batch", B, "idle_s", idle, flush=True)
    print("clocks before A:", clocks(), flush=True)
    begin()
    series("A (fresh model: eager, capture, replays)", step, n)
    print("captures / eager steps:", gm.graph_stats(), flush=True)
    print("clocks after A :", clocks(), flush=True)
    time.sleep(idle)
    print("clocks after idle:", clocks(), flush=True)
    series(f"B (after {idle:.0f} s idle, same graph)", step, n)
    begin()
    series("C (re-begun, same shape)", step, n)
    print("captures / eager steps:", gm.graph_stats(), flush=True)
    # the host side of an idle phase: a CPU-bound phase in THIS process (what the oracle check is), then replays
    try:
        print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip(), "| visible CPUs:", os.cpu_count(), flush=True)
    except OSError:
        pass
    print("cpu.stat before the CPU-bound phase:", cpu_stat(), flush=True)
    t0 = time.time()
    x = np.random.default_rng(0).standard_normal((3000, 3000))
    while time.time() - t0 < min(idle, 20.0):
        x = x @ x.T / 3000.0
    print("cpu.stat after the CPU-bound phase :", cpu_stat(), flush=True)
    begin()
    t0 = time.perf_counter(); step(); dt1 = (time.perf_counter() - t0) * 1e3
    print(f"first step after the CPU-bound phase: {dt1:.2f} ms; cpu.stat now:", cpu_stat(), flush=True)
    series("D (after a CPU-bound phase of this process)", step, n)
    # the same with the BLAS / OpenMP pools parked first (threadpoolctl): if the stall is the quota, limiting the burst removes it
    try:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=8):
            t0 = time.time()
            while time.time() - t0 < 8.0:
                x = x @ x.T / 3000.0
        print("cpu.stat after an 8-thread CPU phase:", cpu_stat(), flush=True)
        begin()
        t0 = time.perf_counter(); step(); dt2 = (time.perf_counter() - t0) * 1e3
        print(f"first step after the 8-thread CPU phase: {dt2:.2f} ms", flush=True)
    except Exception as e:
        print("threadpoolctl leg skipped:", repr(e), flush=True)


if __name__ == "__main__":
    main()
