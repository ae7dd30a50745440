"""scratch: only the ragged batch-32 decode leg of bench.py (for rocprofv3 runs and A/B of mi355_set_tuning keys)"""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                           # noqa: E402
from candle_vllm_amd import model as M                 # noqa: E402

for kv in filter(None, os.environ.get("MI355_TUNE", "").split(",")):
    k, v = kv.split("=")
    M.lib.mi355_set_tuning(int(k), int(v))
cfg = bench.llama3_8b()
args = types.SimpleNamespace(b32_steps=int(os.environ.get("B32_STEPS", "24")))
bps = -(-(4096 + args.b32_steps + 4 + 2) // cfg.block_size)
nb = 32 * bps + 8
gm = M.GGUFLLaMa(cfg, max_batch=32, max_blocks_per_seq=bps, kv_layout=M.KV_PAGED)
gm.load_synthetic(seed=1235, recipe="q4_k_m")
gm.alloc_kv_cache(nb)
gm.kv_fill_random(seed=7)
perm = np.random.default_rng(1235).permutation(nb - 1) + 1
stream = torch.cuda.Stream()
gm.set_graph(True)
kv_per_tok = 2 * cfg.n_layers * cfg.n_kv_heads * cfg.head_dim * 2
# B32_AB="5=0;5=256;5=512,44=1": one measurement per ';'-separated group of tuning keys, same model and cache (the graph is dropped in
# between: the partition size is baked into the captured launches); B32_ROUNDS repeats the whole list (drift check)
def bench_b1(K=48, Wm=6, ctx=4000):
    """batch 1 at a 4 k context on the same model: tok/s of K graph-replayed greedy steps"""
    import time
    bt1 = perm[: bps].reshape(1, bps).astype(np.uint32)
    st = stream.cuda_stream
    gm.decode_begin(np.array([17], np.uint32), np.array([ctx], np.uint32), bt1, ctx_cap=ctx + K + Wm + 2, stream=st)
    for _ in range(Wm):
        gm.decode_step(st)
        gm.read_tokens(st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        gm.decode_step(st)
        gm.read_tokens(st)
    torch.cuda.synchronize()
    return K / (time.perf_counter() - t0)


ab = [g for g in os.environ.get("B32_AB", "").split(";") if g]
if not ab:
    print(bench.bench_batch32(gm, cfg, args, perm, bps, stream, kv_per_tok), flush=True)
for rnd in range(int(os.environ.get("B32_ROUNDS", "1")) if ab else 0):
    for grp in ab:
        for kv in grp.split(","):
            k, v = kv.split("=")
            M.lib.mi355_set_tuning(int(k), int(v))
        gm.set_graph(False)
        gm.set_graph(True)
        r = bench.bench_batch32(gm, cfg, args, perm, bps, stream, kv_per_tok)
        b1 = f" | batch 1: {bench_b1():.1f} tok/s" if os.environ.get("B32_B1") else ""
        print(grp, r["value"], "tok/s", r["ms_per_step"], "ms/step" + b1, flush=True)
