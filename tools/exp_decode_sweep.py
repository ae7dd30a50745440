"""scratch experiment: decode tok/s at several batch sizes under different mi355_set_tuning settings,
plus a hot-vs-cold probe of one launch group (same layer's weights re-read = Infinity-Cache resident)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import llama3_8b                       # noqa: E402
from candle_vllm_amd import model as M            # noqa: E402
from candle_vllm_amd.ops import _check            # noqa: E402

lib = M.lib
cfg = llama3_8b()
CTX, K, Wm = 4096, 64, 8
# TUNES: ';'-separated settings, each "key=val,key=val" for mi355_set_tuning (keys not named are reset to 0)
# defaults of the mi355_set_tuning keys (qmatmul.hip / paged_attention.hip): 0 NW, 1 R, 2 probe mode, 3 fused attention
# merge, 5 partition override, 8 attention waves per workgroup, 9 launch chaining, 10 K-split target
DEFAULTS = {0: 0, 1: 0, 2: 0, 3: 1, 5: 0, 8: 0, 9: 1, 10: 1024, 14: 1, 15: 0, 17: 2, 18: 0}
modes = os.environ.get("TUNES", "0=0;0=4;0=8;1=1;1=2;1=4;0=4,1=4;0=4,1=1;0=0").split(";")
batches = [int(x) for x in os.environ.get("PF_BATCHES", "1,32").split(",")]
Bmax = max(batches)
bps = -(-(CTX + K + Wm + 2) // cfg.block_size)
nb = Bmax * bps + 8
gm = M.GGUFLLaMa(cfg, max_batch=Bmax, max_blocks_per_seq=bps, kv_layout=M.KV_PAGED)
gm.load_synthetic(seed=1235, recipe="q4_k_m")
gm.alloc_kv_cache(nb)
gm.kv_fill_random(seed=7)
stream = torch.cuda.Stream()
st = stream.cuda_stream
gm.set_graph(True)
rng = np.random.default_rng(1235)
perm = rng.permutation(nb - 1) + 1
out = {}
for B in batches:
    bt = perm[: B * bps].reshape(B, bps).astype(np.uint32)
    for mode in modes:
        gm.set_graph(False); gm.set_graph(True)          # tuning is baked into a captured graph
        for k, v in DEFAULTS.items():                    # every knob back to its default before the mode's own settings
            lib.mi355_set_tuning(k, v)
        for kv in mode.split(","):
            k, v = kv.split("=")
            lib.mi355_set_tuning(int(k), int(v))
        tokens = rng.integers(0, cfg.vocab, B).astype(np.uint32)
        seq_lens = np.full(B, CTX + 1, np.uint32) if B == 1 else rng.integers(256, 4096, B).astype(np.uint32)
        gm.set_graph(False)                              # one eager step per mode: scratch a mode needs is created outside a capture
        gm.decode_begin(tokens, seq_lens, bt, ctx_cap=CTX + K + Wm + 2, stream=st)
        gm.decode_step(st); gm.read_tokens(st)
        gm.set_graph(True)
        gm.decode_begin(tokens, seq_lens, bt, ctx_cap=CTX + K + Wm + 2, stream=st)
        for _ in range(Wm):
            gm.decode_step(st); gm.read_tokens(st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            gm.decode_step(st); gm.read_tokens(st)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[f"B{B}_mode{mode}"] = round(B * K / dt, 1)
        print(f"B={B} mode={mode}: {B * K / dt:.1f} tok/s  {dt / K * 1e3:.3f} ms/step", flush=True)
for k, v in DEFAULTS.items():
    lib.mi355_set_tuning(k, v)
if os.environ.get("NO_PROBE"):
    sys.exit(0)
# hot vs cold: launch group `part` of ONE layer 32x (weights stay in the Infinity Cache) vs of 32 layers in turn
gm.decode_begin(rng.integers(0, cfg.vocab, 1).astype(np.uint32), np.full(1, CTX + 1, np.uint32),
                perm[:bps].reshape(1, bps).astype(np.uint32), ctx_cap=CTX + K + Wm + 2, stream=st)
gm.set_graph(False)
gm.decode_step(st)
torch.cuda.synchronize()
for part, name in ((0, "qkv"), (2, "wo"), (3, "gateup"), (4, "down")):
    res = {}
    for kind in ("cold", "hot"):
        best = 1e9
        for rep in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(stream):
                e0.record(stream)
                for l in range(cfg.n_layers):
                    _check(lib.mi355_llama_run_part(gm.h, l if kind == "cold" else 3, part, st), "run_part")
                e1.record(stream)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / cfg.n_layers)
        res[kind] = round(best, 2)
    out["probe_" + name] = res
    print(name, res, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/exp_prefetch.json", "w"), indent=1)
