"""scratch: hipEvent time of each batch-1 launch group for forced waves-per-workgroup (probe build: mi355_set_tuning(0, nw))"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import llama3_8b
from candle_vllm_amd import model as M
from candle_vllm_amd.ops import _check
lib = M.lib
cfg = llama3_8b()
CTX = 4096
bps = -(-(CTX + 16) // cfg.block_size)
gm = M.GGUFLLaMa(cfg, max_batch=1, max_blocks_per_seq=bps, kv_layout=M.KV_PAGED)
gm.load_synthetic(seed=1235, recipe="q4_k_m")
gm.alloc_kv_cache(bps + 8)
gm.kv_fill_random(seed=7)
stream = torch.cuda.Stream(); st = stream.cuda_stream
bt = (np.arange(bps) + 1).reshape(1, bps).astype(np.uint32)
gm.set_graph(False)
gm.decode_begin(np.array([5], np.uint32), np.full(1, CTX + 1, np.uint32), bt, ctx_cap=CTX + 16, stream=st)
gm.decode_step(st); torch.cuda.synchronize()
for part, name in ((0, "qkv"), (2, "wo"), (3, "gateup"), (4, "down"), (5, "lm_head")):
    row = []
    for nw in (0, 2, 4, 8):
        lib.mi355_set_tuning(0, nw)
        ts = []
        for rep in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(stream):
                e0.record(stream)
                for l in range(cfg.n_layers):
                    _check(lib.mi355_llama_run_part(gm.h, l if part < 5 else 0, part, st), "run_part")
                e1.record(stream)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / cfg.n_layers)
        row.append(f"nw{nw}={sorted(ts)[len(ts) // 2]:.2f}")
    lib.mi355_set_tuning(0, 0)
    print(name, "  ".join(row), flush=True)
