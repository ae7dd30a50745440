"""scratch: hipEvent time of each batch-1 launch group under probe modes (mi355_set_tuning(2, v))"""
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
modes = [int(x) for x in os.environ.get("DBG", "0,1,3,4").split(",")]
for part, name in ((0, "qkv"), (2, "wo"), (3, "gateup"), (4, "down")):
    row = []
    for dbg in modes:
        lib.mi355_set_tuning(2, dbg)
        best = 1e9
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(stream):
                e0.record(stream)
                for l in range(cfg.n_layers):
                    _check(lib.mi355_llama_run_part(gm.h, l, part, st), "run_part")
                e1.record(stream)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / cfg.n_layers)
        row.append(f"dbg{dbg}={best:.2f}")
    lib.mi355_set_tuning(2, 0)
    print(name, "  ".join(row), flush=True)
