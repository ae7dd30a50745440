"""scratch: per launch-group timing of the batch-1 decode step under tuning settings (SETS: ';'-separated "k=v,k=v")"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import llama3_8b                       # noqa: E402
from candle_vllm_amd import model as M            # noqa: E402
from candle_vllm_amd.ops import _check            # noqa: E402

lib = M.lib
cfg = llama3_8b()
B = int(os.environ.get("B", "1"))
CTX = 4096
bps = -(-(CTX + 64) // cfg.block_size)
nb = B * bps + 8
gm = M.GGUFLLaMa(cfg, max_batch=B, max_blocks_per_seq=bps, kv_layout=M.KV_PAGED)
gm.load_synthetic(seed=1235, recipe="q4_k_m")
gm.alloc_kv_cache(nb)
gm.kv_fill_random(seed=7)
stream = torch.cuda.Stream()
st = stream.cuda_stream
rng = np.random.default_rng(1235)
perm = rng.permutation(nb - 1) + 1
gm.set_graph(False)
gm.decode_begin(rng.integers(0, cfg.vocab, B).astype(np.uint32), np.full(B, CTX + 1, np.uint32),
                perm[:B * bps].reshape(B, bps).astype(np.uint32), ctx_cap=CTX + 64, stream=st)
gm.decode_step(st)
torch.cuda.synchronize()
for setting in os.environ.get("SETS", "2=0;2=1").split(";"):
    for kv in setting.split(","):
        k, v = kv.split("=")
        lib.mi355_set_tuning(int(k), int(v))
    row = {}
    for part, name in ((0, "qkv"), (1, "attn"), (2, "wo"), (3, "gateup"), (4, "down")):
        best = 1e9
        for rep in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(stream):
                e0.record(stream)
                for l in range(cfg.n_layers):
                    _check(lib.mi355_llama_run_part(gm.h, l, part, st), "run_part")
                e1.record(stream)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / cfg.n_layers)
        row[name] = round(best, 2)
    print(setting, row, "sum", round(sum(row.values()), 1), flush=True)
