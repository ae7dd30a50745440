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
print(bench.bench_batch32(gm, cfg, args, perm, bps, stream, kv_per_tok), flush=True)
