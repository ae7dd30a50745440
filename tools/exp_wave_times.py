"""scratch: per-wave timestamps of one batch-1 launch group (mi355_set_tuning(2, 7) + mi355_debug_set_timestamps)"""
import os, sys, ctypes
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
rng = np.random.default_rng(1)
bt = (np.arange(bps) + 1).reshape(1, bps).astype(np.uint32)
gm.set_graph(False)
gm.decode_begin(rng.integers(0, cfg.vocab, 1).astype(np.uint32), np.full(1, CTX + 1, np.uint32), bt, ctx_cap=CTX + 16, stream=st)
gm.decode_step(st); torch.cuda.synchronize()
NWG = 4096
ts = torch.zeros(NWG * 16 * 4, dtype=torch.int64, device="cuda")

_check(lib.mi355_debug_set_timestamps(ts.data_ptr()), "ts")
for part, name in ((0, "qkv"), (2, "wo"), (3, "gateup"), (4, "down")):
    for layer in (5, 6):                                  # second pass = the one reported (first warms code / TLB)
        ts.zero_(); torch.cuda.synchronize()
        lib.mi355_set_tuning(2, 7)
        _check(lib.mi355_llama_run_part(gm.h, layer, part, st), "run_part")
        torch.cuda.synchronize()
        lib.mi355_set_tuning(2, 0)
    t = ts.cpu().numpy().reshape(NWG, 16, 4).astype(np.float64)
    live = t[:, :, 0] > 0
    t0 = t[:, :, 0][live].min()
    us = lambda x: (x - t0) / 100.0                         # 100 MHz clock
    start, loop, bar, end = (us(t[:, :, i][live]) for i in range(4))
    q = lambda a: "min %.2f  p10 %.2f  p50 %.2f  p90 %.2f  max %.2f" % (a.min(), *np.percentile(a, [10, 50, 90]), a.max())
    print(f"== {name}: {int(live.sum())} waves in {int(live.any(axis=1).sum())} workgroups")
    print("  wave start      ", q(start))
    print("  main loop done  ", q(loop))
    print("  past barrier    ", q(bar))
    print("  exit            ", q(end))
    print("  loop duration   ", q(loop - start))
    print("  barrier wait    ", q(bar - loop))
    print("  epilogue        ", q(end - bar), flush=True)
