"""scratch: does the gate/up launch of the batch-1 step pay for 896 workgroups on 256 CUs (3.5 per CU: half the CUs hold 4, half 3)?
hipEvent time per launch of the gate/up and down groups for intermediate sizes that put 2 / 2.5 / 3 / 3.5 / 4 / 4.5 / 5 workgroups on a CU.
If the time follows the bytes, the launch is bandwidth-bound; if 3.5 costs what 4 costs, the heaviest CU sets it."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import llama3_8b
from candle_vllm_amd import model as M
from candle_vllm_amd.ops import _check
lib = M.lib
CTX = 256
stream = torch.cuda.Stream(); st = stream.cuda_stream
for inter in [int(x) for x in os.environ.get("INTER", "8192,10240,12288,14336,16384,18432,20480").split(",")]:
    cfg = llama3_8b()
    cfg.intermediate = inter
    cfg.n_layers = 12
    bps = -(-(CTX + 16) // cfg.block_size)
    gm = M.GGUFLLaMa(cfg, max_batch=1, max_blocks_per_seq=bps, kv_layout=M.KV_PAGED)
    gm.load_synthetic(seed=1235, recipe="q4_k")
    gm.alloc_kv_cache(bps + 8)
    gm.kv_fill_random(seed=7)
    bt = (np.arange(bps) + 1).reshape(1, bps).astype(np.uint32)
    gm.set_graph(False)
    gm.decode_begin(np.array([5], np.uint32), np.full(1, CTX + 1, np.uint32), bt, ctx_cap=CTX + 16, stream=st)
    gm.decode_step(st); torch.cuda.synchronize()
    out = []
    for part, name in ((3, "gateup"), (4, "down")):
        ts = []
        for rep in range(9):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(stream):
                e0.record(stream)
                for l in range(cfg.n_layers):
                    _check(lib.mi355_llama_run_part(gm.h, l, part, st), "run_part")
                e1.record(stream)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / cfg.n_layers)
        mb = (2 if part == 3 else 1) * inter * 4096 * 0.5625 / 1e6
        out.append(f"{name} {np.median(ts):6.2f} us  {mb:6.1f} MB  {mb / np.median(ts) / 1e3 * 1e3:5.2f} GB/ms")
    print(f"intermediate {inter:6d}: pairs/CU {inter / 16 / 256:4.2f}   " + "   ".join(out), flush=True)
    del gm
    torch.cuda.empty_cache()
