"""Which launch group of the batch-1 step produces non-finite values (and how often)?  Full-size Llama-3-8B Q4_K_M shapes, ctx 4096, eager
launch groups through mi355_llama_run_part, every activation buffer checked after every group.  NW = 0 (heuristic) and forced 2 / 4 / 8."""
import ctypes, os, sys
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
gm.load_synthetic(seed=1235, recipe="q4_k_m", scale=0.2)
gm.alloc_kv_cache(bps + 8)
gm.kv_fill_random(seed=7)
stream = torch.cuda.Stream(); st = stream.cuda_stream
bt = (np.arange(bps) + 1).reshape(1, bps).astype(np.uint32)
gm.set_graph(False)
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
lib.mi355_llama_act_ptr.restype = ctypes.c_void_p
H, D, hid, I = cfg.n_heads, cfg.head_dim, cfg.hidden, cfg.intermediate
bufs = {"xs": (0, np.float32, hid), "q": (1, np.uint16, H * D), "attn": (2, np.uint16, H * D), "h": (3, np.float32, I)}


def bad(name):
    which, dt, n = bufs[name]
    a = np.empty(n, dt)
    hip.hipMemcpy(a.ctypes.data, lib.mi355_llama_act_ptr(gm.h, which), a.nbytes, 2)
    if dt == np.uint16:
        a = (a.astype(np.uint32) << 16).view(np.float32)
    m = ~np.isfinite(a)
    return int(m.sum()), (np.nonzero(m)[0][:6].tolist() if m.any() else []), float(np.abs(a[~m]).max()) if (~m).any() else 0.0


reads = {0: ["q"], 1: ["attn"], 2: ["xs"], 3: ["h"], 4: ["xs"]}
names = {0: "qkv", 1: "attention", 2: "wo", 3: "gateup", 4: "down"}
for nw in (0, 8, 4, 2):
    lib.mi355_set_tuning(0, nw)
    fails = {}
    for rep in range(int(os.environ.get("REPS", "6"))):
        gm.decode_begin(np.array([5 + rep], np.uint32), np.full(1, CTX + 1, np.uint32), bt, ctx_cap=CTX + 16, stream=st)
        _check(lib.mi355_llama_run_part(gm.h, 0, 6, st), "embed")
        for l in range(cfg.n_layers):
            for part in range(5):
                _check(lib.mi355_llama_run_part(gm.h, l, part, st), "run_part")
                torch.cuda.synchronize()
                for b in reads[part]:
                    n, idx, mx = bad(b)
                    if n:
                        fails.setdefault((names[part], b), []).append((rep, l, n, idx, mx))
            if fails:
                break
        if fails:
            break
    print(f"NW={nw}: ", "clean" if not fails else {k: v[:3] for k, v in fails.items()}, flush=True)
lib.mi355_set_tuning(0, 0)
