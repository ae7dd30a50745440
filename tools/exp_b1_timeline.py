"""The INSIDE of the batch-1 launches on a timeline: eight in-kernel 100 MHz timestamps per wave of the PRODUCT
single-token mat-vec kernel (qmm_kernel<1, R, ...>; -DMI355_QMM_TIMELINE build: tools/build_variant_lib.sh tl -DMI355_QMM_TIMELINE), for the
launch groups q|k|v (+ RoPE + cache write), wo, gate/up, down of one Llama-3-8B layer at ctx 4096:
   0 wave entry  1 kernarg fields in SGPRs  2 prologue loads issued  3 first k-block staged (x, norm weight arrived)
   4 first (k-block, tile) unit computed (first weights arrived)  5 main loop done  6 past the workgroup barrier  7 epilogue done
Run:  MI355_LIB_PATH=$PWD/build_probe/libmi355vllm_tl.so python tools/exp_b1_timeline.py"""
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
rng = np.random.default_rng(1)
bt = (np.arange(bps) + 1).reshape(1, bps).astype(np.uint32)
gm.set_graph(False)
gm.decode_begin(rng.integers(0, cfg.vocab, 1).astype(np.uint32), np.full(1, CTX + 1, np.uint32), bt, ctx_cap=CTX + 16, stream=st)
gm.decode_step(st); torch.cuda.synchronize()
NWG = 8192
ts = torch.zeros(NWG * 16 * 8, dtype=torch.int64, device="cuda")
_check(lib.mi355_debug_set_timestamps(ts.data_ptr()), "ts")
names = ["entry", "kernarg in SGPRs", "prologue loads issued", "first k-block staged", "first unit computed", "main loop done", "past barrier",
         "epilogue done"]
print("# us since the first wave's entry (100 MHz clock, 10 ns steps); library:", os.environ.get("MI355_LIB_PATH", "product"))
for part, name in ((0, "qkv"), (2, "wo"), (3, "gateup"), (4, "down")):
    for layer in (4, 5, 6):                                  # the last pass is the one reported (the first ones warm code / TLB)
        # the preceding launch group of the step runs first so that the launch under test starts behind a real predecessor
        ts.zero_(); torch.cuda.synchronize()
        _check(lib.mi355_llama_run_part(gm.h, layer, part, st), "run_part")
        torch.cuda.synchronize()
    t = ts.cpu().numpy().reshape(NWG, 16, 8).astype(np.float64)
    live = t[:, :, 0] > 0
    if not live.any():
        print(f"== {name}: no timestamps (this build has no -DMI355_QMM_TIMELINE)")
        continue
    t0 = t[:, :, 0][live].min()
    us = lambda x: (x - t0) / 100.0
    q = lambda a: "min %5.2f  p10 %5.2f  p50 %5.2f  p90 %5.2f  max %5.2f" % (a.min(), *np.percentile(a, [10, 50, 90]), a.max())
    print(f"== {name}: {int(live.sum())} waves in {int(live.any(axis=1).sum())} workgroups; first entry -> last exit {us(t[:, :, 7][live]).max():.2f} us")
    cols = [us(t[:, :, i][live]) for i in range(8)]
    for i in range(8):
        print(f"  {i} {names[i]:24s}", q(cols[i]))
    for (i, j, what) in ((0, 1, "kernarg round trip(s)"), (1, 2, "address arithmetic + issue of every prologue load"), (2, 3, "x / norm weight arrive + stage"),
                         (3, 4, "first weights arrive + first unit"), (4, 5, "rest of the loop"), (5, 6, "LDS reduction + barrier wait"), (6, 7, "epilogue")):
        print(f"  d{i}{j} {what:50s}", q(cols[j] - cols[i]))
    sys.stdout.flush()

# ---- the attention launch (paged_attn_mfma_kernel, fused workgroup merge): -DMI355_PA_TIMELINE builds export mi355_debug_set_pa_timestamps
if hasattr(lib, "mi355_debug_set_pa_timestamps"):
    import ctypes
    lib.mi355_debug_set_pa_timestamps.argtypes = [ctypes.c_void_p]
    lib.mi355_debug_set_pa_timestamps.restype = ctypes.c_int
    _check(lib.mi355_debug_set_pa_timestamps(ts.data_ptr()), "pa ts")
    anames = ["entry", "context length known", "partition done (table, K/V, QK, PV)", "state in LDS + barrier", "workgroup merge, stores issued",
              "stores drained + barrier", "arrival ticket back", "exit (last arriver: merged)"]
    for layer in (4, 5, 6):
        ts.zero_(); torch.cuda.synchronize()
        _check(lib.mi355_llama_run_part(gm.h, layer, 0, st), "run_part")            # q|k|v first: the new token's K / V are in the cache
        torch.cuda.synchronize()
        ts.zero_(); torch.cuda.synchronize()
        _check(lib.mi355_llama_run_part(gm.h, layer, 1, st), "run_part")
        torch.cuda.synchronize()
    t = ts.cpu().numpy().reshape(NWG, 16, 8).astype(np.float64)
    live = t[:, :, 7] > 0
    t0 = t[:, :, 0][live].min()
    us = lambda x: (x - t0) / 100.0
    q = lambda a: "min %5.2f  p10 %5.2f  p50 %5.2f  p90 %5.2f  max %5.2f" % (a.min(), *np.percentile(a, [10, 50, 90]), a.max())
    print(f"== attention: {int(live.sum())} waves in {int(live.any(axis=1).sum())} workgroups; first entry -> last exit {us(t[:, :, 7][live]).max():.2f} us")
    cols = [us(t[:, :, i][live]) for i in range(8)]
    for i in range(8):
        print(f"  {i} {anames[i]:38s}", q(cols[i]))
    for (i, j, what) in ((0, 1, "kernarg + context length"), (1, 2, "table entry -> K / V -> QK^T -> softmax -> P.V of the partition"), (2, 3, "state to LDS + barrier (slowest wave)"),
                         (3, 4, "workgroup merge + write-through partial stores issued"), (4, 5, "store drain + barrier"), (5, 6, "arrival ticket (agent-scope atomic) + barrier"),
                         (6, 7, "exit; the LAST arriver of a kv head: acquire + load 17 partials + merge + store")):
        print(f"  d{i}{j} {what:66s}", q(cols[j] - cols[i]))
    last = (cols[7] - cols[6]) > 0.5
    print(f"  last arrivers: {int(last.sum())} waves; their exit: ", q(cols[7][last]) if last.any() else "-")
