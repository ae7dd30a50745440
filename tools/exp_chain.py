"""Batch-1 step with the four mat-vecs between two attention calls chained in one persistent launch (csrc/probes/qmv_chain.inc,
tuning key 23) against the same step launch by launch: logits / tokens compared, then hipGraph-replayed step times and the
chain's per-wave cycle accounting (probe mode 15)."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import llama3_8b  # noqa: E402
from candle_vllm_amd import model as M  # noqa: E402
from candle_vllm_amd.ops import _check  # noqa: E402

lib = M.lib
cfg = llama3_8b()
CTX = int(os.environ.get("CTX", "4096"))
bps = -(-(CTX + 80) // cfg.block_size)
gm = M.GGUFLLaMa(cfg, max_batch=1, max_blocks_per_seq=bps, kv_layout=M.KV_PAGED)
gm.load_synthetic(seed=1235, recipe="q4_k_m")
gm.alloc_kv_cache(bps + 8)
gm.kv_fill_random(seed=7)
stream = torch.cuda.Stream()
st = stream.cuda_stream
bt = (np.arange(bps) + 1).reshape(1, bps).astype(np.uint32)


def qmv_err():
    v = ctypes.c_int32(0)
    _check(lib.mi355_qmv_error(ctypes.byref(v), 1), "qmv_error")
    return v.value


def steps(chain, n, graph):
    lib.mi355_set_tuning(23, chain)
    gm.set_graph(False)
    gm.set_graph(graph)
    gm.kv_fill_random(seed=7)
    gm.decode_begin(np.array([5], np.uint32), np.full(1, CTX + 1, np.uint32), bt, ctx_cap=CTX + 64, stream=st)
    toks, lg = [], []
    for _ in range(n):
        gm.decode_step(st)
        toks.append(int(gm.read_tokens(st)[0]))
        lg.append(gm.logits_numpy(1)[0].copy())
    return toks, lg


for graph in (False, True):
    t0, l0 = steps(0, 4, graph)
    t1, l1 = steps(1, 4, graph)
    d = [float(np.abs(a - b).max()) for a, b in zip(l0, l1)]
    print(f"graph={graph}: tokens launch-by-launch {t0} chained {t1}  max |dlogit| per step {d}  nan {int(np.isnan(l1[0]).sum())}  engine_error {qmv_err()}", flush=True)

for chain in (0, 1):
    lib.mi355_set_tuning(23, chain)
    gm.set_graph(False)
    gm.set_graph(True)
    gm.decode_begin(np.array([5], np.uint32), np.full(1, CTX + 1, np.uint32), bt, ctx_cap=CTX + 64, stream=st)
    for _ in range(4):
        gm.decode_step(st)
        gm.read_tokens(st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 40
    for _ in range(N):
        gm.decode_step(st)
        gm.read_tokens(st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    print(f"graph replay, chained={chain}: {dt * 1e3:.3f} ms/step  {1 / dt:.1f} tok/s  engine_error {qmv_err()}", flush=True)

# ---- cycle accounting of the chained launches of one eager step (the buffer keeps the LAST chained launch = layer 31 + lm_head;
# and of a middle layer by running the chain of layer 5 alone)
ts = torch.zeros(256 * 16 * 4, dtype=torch.int64, device="cuda")
_check(lib.mi355_debug_set_timestamps(ts.data_ptr()), "set_timestamps")
lib.mi355_set_tuning(23, 1)
gm.set_graph(False)
gm.decode_begin(np.array([5], np.uint32), np.full(1, CTX + 1, np.uint32), bt, ctx_cap=CTX + 64, stream=st)
lib.mi355_set_tuning(2, 15)
gm.decode_step(st)
torch.cuda.synchronize()
lib.mi355_set_tuning(2, 0)
t = ts.cpu().numpy().reshape(256, 16, 4).astype(np.float64)
NC = 8
live = t[:, NC, 3] > 0
L, C = t[live, NC, :], t[live, :NC, :]
f = lambda a: f"{np.median(a):8.0f} (max {a.max():8.0f})"
print(f"last chain (layer 31: wo, gate/up, down, lm_head) wgs={int(live.sum())}: loader total {f(L[:, 3] - L[:, 0])} blocked-on-space {f(L[:, 1])} "
      f"wait-lag {f(L[:, 2])} | consumer total {f((C[:, :, 3] - C[:, :, 0]).ravel())} waiting-for-data {f(C[:, :, 1].ravel())} at-edges {f(C[:, :, 2].ravel())}", flush=True)
# per phase (consumer wave 0 of every workgroup): edge wait, image build, unit loop, and the gap to the next phase's start
ph = t[live, 9:13, :]
base0 = ph[:, 0, 0]
for p_i, name in enumerate(("wo", "gate/up", "down", "lm_head")):
    st0, edge, img, loop = ph[:, p_i, 0], ph[:, p_i, 1], ph[:, p_i, 2], ph[:, p_i, 3]
    nxt = ph[:, p_i + 1, 0] if p_i < 3 else C[:, 0, 3]
    print(f"  phase {name:8s}: starts at {f(st0 - base0)}  edge wait {f(edge - st0)}  image {f(img - edge)}  units {f(loop - img)}  epilogue+arrive {f(nxt - loop)}", flush=True)
_check(lib.mi355_debug_set_timestamps(None), "set_timestamps")
print("engine_error", qmv_err())
