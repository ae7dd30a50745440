"""Batch-1 mat-vec engine (csrc/probes/qmv_engine.inc) against the register-ring kernel (qmm_kernel) at the Llama-3-8B Q4_K_M launch
shapes: same inputs through both, outputs compared, then the hipEvent time of each launch group for a sweep of consumer-wave
counts and ring sizes.   python tools/exp_qmv.py [NC list, default 4,6,8,11,15]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import llama3_8b  # noqa: E402
from candle_vllm_amd import model as M  # noqa: E402
from candle_vllm_amd.ops import _check  # noqa: E402

lib = M.lib
cfg = llama3_8b()
CTX = 4096
bps = -(-(CTX + 16) // cfg.block_size)
gm = M.GGUFLLaMa(cfg, max_batch=1, max_blocks_per_seq=bps, kv_layout=M.KV_PAGED)
gm.load_synthetic(seed=1235, recipe="q4_k_m")
gm.alloc_kv_cache(bps + 8)
gm.kv_fill_random(seed=7)
stream = torch.cuda.Stream()
st = stream.cuda_stream
bt = (np.arange(bps) + 1).reshape(1, bps).astype(np.uint32)
gm.set_graph(False)
gm.decode_begin(np.array([5], np.uint32), np.full(1, CTX + 1, np.uint32), bt, ctx_cap=CTX + 16, stream=st)
lib.mi355_set_tuning(20, 0)
gm.decode_step(st)
torch.cuda.synchronize()

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
hip.hipMemcpy.restype = ctypes.c_int
lib.mi355_llama_act_ptr.restype = ctypes.c_void_p
lib.mi355_qmv_error.argtypes = [ctypes.POINTER(ctypes.c_int32), ctypes.c_int32]
p_xs, p_q, p_att, p_h = [lib.mi355_llama_act_ptr(gm.h, i) for i in range(4)]
hid, I, HD = cfg.hidden, cfg.intermediate, cfg.n_heads * cfg.head_dim
rng = np.random.default_rng(3)


def up(ptr, a):
    a = np.ascontiguousarray(a)
    _check(hip.hipMemcpy(ptr, a.ctypes.data, a.nbytes, 1), "H2D")


def down(ptr, shape, dtype):
    torch.cuda.synchronize()
    a = np.empty(shape, dtype)
    _check(hip.hipMemcpy(a.ctypes.data, ptr, a.nbytes, 2), "D2H")
    return a


def bf16_bits(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def qmv_err():
    v = ctypes.c_int32(0)
    _check(lib.mi355_qmv_error(ctypes.byref(v), 1), "qmv_error")
    return v.value


xs0 = rng.standard_normal(hid).astype(np.float32)
att0 = bf16_bits(rng.standard_normal(HD).astype(np.float32))
h0 = (rng.standard_normal(I) * 0.5).astype(np.float32)
PARTS = ((0, "qkv"), (2, "wo"), (3, "gateup"), (4, "down"), (5, "head"))


def run_once(part, layer):
    """inputs -> run -> the launch group's outputs as one float64 vector"""
    up(p_xs, xs0); up(p_att, att0); up(p_h, h0)
    _check(lib.mi355_llama_run_part(gm.h, layer, part, st), "run_part")
    if part == 0:
        q = down(p_q, (HD,), np.uint16).astype(np.uint32) << 16
        k, v = gm.kv_download(layer)
        kv = [(np.asarray(t).astype(np.uint32).ravel() << 16).view(np.float32).astype(np.float64) for t in (k, v)]
        return np.concatenate([q.view(np.float32).astype(np.float64)] + kv)
    if part == 3:
        return down(p_h, (I,), np.float32).astype(np.float64)
    if part == 5:
        return gm.logits_numpy(1)[0].astype(np.float64)
    return down(p_xs, (hid,), np.float32).astype(np.float64)


ok = True
for part, name in PARTS:
    for layer in ((0, 1, 31) if part != 5 else (0,)):
        lib.mi355_set_tuning(20, 0)
        ref = run_once(part, layer)
        lib.mi355_set_tuning(20, 1)
        for nc in (4, 8, 15):
            for ring in (0, 64):
                lib.mi355_set_tuning(21, nc); lib.mi355_set_tuning(22, ring)
                got = run_once(part, layer)
                err = float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))
                e = qmv_err()
                bad = (not np.isfinite(err)) or err > (2.0 ** -7 if part == 0 else 2e-5) or e != 0
                ok = ok and not bad
                if bad or (nc == 8 and ring == 0):
                    print(f"{name} layer {layer} NC={nc} ring={ring or 128}: max rel diff {err:.3e} engine_error={e} {'BAD' if bad else 'ok'}", flush=True)
lib.mi355_set_tuning(21, 8); lib.mi355_set_tuning(22, 0)
print("engine == register-ring kernel:", ok, flush=True)


def time_part(part, reps=5):
    best = 1e9
    nl = cfg.n_layers if part != 5 else 4
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            e0.record(stream)
            for l in range(nl):
                _check(lib.mi355_llama_run_part(gm.h, l if part != 5 else 0, part, st), "run_part")
            e1.record(stream)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / nl)
    return best


# ---- per-wave cycle accounting of one launch per group (probe mode 15)
ts = torch.zeros(256 * 16 * 4, dtype=torch.int64, device="cuda")
_check(lib.mi355_debug_set_timestamps(ts.data_ptr()), "set_timestamps")
for nc in (8, 4):
    lib.mi355_set_tuning(21, nc)
    for part, name in PARTS:
        up(p_xs, xs0); up(p_att, att0); up(p_h, h0)
        ts.zero_()
        lib.mi355_set_tuning(2, 15)
        _check(lib.mi355_llama_run_part(gm.h, 1, part, st), "run_part")
        torch.cuda.synchronize()
        lib.mi355_set_tuning(2, 0)
        t = ts.cpu().numpy().reshape(256, 16, 4).astype(np.float64)
        live = t[:, nc, 3] > 0                                          # workgroups that ran (loader's exit stamp)
        L, C = t[live, nc, :], t[live, :nc, :]
        t0 = np.minimum(L[:, 0], C[:, :, 0].min(axis=1)).min()
        f = lambda a: f"{np.median(a):8.0f} (max {a.max():8.0f})"
        print(f"{name:7s} NC={nc} wgs={int(live.sum())}: loader total {f(L[:, 3] - L[:, 0])} blocked-on-space {f(L[:, 1])} wait-lag {f(L[:, 2])} | "
              f"consumer entry->loop {f((C[:, :, 2] - C[:, :, 0]).ravel())} loop {f((C[:, :, 3] - C[:, :, 2]).ravel())} of which waiting {f(C[:, :, 1].ravel())} | "
              f"launch span {max(L[:, 3].max(), C[:, :, 3].max()) - t0:8.0f} ticks", flush=True)
lib.mi355_set_tuning(21, 8)
_check(lib.mi355_debug_set_timestamps(None), "set_timestamps")

ncs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4,6,8,11,15").split(",")]
print("us per launch (32 layers back to back, eager):")
for part, name in PARTS:
    lib.mi355_set_tuning(20, 0)
    row = [f"ring-kernel {time_part(part):6.2f}"]
    lib.mi355_set_tuning(20, 1)
    for ring in (0, 64):
        lib.mi355_set_tuning(22, ring)
        for nc in ncs:
            lib.mi355_set_tuning(21, nc)
            row.append(f"NC{nc}/r{ring or 128} {time_part(part):6.2f}")
    lib.mi355_set_tuning(22, 0); lib.mi355_set_tuning(21, 8); lib.mi355_set_tuning(2, 16)
    row.append(f"loader-only {time_part(part):6.2f}")
    lib.mi355_set_tuning(2, 0)
    print(f"{name:7s}", "  ".join(row), " err", qmv_err(), flush=True)
lib.mi355_set_tuning(21, 8); lib.mi355_set_tuning(22, 0)

# whole step, graph replay, engine on / off
for eng in (0, 1):
    lib.mi355_set_tuning(20, eng)
    gm.set_graph(True)
    gm.decode_begin(np.array([5], np.uint32), np.full(1, CTX + 1, np.uint32), bt, ctx_cap=CTX + 16, stream=st)
    for _ in range(2):
        gm.decode_step(st)
    gm.read_tokens(st)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(10):
        gm.decode_step(st)
        gm.read_tokens(st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"graph replay, engine={eng}: {dt * 1e3:.3f} ms/step  {1 / dt:.1f} tok/s  err {qmv_err()}", flush=True)
