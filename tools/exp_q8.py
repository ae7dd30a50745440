"""scratch: the Q8_K-activation experiment (mi355_set_tuning(18, 1)) against the O2 oracle, and batch-1 decode speed"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_vllm_amd.ops as cv
from candle_vllm_amd import _lib as _L
from oracle import kquants as kq
lib = _L.lib
def rel(a, b): return float(np.abs(a - b).max() / np.abs(b).max())
rng = np.random.default_rng(3)
try:
    for t, name in ((kq.GGML_Q4_K, "Q4_K"), (kq.GGML_Q6_K, "Q6_K")):
        for N, K in ((64, 1024), (4096, 4096), (512, 14336)):
            blocks = kq.quantize(rng.normal(0, 0.05, (N, K)).astype(np.float32), t)
            mm = cv.QMatMul(blocks, t, "cuda")
            x = rng.normal(size=(1, K)).astype(np.float32)
            xd = torch.from_numpy(x).cuda()
            lib.mi355_set_tuning(18, 0)
            y0 = mm.forward(xd).cpu().numpy()
            lib.mi355_set_tuning(18, 1)
            y1 = mm.forward(xd).cpu().numpy()
            o1 = kq.qmatmul_o1(x, blocks, t)
            o2 = kq.qmatmul_o2(x, blocks, t)
            print(f"{name} N={N} K={K}: f16 path vs O1 {rel(y0, o1):.2e} | q8 path vs O2 {rel(y1, o2):.2e}  vs O1 {rel(y1, o1):.2e} | O2 vs O1 {rel(o2, o1):.2e}", flush=True)
finally:
    lib.mi355_set_tuning(18, 0)
