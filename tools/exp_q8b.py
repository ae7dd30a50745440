import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_vllm_amd.ops as cv
from candle_vllm_amd import _lib as _L
from oracle import kquants as kq
lib = _L.lib
rng = np.random.default_rng(3)
t = kq.GGML_Q6_K
N, K = 16, 256
blocks = kq.quantize(rng.normal(0, 0.05, (N, K)).astype(np.float32), t)
W = kq.dequantize(blocks, t).reshape(N, K)
mm = cv.QMatMul(blocks, t, "cuda")
lib.mi355_set_tuning(18, 1)
try:
    for e in (0, 1, 3, 4, 5, 15, 16, 17, 31, 32, 33, 64, 96, 128, 130, 255):
        x = np.zeros((1, K), np.float32); x[0, e] = 1.0
        y = mm.forward(torch.from_numpy(x).cuda()).cpu().numpy()[0]
        # which column of W does y look like?
        errs = np.abs(W - y[:, None]).max(axis=0)
        best = int(errs.argmin())
        print(f"e={e}: matches column {best} (err {errs[best]:.2e}); err at own column {errs[e]:.2e}; y[0..3]={y[:4]}  W[0..3,e]={W[:4, e]}")
finally:
    lib.mi355_set_tuning(18, 0)
