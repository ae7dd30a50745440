import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from candle_vllm_amd import ops as cv
from oracle import kquants as kq
for (T, N, K) in [(1, 4096, 4096), (1, 4096, 2048), (1, 2048, 4096), (1, 4096, 14336)]:
    for t in (kq.GGML_Q4_K, kq.GGML_Q6_K):
        rng = np.random.default_rng(12 + T + N)
        blocks = kq.quantize(rng.normal(0, 0.05, (N, K)).astype(np.float32), t)
        x = rng.normal(0, 1, (T, K)).astype(np.float32)
        mm = cv.QMatMul(blocks, t, "cuda")
        ref = kq.qmatmul_o1(x, blocks, t)
        for nw in (0, 2, 4, 8):
            cv.lib.mi355_set_tuning(0, nw)
            got = mm.forward(torch.from_numpy(x).cuda()).cpu().numpy()
            print(T, N, K, t, "nw", nw, "nan", int(np.isnan(got).sum()), "err", float(np.nanmax(np.abs(got - ref)) / np.abs(ref).max()), flush=True)
        cv.lib.mi355_set_tuning(0, 0)
