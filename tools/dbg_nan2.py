import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from candle_vllm_amd import ops as cv
from oracle import kquants as kq
res = []
for nkb in [int(x) for x in os.environ.get("NKB", "8,16,24,32,40,48,56,57,58,64,72").split(",")]:
    for N in (256, 4096):
        K = nkb * 256
        rng = np.random.default_rng(nkb)
        blocks = kq.quantize(rng.normal(0, 0.05, (N, K)).astype(np.float32), kq.GGML_Q4_K)
        x = rng.normal(0, 1, (1, K)).astype(np.float32)
        mm = cv.QMatMul(blocks, kq.GGML_Q4_K, "cuda")
        got = mm.forward(torch.from_numpy(x).cuda()).cpu().numpy()
        ref = kq.qmatmul_o1(x, blocks, kq.GGML_Q4_K)
        res.append((nkb, N, int(np.isnan(got).sum()), float(np.nanmax(np.abs(got - ref)) / np.abs(ref).max()) if not np.isnan(got).all() else float("nan")))
        print(res[-1], flush=True)
