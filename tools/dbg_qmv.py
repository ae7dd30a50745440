"""TEMP: engine debug on one plain mat-vec (STORE epilogue)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from candle_vllm_amd import ops
from candle_vllm_amd.ops import _check
from oracle import kquants as KQ
lib = ops.lib
dev = "cuda"
rng = np.random.default_rng(0)
for (N, K, t) in ((4096, 4096, 12), (256, 4096, 14), (4096, 14336, 12)):
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    blocks = KQ.quantize_q4_k(w) if t == 12 else KQ.quantize_q6_k(w)
    qm = ops.QMatMul(blocks, t, dev)
    x = torch.from_numpy(rng.standard_normal((1, K)).astype(np.float32)).to(dev)
    outs = {}
    for mode in (("ring", 0, 0), ("eng", 1, 0), ("eng_global_tiles", 1, 11), ("eng_cmp", 1, 13), ("eng_ximg", 1, 12), ("eng_singles", 1, 14)):
        lib.mi355_set_tuning(20, mode[1]); lib.mi355_set_tuning(2, mode[2])
        y = qm.forward(x)
        torch.cuda.synchronize()
        outs[mode[0]] = y.cpu().numpy()[0].astype(np.float64)
    lib.mi355_set_tuning(2, 0)
    ref = outs["ring"]
    for k in ("eng", "eng_global_tiles", "eng_singles"):
        o = outs[k]
        print(N, K, t, k, "nan", int(np.isnan(o).sum()), "maxrel", float(np.nanmax(np.abs(o - ref)) / np.abs(ref).max()), "first", o[:4], ref[:4])
    c = outs["eng_cmp"]
    print(N, K, t, "mismatching words per row tile: total", float(np.nansum(c[::16])), "tiles with mismatch", int((c[::16] != 0).sum()), "of", N // 16, "first tiles", c[::16][:8])
    xi = outs["eng_ximg"]
    print(N, K, t, "ximg sum of hi plane:", xi[:2], "expected ~", float(x.sum()))
