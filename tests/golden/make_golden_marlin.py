"""Generate tests/golden/marlin_perms.json by IMPORTING the reference's in-tree Python
(/root/reference/examples/convert_awq_marlin.py): scale perms, pack/unpack, Marlin zero points.
Run in the build container only (the GPU box has no /root/reference):  python tests/golden/make_golden_marlin.py
"""
import importlib.util
import json
import os

import numpy as np
import torch

REF = "/root/reference/examples/convert_awq_marlin.py"
spec = importlib.util.spec_from_file_location("ref_convert_awq_marlin", REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

rng = np.random.default_rng(20260924)
out = {"source": "examples/convert_awq_marlin.py", "cases": []}
sp, sps = ref.get_scale_perms()
out["scale_perm"] = [int(v) for v in sp]
out["scale_perm_single"] = [int(v) for v in sps]
for (G, N) in [(1, 64), (2, 128), (4, 256), (3, 192)]:
    zp = rng.integers(0, 16, size=(G, N)).astype(np.int32)
    packed_cols = ref.pack_cols(torch.from_numpy(zp), 4, G, N).numpy().astype(np.uint32)
    unpacked = ref.unpack_cols(torch.from_numpy(packed_cols.astype(np.int32)), 4, G, N).numpy()
    assert (unpacked == zp).all()
    mzp = ref.marlin_zero_points(torch.from_numpy(zp), G, N, 4).numpy().astype(np.uint32)
    # an AWQ-packed qzeros tensor (nibble i of word c = zp[8c + order[i]]) and its conversion
    order = np.array([0, 2, 4, 6, 1, 3, 5, 7])
    awq_cols = zp.reshape(G, N // 8, 8)[:, :, order].reshape(G, N)
    awq_packed = ref.pack_cols(torch.from_numpy(awq_cols), 4, G, N)
    a2m = ref.awq_to_marlin_zero_points(awq_packed, G, N, 4).numpy().astype(np.uint32)
    out["cases"].append({
        "G": G, "N": N, "zp": zp.tolist(), "pack_cols": packed_cols.tolist(),
        "marlin_zero_points": mzp.tolist(),
        "awq_packed": awq_packed.numpy().astype(np.uint32).tolist(),
        "awq_to_marlin_zero_points": a2m.tolist(),
    })
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "marlin_perms.json")
with open(dst, "w") as f:
    json.dump(out, f)
print("wrote", dst, os.path.getsize(dst), "bytes")
