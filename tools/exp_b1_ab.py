"""Same-box A/B of library builds on the batch-1 step (BASELINE configs[1]): every build in MI355_LIBS (comma list of name=path; "product" =
the in-tree library) runs in its OWN subprocess, alternated ROUNDS times; per run: settle by time, then 3 blocks of 128 graph-replayed
steps (median), plus the hipEvent time of each launch group.  Prints one line per run."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json, time
import numpy as np, torch
sys.path.insert(0, %r)
import bench_timing
from bench import llama3_8b
from candle_vllm_amd import model as M
cfg = llama3_8b()
B = int(os.environ.get("AB_BATCH", "1"))
CTX, K, Wm = 4096, int(os.environ.get("AB_STEPS", "128")), 8
bps = -(-(CTX + K + Wm + 2) // cfg.block_size)
gm = M.GGUFLLaMa(cfg, max_batch=B, max_blocks_per_seq=bps, kv_layout=M.KV_PAGED)
gm.load_synthetic(seed=1235, recipe="q4_k_m")
gm.alloc_kv_cache(B * bps + 8)
gm.kv_fill_random(seed=7)
rng = np.random.default_rng(1235)
perm = rng.permutation(B * bps + 7) + 1
bt = perm[: B * bps].reshape(B, bps).astype(np.uint32)
tok = rng.integers(0, cfg.vocab, B).astype(np.uint32)
sl = np.full(B, CTX + 1, np.uint32) if B == 1 else np.random.default_rng(4321).integers(256, 4097, B).astype(np.uint32)
stream = torch.cuda.Stream(); st = stream.cuda_stream
gm.set_graph(True)
def reset(): gm.decode_begin(tok, sl, bt, ctx_cap=int(sl.max()) + K + Wm + 2, stream=st)
def step(): gm.decode_step(st); gm.read_tokens(st)
tb = bench_timing.timed_blocks(step, torch.cuda.synchronize, K, warmup=Wm, blocks=3, reset=reset, stats=gm.graph_stats)
out = {"ms_per_step": round(1e3 * tb["median_s"] / K, 4), "tok_s": round(B * K / tb["median_s"], 1), "blocks": tb["blocks_ms_per_step"]}
if B == 1:
    r = gm.dominant_kernel_roofline(stream, 8000.0, reps=5)
    out["groups_us"] = [g["avg_us"] for g in r["groups"]]
print("RESULT " + json.dumps(out), flush=True)
''' % ROOT
libs = [kv.split("=", 1) for kv in os.environ.get("MI355_LIBS", "product=").split(",")]
rounds = int(os.environ.get("ROUNDS", "2"))
for r in range(rounds):
    for name, path in libs:
        env = dict(os.environ)
        if path:
            env["MI355_LIB_PATH"] = path if os.path.isabs(path) else os.path.join(ROOT, path)
        else:
            env.pop("MI355_LIB_PATH", None)
        p = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        res = [ln[7:] for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
        print(f"{name:12s} round {r}: " + (res[0] if res else "FAILED rc %d: %s" % (p.returncode, p.stdout[-400:])), flush=True)
