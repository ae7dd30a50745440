"""Why does the host baseline (bench.py cpu_baseline: the C port of the reference's CPU arithmetic, oracle/oracle.c) not scale with
threads on the GPU box (0.154 / 0.121 / 0.144 / 0.115 / 0.136 s per step at 8 / 16 / 32 / 64 / 128 threads)?  Prints what the
host offers (sockets, cores, affinity, cgroup quota) and the per-phase step times at several thread counts under a few OpenMP
placements.  Run on the GPU box: python tools/exp_cpu_baseline.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import ctypes
    import time
    import numpy as np
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    from oracle import cref, ops as O
    from candle_vllm_amd.model import ModelDims, q4km_type_for
    cfg = ModelDims.llama3_8b()
    cfg.n_layers = int(os.environ.get("EXP_LAYERS", "8"))
    ctx = 4096
    names = ["wq", "wk", "wv", "wo", "w1", "w2", "w3"]
    types = [q4km_type_for(n, l, cfg.n_layers) for l in range(cfg.n_layers) for n in names] + [q4km_type_for("output", 0, cfg.n_layers)]
    cref.build()
    m = cref.CLlama(cfg, W=None, types=types, seed=1235)
    nblk = -(-(ctx + 40) // cfg.block_size) + 1
    rng = np.random.default_rng(3)
    shape = (nblk, cfg.block_size, cfg.n_kv_heads, cfg.head_dim)
    kb = (rng.standard_normal(shape).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    cache = [(kb.copy(), kb.copy()) for _ in range(cfg.n_layers)]
    toks = [int(t) for t in rng.integers(0, cfg.vocab, ctx)]
    L = cref.lib()
    ph = (ctypes.c_double * 3)()
    for n in [int(x) for x in os.environ.get("EXP_THREADS", "4,8,16,32,64,128").split(",")]:
        if n > int(L.orc_num_threads()) and n > (os.cpu_count() or 1):
            continue
        L.orc_set_num_threads(n)
        ts = []
        for rep in range(3):
            meta = O.prepare_decode([{"tokens": toks, "block_table": list(range(nblk))}], cfg.block_size)
            L.orc_llama_phase_times(ph, 1)
            t0 = time.time()
            m.decode(meta, cache, o2=True)
            ts.append((time.time() - t0, ph[0], ph[1], ph[2]))
            L.orc_llama_phase_times(ph, 1)
            ts[-1] = (ts[-1][0], ph[0], ph[1], ph[2])
        best = min(ts)
        print(f"   threads {n:4d}: step {best[0] * 1e3:8.1f} ms   mat-vecs {best[1] * 1e3:8.1f}   attention {best[2] * 1e3:8.1f}   rest {best[3] * 1e3:7.1f}", flush=True)


if __name__ == "__main__":
    if os.environ.get("EXP_CHILD"):
        child()
        sys.exit(0)
    import bench
    print("host:", json.dumps(bench.host_cpus()))
    try:
        print("loadavg:", open("/proc/loadavg").read().strip())
    except OSError:
        pass
    for tag, env in (("default (passive wait)", {}), ("OMP_PROC_BIND=spread OMP_PLACES=cores", {"OMP_PROC_BIND": "spread", "OMP_PLACES": "cores"}),
                     ("OMP_PROC_BIND=close OMP_PLACES=cores", {"OMP_PROC_BIND": "close", "OMP_PLACES": "cores"})):
        print(f"{tag}  ({os.environ.get('EXP_LAYERS', '8')} layers of Llama-3-8B Q4_K_M + lm_head, ctx 4096, batch 1):", flush=True)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, EXP_CHILD="1", **env), stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=600)
        print(r.stdout.rstrip()[-2500:], flush=True)
