"""End-to-end distance to the oracle against model depth, three ways:
   product : the product's mat-vec kernels (x as two bf16 pieces, 2^-17 per product) + reference-faithful attention numerics
   exact   : parity mode 2 -- every mat-vec of the step exact to f32 rounding (csrc/qmm_exact.inc) + the same attention
   self    : the oracle against itself, mat-vecs summed in f64 (O1) vs blocked f32 (O1f), bf16 attention tensors on both sides
at Llama-3-8B Q4_K_M shapes, batch 1, ctx 4097, "trained" weight scale (fill 0.2), n_layers = 1 .. 32.  One greedy step each.
Run on the GPU box:  python tools/exp_parity_depth.py [depths ...] > profiles/r05_parity_depth.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import gc
    import torch
    from tests.fullsize_parity import Pair
    from oracle.llama import LlamaConfig
    depths = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32]
    seeds = [int(s) for s in os.environ.get("DEPTH_SEEDS", "1235,77").split(",")]
    print("# distance of the step's logits to the oracle (max |got - ref| / max |ref| of the row), batch 1, ctx 4097, Llama-3-8B Q4_K_M shapes,")
    print("# reference-faithful attention numerics on both sides; product = the product mat-vec kernels, exact = parity mode 2,")
    print("# self = oracle O1 (f64 sums) vs oracle O1f (blocked f32 sums).  One greedy step per cell; tokens = greedy token equal to the oracle's.")
    print(f"{'layers':>6} {'seed':>6} {'product':>11} {'exact':>11} {'self':>11} {'tokens (product / exact)':>26}")
    for d in depths:
        for seed in seeds:
            cfg = LlamaConfig.llama3_8b()
            cfg.n_layers = d
            p = Pair(cfg=cfg, seed=seed, fill_scale=0.2, max_batch=32 if os.environ.get('DEPTH_BATCH32') else 1)
            r_prod = p.run_decode_faithful([4097], steps=1, o2=0, graph=False)
            r_ex = p.run_decode_faithful([4097], steps=1, o2=0, graph=False, exact=True)
            print(f"{d:>6} {seed:>6} {r_prod['max_rel_err']:>11.3e} {r_ex['max_rel_err']:>11.3e} {r_prod['oracle_self_spread_f64_vs_f32_dots']:>11.3e} "
                  f"{str(r_prod['tokens_equal']) + ' / ' + str(r_ex['tokens_equal']):>26}", flush=True)
            if os.environ.get("DEPTH_BATCH32") and d <= int(os.environ["DEPTH_BATCH32"]):
                from tests.fullsize_parity import ragged_batch32
                lens = ragged_batch32(np.random.default_rng(4321))
                b_prod = p.run_decode_faithful(lens, steps=1, o2=2, graph=False)
                b_ex = p.run_decode_faithful(lens, steps=1, o2=2, graph=False, exact=True)
                print(f"{d:>6} {seed:>6} {b_prod['max_rel_err']:>11.3e} {b_ex['max_rel_err']:>11.3e} {'(batch 32)':>11} "
                      f"{str(b_prod['tokens_equal']) + ' / ' + str(b_ex['tokens_equal']):>26}   ragged batch 32, oracle O1f", flush=True)
            del p
            gc.collect(); torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
