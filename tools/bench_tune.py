#!/usr/bin/env python3
"""Config 4: 4-way tuned weighted-linear fusion -- the whole 1771-vector weight grid (hybrid.py:404-426) on the device."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fusion_amd import ops
from fusion_amd.planes import RankedSystem
from fusion_amd.retrievers.hybrid import Aggregator, weight_grid

def main(Q=195, N=27942, S=4):
    g = torch.Generator(device="cuda").manual_seed(0)
    names = ["bm25", "dpr", "splade", "colbert"][:S]
    ids = np.arange(1, N + 1)
    systems = {}
    for i, n in enumerate(names):
        p = ops.alloc_plane(Q, N, torch.float32, "cuda"); p.copy_(torch.randn((Q, N), generator=g, device="cuda") * (i + 1))
        od, _, rk = ops.sort_rows_desc(p, want_rank=True)
        systems[n] = RankedSystem(scores=p, order=od, rank=rk, lens=torch.full((Q,), N, dtype=torch.int32, device="cuda"), ids=ids, full=True)
    rng = np.random.default_rng(0)
    labels = [rng.choice(ids, size=int(rng.integers(1, 6)), replace=False).tolist() for _ in range(Q)]
    grid = weight_grid(names)
    for norm in ("min-max", "z-score"):
        Aggregator.tune(systems, norm, grid[:8], labels, {})
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = Aggregator.tune(systems, norm, grid, labels, {})
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        # one weight vector the reference way (fuse + sort + evaluate), for scale
        t1 = time.perf_counter()
        for w in grid[:4]:
            f = Aggregator.fuse(systems, "nsf", norm, w, {}, as_device=True); f.predictions(1000)
        torch.cuda.synchronize(); per = (time.perf_counter() - t1) / 4
        print(json.dumps(dict(workload=f"tune {norm}", S=S, Q=Q, N=N, W=len(grid), total_s=round(dt, 4), ms_per_weight_vector=round(1e3 * dt / len(grid), 4),
                              fuse_sort_eval_ms_per_weight_vector=round(1e3 * per, 2), best_recall10=max(r["recall@10"] for r in res))), flush=True)

if __name__ == "__main__":
    main()
