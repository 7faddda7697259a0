import sys, time, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fusion_amd import ops
from fusion_amd.planes import RankedSystem
from fusion_amd.retrievers.hybrid import Aggregator, weight_grid
from fusion_amd.utils import metrics as M
Q, N, S = 195, 27942, 4
g = torch.Generator(device="cuda").manual_seed(0)
T = []
for i in range(S):
    p = ops.alloc_plane(Q, N, torch.float32, "cuda"); p.copy_(torch.rand((Q, N), generator=g, device="cuda")); T.append(p)
od, _, pos = ops.sort_rows_desc(T[0], want_rank=True)
names = ["a", "b", "c", "d"]
grid = weight_grid(names)
W = len(grid)
weights = torch.tensor([[np.float32(w[n]) for n in names] for w in grid], dtype=torch.float32, device="cuda")
rng = np.random.default_rng(0)
gold = np.full((Q, 8), -1, dtype=np.int32)
for q in range(Q):
    k = int(rng.integers(1, 6)); gold[q, :k] = rng.choice(N, size=k, replace=False)
gd = torch.from_numpy(gold).cuda()
def tm(f, n=3):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n, r
t_k, out = tm(lambda: ops.gold_ranks(T, pos, weights, gd))
t_d2h, o2 = tm(lambda: out.cpu().numpy().astype(np.int64))
t_pos, ph = tm(lambda: pos.cpu().numpy())
ranks = o2
t_m, res = tm(lambda: M.metrics_from_gold_ranks(ranks, (gold >= 0).sum(1).astype(np.int64), np.full(Q, N)))
w64 = weights.double()
t_k64, out64 = tm(lambda: ops.gold_ranks(T, pos, w64, gd))
print(json.dumps(dict(kernel_f64w_ms=t_k64 * 1e3)))
print(json.dumps(dict(kernel_ms=t_k * 1e3, out_d2h_ms=t_d2h * 1e3, pos_d2h_ms=t_pos * 1e3, metrics_ms=t_m * 1e3)))
