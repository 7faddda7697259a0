"""Encoder forward at the bench's shape: HF-per-bucket vs FusedBertForward vs PackedBertForward (ms, max |diff|)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fusion_amd import encoders

Q = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
enc = encoders.random_init("dpr", "cuda", size="base")
rng = np.random.default_rng(0)
lens = np.clip(rng.normal(36, 14, Q).round().astype(np.int64), 4, 64)
ids = np.full((Q, 64), 1, dtype=np.int64)
for i, L in enumerate(lens):
    ids[i, :L] = rng.integers(7, 32000, size=L)
I = torch.from_numpy(ids).cuda()
M = (torch.arange(64, device="cuda")[None, :] < torch.from_numpy(lens).cuda()[:, None]).long()

def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3, r

t_b, a = timeit(lambda: enc.encode_ids_bucketed(I, M, lens, 8))
t_f, b = timeit(lambda: enc.encode_ids_fused(I, lens, 8))
t_p, c = timeit(lambda: enc.encode_ids_packed(I, lens))
print({"tokens": int(lens.sum()), "hf_bucketed_ms": round(t_b, 2), "fused_ms": round(t_f, 2), "packed_ms": round(t_p, 2),
       "fused_vs_hf": float((a - b).abs().max()), "packed_vs_hf": float((a - c).abs().max()), "emb_absmax": float(a.abs().max())})
