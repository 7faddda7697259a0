"""fz_attn_varlen_f32 alone at the bench's shape (1024 queries, LLeQA-like length mix, 12 heads): time + effective HBM rate."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fusion_amd import _lib, ops
if os.environ.get("FZ_LIB"):          # ablation builds (tools/ablate)
    _lib.LIB_PATH = os.path.abspath(os.environ["FZ_LIB"])

Q = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mode = sys.argv[3] if len(sys.argv) > 3 else "queries"
rng = np.random.default_rng(0)
if mode.startswith("fixed:"):
    lens = np.full(Q, int(mode[6:]), dtype=np.int64)
else:
    lo, hi, mu, sd = (4, 64, 36, 14) if mode == "queries" else (16, 512, 300, 120)
    lens = np.clip(rng.normal(mu, sd, Q).round().astype(np.int64), lo, hi)
T = int(lens.sum())
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn((T, 2304), generator=g, device="cuda")
strips, cu = ops.attn_strips(lens)
sd_ = torch.from_numpy(strips).cuda()
out = torch.empty((T, 768), device="cuda")
for _ in range(3): ops.attn_varlen(qkv, sd_, 12, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): ops.attn_varlen(qkv, sd_, 12, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
flops = 4.0 * 64 * 12 * float((lens.astype(np.float64) ** 2).sum())
print({"T": T, "strips": len(strips), "ms": round(ms, 4), "GB/s": round(T * 3072 * 4 / ms / 1e6, 1), "TFLOP/s(useful)": round(flops / ms / 1e9, 2)})
