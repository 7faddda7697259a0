import sys, numpy as np, torch
sys.path.insert(0, ".")
from fusion_amd import ops
rng = np.random.default_rng(7)
g = torch.Generator(device="cuda").manual_seed(2)
for it in range(12):
    H = int(rng.integers(1, 13))
    lens = rng.integers(1, int(rng.choice([20, 70, 600])), int(rng.integers(1, 12)))
    T = int(lens.sum())
    qkv = torch.randn((T, 3 * H * 64), generator=g, device="cuda")
    strips, cu = ops.attn_strips(lens)
    sd = torch.from_numpy(strips).cuda()
    outs = [ops.attn_varlen(qkv, sd, H).clone() for _ in range(4)]
    torch.cuda.synchronize()
    ne = [(outs[0] != o).sum().item() for o in outs[1:]]
    md = max(((outs[0] - o).abs() / outs[0].abs().clamp_min(1e-30)).max().item() for o in outs[1:])
    idx = (outs[0] != outs[1]).nonzero()
    where = ""
    if idx.numel():
        r, c = idx[0].tolist(); b = int(np.searchsorted(cu, r, side="right") - 1)
        where = f"first at row {r - int(cu[b])} of L={int(lens[b])}, col {c} (head {c // 64}, dim {c % 64}); values {outs[0][r, c].item():.9g} vs {outs[1][r, c].item():.9g}; rows differing: {sorted(set((idx[:, 0] - int(cu[b])).tolist()))[:10]}"
    print("H", H, "T", T, "n_strips", len(strips), "differing elements per rerun", ne, "max rel diff", f"{md:.2e}", where, flush=True)
# the same for the float16 form and for the score GEMM (bit-identical reruns)
import itertools
bad = 0
for it in range(20):
    H = int(rng.integers(1, 13)); lens = rng.integers(100, 600, int(rng.integers(4, 12))); T = int(lens.sum())
    qkv16 = torch.randn((T, 3 * H * 64), generator=g, device="cuda").half()
    strips, cu = ops.attn_strips(lens); sd = torch.from_numpy(strips).cuda()
    outs = []
    for _ in range(3):
        o = torch.empty((T, H * 64), dtype=torch.float16, device="cuda"); ops.attn_varlen_f16(qkv16, sd, H, o); outs.append(o)
    ref = ops.attn_varlen(qkv16.float(), sd, H).half()
    bad += sum(int(not torch.equal(outs[0], o)) for o in outs[1:]) + int(not torch.equal(outs[0], ref))
print("float16 attention: reruns or the float32 twin that differ:", bad)
