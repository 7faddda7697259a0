import sys, numpy as np, torch
sys.path.insert(0, ".")
from fusion_amd import ops
rng = np.random.default_rng(5)
g = torch.Generator(device="cuda").manual_seed(1)
bad = 0
for it in range(300):
    H = int(rng.integers(1, 13))
    lens = rng.integers(1, int(rng.choice([20, 70, 600])), int(rng.integers(1, 12)))
    T = int(lens.sum())
    sc = float(rng.choice([0.3, 1.0, 3.0]))
    qkv16 = (torch.randn((T, 3 * H * 64), generator=g, device="cuda") * sc).half()
    strips, cu = ops.attn_strips(lens)
    sd = torch.from_numpy(strips).cuda()
    ctx16 = torch.empty((T, H * 64), dtype=torch.float16, device="cuda")
    ops.attn_varlen_f16(qkv16, sd, H, ctx16)
    again = torch.empty_like(ctx16); ops.attn_varlen_f16(qkv16, sd, H, again)
    ref32 = ops.attn_varlen(qkv16.float(), sd, H)
    ref32b = ops.attn_varlen(qkv16.float(), sd, H)
    ref = ref32.half()
    if not torch.equal(ctx16, ref):
        bad += 1
        idx = (ctx16 != ref).nonzero()
        r, c = idx[0].tolist()
        b = int(np.searchsorted(cu, r, side="right") - 1); L = int(lens[b]); h = c // 64
        blk = qkv16[cu[b]: cu[b] + L].double().view(L, 3, H, 64)
        q, k, v = blk[:, 0, h], blk[:, 1, h], blk[:, 2, h]
        ref64 = (torch.softmax(q @ k.T / 8.0, -1) @ v)[r - cu[b], c % 64].item()
        print("H", H, "L", L, "row_in_seq", r - int(cu[b]), "f16 kernel", ctx16[r, c].item(), "f32 kernel", ref32[r, c].item(), "-> half", ref[r, c].item(), "fp64", ref64,
              "| f16 deterministic", torch.equal(ctx16, again), "f32 deterministic", torch.equal(ref32, ref32b), flush=True)
        if bad > 6: break
print("bad", bad)
