import sys, ctypes, numpy as np, torch
sys.path.insert(0, ".")
from fusion_amd import ops, _lib
L = _lib.lib()
rng = np.random.default_rng(7)
g = torch.Generator(device="cuda").manual_seed(2)
found = 0
for it in range(40):
    H = int(rng.integers(1, 13))
    lens = rng.integers(100, 600, int(rng.integers(4, 12)))
    T = int(lens.sum())
    qkv = torch.randn((T, 3 * H * 64), generator=g, device="cuda")
    strips, cu = ops.attn_strips(lens)
    sd = torch.from_numpy(strips).cuda()
    dbg = [torch.zeros((len(strips), H, 2, 64, 2), device="cuda") for _ in range(2)]
    outs = []
    for k in range(2):
        L.fz_dbg_set(ctypes.c_void_p(dbg[k].data_ptr()))
        outs.append(ops.attn_varlen(qkv, sd, H).clone())
        torch.cuda.synchronize()
    L.fz_dbg_set(ctypes.c_void_p(0))
    if not torch.equal(outs[0], outs[1]):
        found += 1
        dm = (dbg[0][..., 0] != dbg[1][..., 0]); dl = (dbg[0][..., 1] != dbg[1][..., 1])
        print("H", H, "lens", lens.tolist(), "out elems differ", (outs[0] != outs[1]).sum().item(), "| m differs in", dm.sum().item(), "lanes, l differs in", dl.sum().item(), "lanes")
        idx = dm.nonzero()[:6].tolist()
        for (st, h, u, ln) in idx:
            print("   strip", strips[st].tolist(), "head", h, "u", u, "lane", ln, "(r", ln % 16, "kg", ln // 16, ") m", dbg[0][st, h, u, ln, 0].item(), dbg[1][st, h, u, ln, 0].item(), "l", dbg[0][st, h, u, ln, 1].item(), dbg[1][st, h, u, ln, 1].item())
        if found >= 3: break
print("found", found)
