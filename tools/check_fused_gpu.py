import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fusion_amd import encoders
enc = encoders.random_init("dpr", device="cuda", size="base")
rng = np.random.default_rng(0)
n, L = 256, 64
lens = rng.integers(8, L + 1, n)
ids = rng.integers(7, 32000, (n, L)); mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64); ids = np.where(mask == 1, ids, 1)
I, M = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
a = enc.encode_ids(I, M); b = enc.encode_ids_fused(I, lens, 8)
print("base-size fused vs HF max abs diff:", float((a - b).abs().max()), "mean |emb|:", float(a.abs().mean()))
an, bn = torch.nn.functional.normalize(a), torch.nn.functional.normalize(b)
print("cosine between the two forwards: min", float((an * bn).sum(1).min()))
