"""ColBERT query / document encode with float16 Linears (colbert-ai's autocast) against the float32 forward: time, token-vector and MaxSim-score
differences.  Usage: python tools/bench_colbert_amp.py [Q]"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fusion_amd import encoders, ops
from tools.bench_kernels import timeit

Q = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.cuda.tunable.enable(False)
enc = encoders.random_init("colbert", device="cuda", size="base", seed=2)
rng = np.random.default_rng(0)
ids = torch.from_numpy(rng.integers(7, 32000, size=(Q, 64))).cuda()
res = {}
toks = {}
for name, amp in (("f32", False), ("f16", True)):
    enc.amp = amp
    toks[name] = enc.encode_query_ids(ids).float()
    res[name + "_ms"] = round(timeit(lambda: enc.encode_query_ids(ids), n=5, warm=2), 3)
# documents: 512 docs of ~300 tokens
L = np.clip(rng.normal(300, 120, 512), 16, 512).astype(np.int64)
dids = torch.from_numpy(rng.integers(7, 32000, size=(512, 512))).cuda()
D = {}
for name, amp in (("f32", False), ("f16", True)):
    enc.amp = amp
    D[name] = enc.encode_doc_ids(dids, L)
    res["doc_" + name + "_ms"] = round(timeit(lambda: enc.encode_doc_ids(dids, L), n=3, warm=1), 3)
res["f16_tok_max_abs_diff"] = float((toks["f16"] - toks["f32"]).abs().max())
S32 = ops.maxsim(toks["f32"].half(), D["f32"][0], D["f32"][1], max_doc_len=512)
S16 = ops.maxsim(toks["f16"].half(), D["f16"][0], D["f16"][1], max_doc_len=512)
res["maxsim_score_max_abs_diff_f16"] = float((S32 - S16).abs().max())
res["maxsim_score_scale"] = float(S32.abs().mean())
o32, o16 = S32.argsort(1, descending=True), S16.argsort(1, descending=True)
res["top10_overlap"] = float(np.mean([len(set(a[:10].tolist()) & set(b[:10].tolist())) / 10 for a, b in zip(o32.cpu(), o16.cpu())]))
print(json.dumps(dict(Q=Q, **res)), flush=True)
