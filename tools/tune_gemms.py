#!/usr/bin/env python3
"""Record TunableOp results for the packed encoder's Linear shapes (run on the GPU box; writes gpurun_out/gemm_gfx950.csv,
to be copied to fusion_amd/tuned/gemm_gfx950.csv).  Shapes: camembert-base Linears at the padded row counts the benches hit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.cuda.tunable as tn
out = os.path.join(ROOT, "gpurun_out", "gemm_gfx950.csv")
os.makedirs(os.path.dirname(out), exist_ok=True)
tn.enable(True); tn.tuning_enable(True); tn.set_max_tuning_duration(int(os.environ.get("TUNE_MS", "40"))); tn.set_filename(out)
have = os.path.join(ROOT, "fusion_amd", "tuned", "gemm_gfx950.csv")
if os.path.exists(have):
    tn.read_file(have)          # keep what is already recorded: the file written at exit holds both
args = [a for a in sys.argv[1:] if a != "--f16"]
dt = torch.float16 if "--f16" in sys.argv else torch.float32      # --f16: the mixed-precision (ColBERT) forward's Linears, e.g. rows 65536 12800
rows = [int(a) for a in args] or [36352, 36864, 37376, 37888, 6656, 7168, 7680, 4608, 5120]
g = torch.Generator(device="cuda").manual_seed(0)
for M in rows:
    for (N, K) in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
        x = torch.randn((M, K), generator=g, device="cuda").to(dt); w = torch.randn((N, K), generator=g, device="cuda").to(dt); b = torch.randn(N, generator=g, device="cuda").to(dt)
        torch.nn.functional.linear(x, w, b)
        torch.cuda.synchronize()
        print("tuned", M, N, K, dt, flush=True)
print("TunableOp writes", out, "when the process exits")
