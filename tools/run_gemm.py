"""Score GEMM and the vendor fp32 GEMM on one large shape, a few launches each (driven by tools/pmc_gemm.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fusion_amd import ops

Q, N, d = 1024, 276307, 768
g = torch.Generator(device="cuda").manual_seed(1)
Qn = ops.normalize_rows(torch.randn((Q, d), generator=g, device="cuda"))
Dn = ops.normalize_rows(torch.randn((N, d), generator=g, device="cuda"))
out = ops.alloc_plane(Q, N, torch.float32, "cuda")
ref = torch.empty((Q, N), device="cuda")
for _ in range(4):
    ops.dot_scores(Qn, Dn, out=out)
    torch.mm(Qn, Dn.t(), out=ref)
torch.cuda.synchronize()
