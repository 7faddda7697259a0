"""The three row sorts of the default bench step at its shape (Q = 1024, N = 27,942), for profiler passes (tools/pmc_sort.sh)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fusion_amd import ops

Q, N = 1024, 27942
g = torch.Generator(device="cuda").manual_seed(0)
k32 = ops.alloc_plane(Q, N, torch.float32, "cuda"); k32.copy_(torch.rand((Q, N), generator=g, device="cuda") * 2 - 1)
# BM25-like float64 scores: ~40 % exact zeros, heavy tail
k64 = ops.alloc_plane(Q, N, torch.float64, "cuda")
k64.copy_((torch.distributions.Gamma(0.5, 0.25).sample((Q, N)).to("cuda").double() - 2.0).clamp_min(0.0))
for _ in range(5):
    ops.sort_rows_desc(k32, want_keys=False, want_rank=True)
    ops.sort_rows_desc(k64, want_keys=False, want_rank=True)
torch.cuda.synchronize()
