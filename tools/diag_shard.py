"""Stage times of one mMARCO 1/8 shard search without overlap (HIP events at every mark)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections, torch
from fusion_amd import ops
from fusion_amd.distributed import ShardedDenseIndex

Q, N, d, k = 1024, 8841823 // 8, 768, 1000
g = torch.Generator(device="cuda").manual_seed(4)
Dn = torch.empty((N, d), dtype=torch.float32, device="cuda")
for c0 in range(0, N, 1 << 19):
    c1 = min(N, c0 + (1 << 19))
    Dn[c0:c1] = ops.normalize_rows(torch.randn((c1 - c0, d), generator=g, device="cuda"))
Qn = ops.normalize_rows(torch.randn((Q, d), generator=g, device="cuda"))
idx = ShardedDenseIndex(Dn, 0)
for head in [int(x) for x in sys.argv[1:]] or [ShardedDenseIndex.HEAD]:
  idx.HEAD = head
  print('HEAD', head)
  for rep in range(3):
      evs = []
      def mark(name):
          e = torch.cuda.Event(enable_timing=True); e.record(); evs.append((name, e))
      mark("start")
      idx.local_topk(Qn, k, mark=mark)
      torch.cuda.synchronize()
      tot = collections.OrderedDict()
      for (n0, e0), (n1, e1) in zip(evs[:-1], evs[1:]):
          tot[n1] = tot.get(n1, 0.0) + e0.elapsed_time(e1)
      print({k_: round(v, 3) for k_, v in tot.items()}, "total", round(sum(tot.values()), 3), flush=True)
