#!/usr/bin/env python3
"""How far does the host run ahead of the device in the bench step?  Times the HOST side of each step_lleqa() call (no synchronisation in
between) against the device time of the same steps: a host time per step close to the device's means something in the step blocks on
the stream (a pageable-memory upload, a .item()).  Usage (GPU box): python tools/diag_launch_ahead.py [--token-ids]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

sys.argv = [sys.argv[0]] + [a for a in sys.argv[1:]]
args = bench.parse()
dev = torch.device("cuda", 0)
st = bench.build_lleqa(args, dev, 0)
for _ in range(3):
    bench.step_lleqa(st)
torch.cuda.synchronize()
host = []
t_all = time.perf_counter()
for _ in range(8):
    t0 = time.perf_counter()
    bench.step_lleqa(st)
    host.append((time.perf_counter() - t0) * 1e3)
t_launch = (time.perf_counter() - t_all) * 1e3
torch.cuda.synchronize()
t_total = (time.perf_counter() - t_all) * 1e3
print("host ms per step call:", [round(h, 2) for h in host])
print(f"all 8 launched after {t_launch:.1f} ms, device done after {t_total:.1f} ms ({t_total / 8:.2f} ms per step)")
