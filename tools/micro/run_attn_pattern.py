import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from fusion_amd import ops
L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libattn_pattern.so"))
for mode in ("queries", "fixed:64", "fixed:16"):
    rng = np.random.default_rng(0)
    lens = np.full(1024, int(mode[6:]), dtype=np.int64) if mode.startswith("fixed") else np.clip(rng.normal(36, 14, 1024).round().astype(np.int64), 4, 64)
    T = int(lens.sum())
    qkv = torch.randn((T, 2304), device="cuda"); out = torch.empty((T, 768), device="cuda")
    strips, cu = ops.attn_strips(lens); sd = torch.from_numpy(strips).cuda()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda: L.attn_pattern(C.c_void_p(qkv.data_ptr()), 2304, C.c_void_p(sd.data_ptr()), len(strips), 12, C.c_void_p(out.data_ptr()), 768, st)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(mode, {"T": T, "ms": round(ms, 4), "GB/s(algorithmic)": round(T * 3072 * 4 / ms / 1e6, 1)})
