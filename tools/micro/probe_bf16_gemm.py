"""Probe: does torch on ROCm expose bf16 x bf16 -> fp32 GEMMs (out_dtype) and how fast are they at the encoder's shapes?"""
import torch, time
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
M = 36176
for (N, K) in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    a = torch.randn((M, 3 * K), device="cuda").bfloat16(); b = torch.randn((3 * K, N), device="cuda").bfloat16()
    c = torch.zeros((M, N), device="cuda")
    for name, f in (("mm out_dtype", lambda: torch.mm(a, b, out_dtype=torch.float32)),
                    ("addmm out_dtype", lambda: torch.addmm(c, a, b, out_dtype=torch.float32)),
                    ("mm bf16 out", lambda: torch.mm(a, b))):
        try:
            t = timeit(f)
            r = f()
            print(f"N={N} K'={3*K}: {name}: {t:.3f} ms  {2.0*M*N*3*K/t/1e9:.0f} TF  out {r.dtype}", flush=True)
        except Exception as e:
            print(f"N={N} K'={3*K}: {name}: FAILED {type(e).__name__}: {str(e)[:150]}", flush=True)
