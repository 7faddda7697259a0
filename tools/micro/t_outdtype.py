import torch, time
x = torch.randn(65536, 768, device="cuda"); w = torch.randn(2304, 768, device="cuda"); b = torch.randn(2304, device="cuda")
x16, w16, b16 = x.half(), w.half(), b.half()
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("linear f16", t(lambda: torch.nn.functional.linear(x16, w16, b16)))
print("linear f16 + float", t(lambda: torch.nn.functional.linear(x16, w16, b16).float()))
print("x.half", t(lambda: x.half()))
try:
    y = torch.mm(x16, w16.t(), out_dtype=torch.float32); print("mm out_dtype ok", y.dtype, t(lambda: torch.mm(x16, w16.t(), out_dtype=torch.float32)))
except Exception as e: print("mm out_dtype:", repr(e)[:200])
try:
    y = torch.addmm(b, x16, w16.t(), out_dtype=torch.float32); print("addmm out_dtype ok", y.dtype, t(lambda: torch.addmm(b, x16, w16.t(), out_dtype=torch.float32)))
except Exception as e: print("addmm out_dtype (f32 bias):", repr(e)[:200])
try:
    y = torch.addmm(b16, x16, w16.t(), out_dtype=torch.float32); print("addmm out_dtype f16 bias ok", y.dtype, t(lambda: torch.addmm(b16, x16, w16.t(), out_dtype=torch.float32)))
    ref = torch.nn.functional.linear(x, w, b); print("err", (y - ref).abs().max().item(), ref.abs().max().item())
except Exception as e: print("addmm out_dtype (f16 bias):", repr(e)[:200])
print("linear f32", t(lambda: torch.nn.functional.linear(x, w, b), 3))
