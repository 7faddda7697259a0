import os, ctypes as C, torch
L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libread_bw.so"))
L.read_bw.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p]
Q, ld = 1024, 27968
planes = [torch.randn((Q, ld), device="cuda") for _ in range(4)]
sink = torch.zeros(1 << 20, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for S in (1, 2, 4):
    for mode in (0, 1):
        n4 = Q * ld // 4
        f = lambda: L.read_bw(*[C.c_void_p(p.data_ptr()) for p in planes], S, n4, mode, C.c_void_p(sink.data_ptr()), st)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print({"planes": S, "mode": ["flat", "stride x4"][mode], "MB": round(S * Q * ld * 4 / 1e6, 1), "ms": round(ms, 4), "read GB/s": round(S * Q * ld * 4 / ms / 1e6, 1)})
