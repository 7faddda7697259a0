import os, ctypes as C, torch
L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libstream_qkv.so"))
for T in (36176, 65536, 16384):
    qkv = torch.randn((T, 2304), device="cuda"); out = torch.empty((T, 768), device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda: L.stream_qkv(C.c_void_p(qkv.data_ptr()), T, 768, C.c_void_p(out.data_ptr()), st)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print({"T": T, "ms": round(ms, 4), "GB/s": round(T * 3072 * 4 / ms / 1e6, 1)})
