import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import bench
from fusion_amd import encoders
encoders.enable_gemm_tuning(); torch.cuda.tunable.tuning_enable(False)
dev = torch.device("cuda")
enc = {k: encoders.random_init(k, device=dev, size="base", seed=i) for i, k in enumerate(("dpr", "splade", "colbert"))}
rng = np.random.default_rng(31)
Q = 1024
qids, _, qlen = bench.synth_query_tokens(rng, Q, 32005, 1)
qd = torch.from_numpy(qids).to(dev)
cq = torch.from_numpy(np.where(np.arange(64)[None, :] < qlen[:, None], qids, encoders.MASK_TOKEN_ID)).to(dev)
def serial():
    a = enc["dpr"].encode_ids_packed(qd, qlen); b = enc["splade"].encode_ids_packed(qd, qlen); c = enc["colbert"].encode_query_ids(cq)
streams = [torch.cuda.Stream() for _ in range(3)]
def parallel():
    cur = torch.cuda.current_stream()
    for s in streams: s.wait_stream(cur)
    with torch.cuda.stream(streams[0]): a = enc["dpr"].encode_ids_packed(qd, qlen)
    with torch.cuda.stream(streams[1]): b = enc["splade"].encode_ids_packed(qd, qlen)
    with torch.cuda.stream(streams[2]): c = enc["colbert"].encode_query_ids(cq)
    for s in streams: cur.wait_stream(s)
for name, f in (("serial", serial), ("parallel", parallel), ("serial", serial), ("parallel", parallel)):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): f()
    torch.cuda.synchronize(); print(name, round((time.perf_counter() - t0) / 3 * 1e3, 2), "ms", flush=True)
