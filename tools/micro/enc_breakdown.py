import sys, numpy as np, torch
sys.path.insert(0, ".")
import bench
from fusion_amd import encoders
torch.cuda.tunable.enable(False)
enc = encoders.random_init("colbert", device="cuda", size="base", seed=2)
rng = np.random.default_rng(0)
n = 220
L = np.clip(rng.normal(300, 120, n), 16, 512).astype(np.int64)
ids = torch.from_numpy(rng.integers(7, 32000, size=(n, 512))).cuda()
fwd = enc._packed_forward(enc.backbone)
for amp in (True, False):
    enc.amp = amp; enc._packed_forward(enc.backbone)
    for _ in range(2): fwd.hidden(ids, L)
    ev = bench.Events(); ev.mark("start")
    fwd.hidden(ids, L, ev.mark)
    torch.cuda.synchronize()
    st, _ = ev.durations_ms()
    print("amp", amp, "tokens", int(L.sum()), {k: round(v, 3) for k, v in st.items()}, "total", round(sum(st.values()), 2))
