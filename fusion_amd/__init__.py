"""fusion_amd -- MI355X-native scoring + fusion engine behind the `src/retrievers/hybrid.py`
surface of maastrichtlawtech/fusion (Ranker / Aggregator / run_hybrid.sh).

Layout (only what the encode -> score -> fuse hot path needs):
  csrc/                 hand-written HIP kernels for gfx950 + the C ABI (include/fusion_hip.h)
  _lib.py               ctypes binding of libfusion_hip.so (raises if the library is missing)
  ops.py                torch-tensor wrappers: device pointers + current HIP stream -> C ABI
  planes.py             device-resident ranked lists (dense planes by corpus position)
  retrievers/hybrid.py  Ranker, Aggregator, run_evaluation, main()  (mirror of the reference module)
  retrievers/bm25.py    BM25 (host index build, device scoring)
  utils/metrics.py      Metrics
  encoders.py           PyTorch-ROCm transformer forwards (DPR mean-pool, SPLADE, ColBERT)
  distributed.py        corpus-sharded top-k with one RCCL all-gather
"""
__version__ = "0.1.0"
