#!/usr/bin/env python3
"""bench.py -- queries/sec end-to-end (encode + score + fuse) on synthetic LLeQA-shaped batches.

N = 1 (the headline line; `--workload lleqa`).  One "step" = one pass of the hot path over one batch of Q synthetic
queries.  The corpus side is resident in HBM; the queries enter as STRINGS, as model.encode(queries) takes them (hybrid.py:101-102):
    0. tokenize    Q French-like questions -> token ids on the host (32,005-piece BPE of camembert's layout, tokenizers' encode_batch on all
                   host cores) + upload -- step i + 1's batch on a host thread while the device runs step i (fusion_amd.tokenization.prefetch;
                   `--serial-tokenize` shows the step without the overlap, `--token-ids` the rounds 1-4 form that starts from resident ids)
    1. encode      query token ids -> CamemBERT-base-shaped encoder (random init, fp32; padding-free forward: PyTorch-ROCm
                   hipBLASLt Linears + this repo's HIP embedding / attention / GELU / LayerNorm / pooling kernels) -> mean pool
    2. dpr score   normalise + fp32-MFMA cos-sim GEMM against the resident corpus embeddings   [Q, N]
    3. dpr rank    stable descending row sort -> order + rank planes
    4. bm25 score  float64 BM25 of the same batch against a resident synthetic index            [Q, N]
    5. bm25 rank   stable descending row sort (float64 keys)
    6. fuse        reciprocal-rank fusion in float64 (hybrid.py:252)
    7. order       stable sort of the fused scores in first-insertion order -> final ranked lists [Q, N]
(the BM25+DPR RRF hybrid of BASELINE.json configs[0] run on the configs[1] scale: N = 27,942, d = 768, Q = 1024).
The corpus side (document embeddings, BM25 index) is built once, untimed: the corpus is static.
In the same run, after the timed region, the other BASELINE.json configs are timed live with HIP events and reported
under "configs_measured" (DPR GEMM, SPLADE-shaped GEMM, ColBERT MaxSim, 4-system nsf fusion with a 40 %-invalid ColBERT
plane, the 1771-vector weight sweep, one mMARCO 1/8 shard) with the same roofline arithmetic.

N > 1 (under torch.distributed.run, one rank per GPU; `--workload mmarco`, the default when WORLD_SIZE > 1): the
corpus-SHARDED config 5 that north_star names for 2/4/8 GPUs -- mMARCO-fr-shaped 8,841,823 x 768 fp32 corpus row-sharded
over the ranks, queries encoded data-parallel (all-gather of the [Q, 768] embeddings), local chunked fp32-MFMA GEMM ->
streaming top-1000, ONE RCCL all-gather of the per-shard [Q, k] lists + identical local merge.  Strong scaling (the
corpus is fixed).  `--workload lleqa` at N > 1 gives query-sharded replicas with no data-path collective instead.

Prints ONE compact JSON line on rank 0's stdout (headline + config + stages + roofline + cpu_baseline + parity + targets: a few kB,
size pinned by tests/test_bench_line_cpu.py); the full record (roofline_all, configs_measured ...) goes to bench_detail.json and stderr.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F32_PEAK_TF = 157.3       # v_mfma_f32_32x32x2_f32, dense
MFMA_F16_PEAK_TF = 2500.0      # v_mfma_f32_32x32x16_f16, dense
METRIC = "queries/sec end-to-end (encode+score+fuse), LLeQA test; recall@500 parity"
TRAFFIC_PROFILES = ("r06_hbm_traffic.json", "r05_hbm_traffic.json", "r04_hbm_traffic.json", "r03_hbm_traffic.json", "r02_hbm_traffic.json", "r01_hbm_traffic.json")
# the stages of the step that ARE the path's kernels (scoring, ranking, fusion behind the C ABI).  The bench line's
# `roofline` names the one of THESE that takes the most time per step; the encoder's kernels (HIP and vendor) stay in `roofline_all`.
PATH_STAGES = ("dpr_score", "dpr_rank", "bm25_score", "bm25_rank", "fuse_rrf", "final_order")


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--workload", default="auto", choices=["auto", "lleqa", "mmarco"],
                   help="auto: lleqa on one GPU, the corpus-sharded mmarco config when WORLD_SIZE > 1")
    p.add_argument("--queries", type=int, default=1024)
    p.add_argument("--corpus", type=int, default=27942)
    p.add_argument("--dim", type=int, default=768)
    p.add_argument("--no-encode", action="store_true", help="skip the transformer forward (score+fuse only; not the headline metric)")
    p.add_argument("--token-ids", action="store_true", help="start the step from token ids resident on the device (rounds 1-4) instead of from query strings")
    p.add_argument("--serial-tokenize", action="store_true", help="tokenise every step's batch on the main thread before launching it (no overlap with the device)")
    p.add_argument("--require-tokenizer", action="store_true",
                   help="fail instead of falling back to resident token ids when the `tokenizers` wheel (or the tokenizer file) is missing: the fallback "
                        "changes what the headline measures (config.input / config.tokenizer_fallback say so)")
    p.add_argument("--tokenizer-threads", type=int, default=None, help="workers of the tokenizer's rayon pool (RAYON_NUM_THREADS; default min(8, half the visible cores))")
    p.add_argument("--encoder-size", default="base", choices=["base", "tiny"])
    p.add_argument("--encode-buckets", type=int, default=8, help="length buckets for the query encoder (1 = pad everything to the batch maximum)")
    p.add_argument("--encode-mode", default="packed", choices=["packed", "fused", "hf"],
                   help="packed: padding-free token rows, HIP attention/LayerNorm/pooling kernels between the hipBLASLt GEMMs; fused: lean torch forward, "
                        "linears over all length buckets' tokens at once; hf: the HF module per length bucket")
    p.add_argument("--overlap-bm25", action="store_true", help="run the BM25 branch on a second stream next to the encoder (measured: no gain, the encoder saturates the GPU)")
    p.add_argument("--no-gemm-tuning", action="store_true",
                   help="leave the encoder's fp32 Linears to the library heuristics instead of PyTorch TunableOp (fusion_amd/tuned/gemm_gfx950.csv)")
    p.add_argument("--two-kernel-fuse", action="store_true", help="round 4's step: fz_fuse_rank_f64 then fz_sort_rows_desc_placed on its float64 plane (A/B against the fused sort)")
    p.add_argument("--bm25-per-posting-expression", action="store_true", help="rounds 1-5's BM25 scoring: the float64 expression per (query, posting) instead of the per-index posting-value table (A/B)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-configs", action="store_true", help="skip the configs_measured block (configs 2-5 timed live after the headline region)")
    p.add_argument("--no-one-gpu-reference", action="store_true",
                   help="sharded workload at N > 1: skip the same workload on rank 0 alone (whole corpus on one GPU), reported next to the N-GPU value")
    p.add_argument("--mmarco-docs", type=int, default=8841823)
    p.add_argument("--no-corpus-encode", action="store_true", help="sharded workload: skip the per-rank corpus-encode rate on mMARCO-shaped passages")
    p.add_argument("--encode-sample", type=int, default=16384, help="passages per rank in that measurement")
    p.add_argument("--topk", type=int, default=1000)
    p.add_argument("--rehearsal", action="store_true",
                   help="allow several ranks on one device (gloo only): rehearses the N > 1 control flow on a smaller box; the numbers mean nothing")
    return p.parse_args()


class Events:
    """HIP events on torch's current stream (every kernel of the step is launched on it: fusion_amd.ops passes
    torch.cuda.current_stream() to the C ABI).  Each mark closes the interval since the previous one."""

    def __init__(self):
        self.marks = []

    def mark(self, name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.marks.append((name, e))

    def durations_ms(self):
        """-> ({name: total ms}, {name: number of intervals})"""
        tot, cnt = {}, {}
        for (n0, e0), (n1, e1) in zip(self.marks[:-1], self.marks[1:]):
            tot[n1] = tot.get(n1, 0.0) + e0.elapsed_time(e1)
            cnt[n1] = cnt.get(n1, 0) + 1
        return tot, cnt


def log(msg):
    """Progress on stderr (stdout carries the one JSON line): a long run must not look hung."""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def timeit_ms(f, n=10, warm=2):
    """Average duration of f() over n back-to-back calls, HIP events on the current stream."""
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def roof(kernel, ms, work, bound, **extra):
    """One roofline record: algorithmic work per launch / measured launch duration against the guide's peak."""
    peak, unit, scale = {"hbm": (HBM_PEAK_GBS * 1e9, "GB/s", 1e9), "mfma_f32": (MFMA_F32_PEAK_TF * 1e12, "TFLOP/s", 1e12),
                         "mfma_f16": (MFMA_F16_PEAK_TF * 1e12, "TFLOP/s", 1e12)}[bound]
    ach = work / (ms * 1e-3) if ms > 0 else 0.0
    return dict(kernel=kernel, bound="hbm" if bound == "hbm" else "mfma", ms=ms, achieved=ach / scale, peak=peak / scale, unit=unit,
                frac=ach / peak, **extra)


def synth_bm25_index(N, rng):
    """Synthetic LLeQA-shaped corpus text statistics: Zipf vocabulary of 20k lemmas, ~150 tokens per article."""
    V = 20000
    p = 1.0 / np.arange(1, V + 1) ** 1.05
    p /= p.sum()
    lens = np.clip(rng.normal(150, 60, N), 16, 512).astype(np.int64)
    tok = rng.choice(V, size=int(lens.sum()), p=p)
    doc = np.repeat(np.arange(N), lens)
    return V, lens, tok, doc, p


def synth_query_tokens(rng, Q, vocab_size, pad_id, L=64):
    """LLeQA questions: 8-64 word pieces."""
    qlen = rng.integers(8, L + 1, Q)
    ids = rng.integers(7, vocab_size - 1, (Q, L))
    mask = (np.arange(L)[None, :] < qlen[:, None]).astype(np.int64)
    return np.where(mask == 1, ids, pad_id), mask, qlen


def splade_like(rng, rows, V, nnz_mean, dev):
    """rows x V float32 SPLADE-shaped vectors (columns padded to a multiple of 4): per row ~nnz_mean distinct Zipf-distributed terms with
    log1p(relu(.)) weights -- a few hundred non-zeros of 32,005 per document is what splade/splade.py:88-99 yields after training."""
    p = 1.0 / np.arange(1, V + 1) ** 0.9; p /= p.sum()
    X = torch.zeros((rows, -(-V // 4) * 4), device=dev)
    k = np.maximum(1, rng.poisson(nnz_mean, rows))
    r = np.repeat(np.arange(rows), k)
    c = rng.choice(V, size=int(k.sum()), p=p)
    w = np.log1p(np.maximum(rng.normal(1.0, 1.0, int(k.sum())), 0.05)).astype(np.float32)
    X[torch.from_numpy(r).to(dev), torch.from_numpy(c).to(dev)] = torch.from_numpy(w).to(dev)
    return X


def build_lleqa(args, dev, rank):
    from fusion_amd import encoders, ops
    rng = np.random.default_rng(1234 + rank)
    Q, N, d = args.queries, args.corpus, args.dim
    st = {}
    if not args.no_encode:
        enc = encoders.random_init("dpr", device=dev, size=args.encoder_size, seed=0)
        if not args.no_gemm_tuning and args.encode_mode == "packed":
            encoders.enable_gemm_tuning()      # TunableOp picks the hipBLASLt / rocBLAS solution per Linear shape (same fp32 arithmetic)
        d = enc.dim
        tok = None
        if not args.token_ids:
            try:
                from fusion_amd.tokenization import SynthFrenchTokenizer, cap_host_threads
                # the program's entry point sizes the tokenizer's (process-global) rayon pool, once, before its first use: left at every visible
                # core it crowds out the thread that launches the GPU work (tokenization.cap_host_threads)
                st["tokenizer_threads"] = cap_host_threads(args.tokenizer_threads, override=args.tokenizer_threads is not None)
                tok = SynthFrenchTokenizer()
                assert tok.vocab_size == enc.backbone.config.vocab_size
            except Exception as ex:   # (a box without the `tokenizers` wheel: measure the rest rather than nothing -- and say so in `config`)
                if args.require_tokenizer:
                    raise
                log(f"WARNING: no tokenizer ({type(ex).__name__}: {ex}): the step starts from RESIDENT TOKEN IDS -- not the headline's definition "
                    "(config.tokenizer_fallback = true; --require-tokenizer makes this an error)")
                tok = None
                st["tokenizer_fallback"] = f"{type(ex).__name__}: {ex}"[:120]
        if tok is None:         # rounds 1-4: the step starts from token ids already on the device
            ids, mask, qlen = synth_query_tokens(rng, Q, enc.backbone.config.vocab_size, enc.backbone.config.pad_token_id)
        else:                   # the step starts from STRINGS, as model.encode(queries) does (hybrid.py:101-102): synthetic French-like
            #                     questions of 5-49 words (8-64 pieces: the token budget of the rounds before), tokenised on the host
            from fusion_amd.synth_text import FrenchLike
            st["texts"] = FrenchLike().sentences(rng, Q, 5, 49, question=True)
            st["tok"] = tok
            ids, qlen = st["tok"].encode_np(st["texts"], 64, pad_to_max=True)
            mask = (np.arange(64)[None, :] < qlen[:, None]).astype(np.int64)
            st["pinned"] = [torch.empty((Q, 64), dtype=torch.int64).pin_memory() for _ in range(2)]
            st["pinned_free"] = [None, None]          # the event after which a pinned buffer may be written again
        st["enc"], st["ids"], st["mask"] = enc, torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
        st["qlen"] = qlen   # host token counts (a tokenizer returns them): drives length-bucketed batching
    else:
        st["q_emb"] = torch.from_numpy(rng.normal(0, 1, (Q, d)).astype(np.float32)).to(dev)
    # corpus embeddings: encoded + normalised once (static corpus), resident in HBM
    g = torch.Generator(device=dev).manual_seed(7)
    De = torch.randn((N, d), generator=g, device=dev, dtype=torch.float32)
    st["Dn"] = ops.normalize_rows(De)
    del De
    # BM25 index (device CSR) + the batch's query terms
    V, lens, tok, doc, p = synth_bm25_index(N, np.random.default_rng(99))
    key = tok.astype(np.int64) * N + doc
    uniq, tf = np.unique(key, return_counts=True)
    pt, pd = uniq // N, uniq % N
    df = np.bincount(pt, minlength=V)
    idf = np.log10((N - df + 0.5) / (df + 0.5))
    toff = np.zeros(V + 1, dtype=np.int64); np.cumsum(df, out=toff[1:])
    qn = rng.integers(4, 16, Q)
    qterms = rng.choice(V, size=int(qn.sum()), p=p).astype(np.int32)
    qoff = np.zeros(Q + 1, dtype=np.int64); np.cumsum(qn, out=qoff[1:])
    st["bm25"] = dict(toff=torch.from_numpy(toff).to(dev), pdoc=torch.from_numpy(pd.astype(np.int32)).to(dev),
                      ptf=torch.from_numpy(tf.astype(np.int32)).to(dev), idf=torch.from_numpy(idf).to(dev),
                      doc_len=torch.from_numpy(lens.astype(np.int32)).to(dev), avgdl=float(lens.mean()),
                      qoff=torch.from_numpy(qoff).to(dev), qterms=torch.from_numpy(qterms).to(dev))
    st["bm25"]["doc_norm"] = ops.bm25_doc_norms(st["bm25"]["doc_len"], st["bm25"]["avgdl"], 2.5, 0.2)   # per index, like the idf table
    st["bm25"]["slice_off"] = ops.bm25_slice_offsets(st["bm25"]["toff"], st["bm25"]["pdoc"], N)         # ditto
    # ditto: every posting's whole BM25 term for this index and (k1, b) -- what BM25._doc_norm keeps next to the norms (round 6)
    st["bm25"]["pval"] = ops.bm25_posting_values(st["bm25"]["toff"], st["bm25"]["pdoc"], st["bm25"]["ptf"], st["bm25"]["idf"], st["bm25"]["doc_norm"], 2.5)
    st["lens2"] = torch.full((2, Q), N, dtype=torch.int32, device=dev)
    st["Q"], st["N"], st["d"] = Q, N, d
    st["buckets"] = args.encode_buckets
    st["encode_mode"] = args.encode_mode
    st["overlap"] = args.overlap_bm25
    st["two_kernel_fuse"] = args.two_kernel_fuse
    st["bm25_expr"] = args.bm25_per_posting_expression
    st["host"] = dict(idf=idf, toff=toff, pd=pd, tf=tf, lens=lens, qoff=qoff, qterms=qterms)
    # which instantiation of the ranking sort BM25.search_device would ask for: its host-side estimate of the rows' share of exact zeros
    from fusion_amd.retrievers.bm25 import LEXICAL_MIN_ZERO_SHARE, expected_zero_share
    st["bm25_zero_share_estimate"] = expected_zero_share(df, N, [qterms[qoff[q]:qoff[q + 1]].tolist() for q in range(Q)])
    st["bm25_lexical"] = st["bm25_zero_share_estimate"] >= LEXICAL_MIN_ZERO_SHARE
    st["bm25_postings"] = int(df[qterms].sum())     # postings the batch's query terms touch (terms repeat: bm25.py:152 does not de-duplicate)
    return st


def tokenized_batches(st, n):
    """n tokenised copies of the step's batch of query strings, one per step: text -> ids on the host (tokenizers' encode_batch: all host
    cores, GIL released) into one of two pinned buffers.  Consumed through tokenization.prefetch, batch i + 1 is produced on a host thread
    while the device runs batch i."""
    for i in range(n):
        buf, ev = st["pinned"][i & 1], st["pinned_free"][i & 1]
        if ev is not None:
            ev.synchronize()                 # the upload that read this buffer two steps ago has run
        ids, qlen = st["tok"].encode_np(st["texts"], 64, pad_to_max=True)
        buf.copy_(torch.from_numpy(ids))
        yield i & 1, qlen


def upload_ids(st, which, qlen):
    st["ids"].copy_(st["pinned"][which], non_blocking=True)
    e = torch.cuda.Event(); e.record()
    st["pinned_free"][which] = e
    st["qlen"] = qlen


def open_query_stream(st, n, serial=False):
    """The stream of n tokenised query batches the next run_steps() calls consume.  ONE stream feeds the warm-up and the timed steps, so
    the timed region starts with the pipeline in steady state (batch i + 1 on the host thread while the device runs batch i) and holds
    exactly one tokenisation per step: the caller asks for one batch more than it consumes (the last step's look-ahead, unused)."""
    from fusion_amd.tokenization import prefetch
    close_query_stream(st)
    if "texts" in st:
        src = tokenized_batches(st, n)
        st["stream"], st["stream_serial"] = iter(src if serial else prefetch(src)), serial


def close_query_stream(st):
    s = st.pop("stream", None)
    if s is not None:
        s.close()                            # (a generator: ends the worker thread)


def run_steps(st, n):
    """n passes of the hot path; with query strings each pass takes its batch from the open stream (tokenised one step ahead on a host
    thread, or -- serial -- right here, after the device has finished the previous pass)."""
    out = None
    for _ in range(n):
        if "stream" in st:
            which, qlen = next(st["stream"])
            upload_ids(st, which, qlen)
        out = step_lleqa(st)
        if st.get("stream_serial"):
            torch.cuda.synchronize()
    return out


def step_lleqa(st, ev=None):
    """One pass of the hot path on torch's current stream.  With --overlap-bm25 the BM25 branch (score + rank), which
    does not depend on the encoder, runs on a second HIP stream next to it."""
    from fusion_amd import ops
    Q, N = st["Q"], st["N"]
    b = st["bm25"]

    def bm25_branch():
        B = ops.bm25_scores(b["toff"], b["pdoc"], b["ptf"], b["idf"], b["doc_len"], b["avgdl"], 2.5, 0.2, b["qoff"], b["qterms"], Q, N,
                            doc_norm=b["doc_norm"], slice_off=b["slice_off"], pval=None if st.get("bm25_expr") else b["pval"])
        if ev: ev.mark("bm25_score")
        o_b, _, r_b = ops.sort_rows_desc(B, want_keys=False, want_rank=True, lexical=st["bm25_lexical"])   # (as BM25.search_device: the zero-compacting instantiation when the rows are expected to be mostly zeros)
        if ev: ev.mark("bm25_rank")
        return B, o_b, r_b

    if ev: ev.mark("start")
    side = None
    if ev is None and st.get("overlap", False):
        side = st.setdefault("side_stream", torch.cuda.Stream())
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            B, o_b, r_b = bm25_branch()
    if "enc" not in st:
        q_emb = st["q_emb"]
    elif st["encode_mode"] == "packed":
        q_emb = st["enc"].encode_ids_packed(st["ids"], st["qlen"], mark=ev.mark if ev else None)
    elif st["encode_mode"] == "fused":
        q_emb = st["enc"].encode_ids_fused(st["ids"], st["qlen"], st["buckets"])
    else:
        q_emb = st["enc"].encode_ids_bucketed(st["ids"], st["mask"], st["qlen"], st["buckets"])
    if ev: ev.mark("encode_pool")
    Qn = ops.normalize_rows(q_emb)
    S = ops.dot_scores(Qn, st["Dn"])
    if ev: ev.mark("dpr_score")
    o_d, _, r_d = ops.sort_rows_desc(S, want_keys=False, want_rank=True)
    if ev: ev.mark("dpr_rank")
    if side is None:
        B, o_b, r_b = bm25_branch()
    else:
        torch.cuda.current_stream().wait_stream(side)
        for t in (B, o_b, r_b):
            t.record_stream(torch.cuda.current_stream())
    if st.get("two_kernel_fuse"):   # round 4's form (--two-kernel-fuse): the float64 fused plane written by one kernel and sorted by the next
        fused = ops.fuse_rank([r_b, r_d], st["lens2"], "rrf")
        if ev: ev.mark("fuse_rrf")
        order, scores, _ = ops.sort_rows_desc(fused, init_rank=r_b, covers_all=True)
    else:   # as Aggregator.fuse_device calls it: RRF formed per key in the final sort's load phase; ties keep BM25's (system 0) order; full lists
        order, scores, _ = ops.sort_rank_fused([r_b, r_d], st["lens2"], "rrf", init_rank=r_b, covers_all=True)
    if ev: ev.mark("final_order")
    return order, scores, (S, B, q_emb)


def algorithmic_work(st):
    """Algorithmic bytes / flops PER LAUNCH of every kernel of the step (SURVEY.md 8d figures x the units one launch
    processes).  hand=False marks vendor kernels (hipBLASLt): reported, never the "dominant hand-written kernel"."""
    Q, N, d = st["Q"], st["N"], st["d"]
    e = Q * N
    w = {}
    if "enc" in st and st.get("encode_mode") == "packed":
        cfg = st["enc"].backbone.config
        h, ff = cfg.hidden_size, cfg.intermediate_size
        T = int(np.minimum(np.asarray(st["qlen"]), st["ids"].shape[1]).sum())          # real token rows
        Tp = -(-T // 512) * 512 if torch.cuda.tunable.is_enabled() else T               # rows the row-wise kernels and GEMMs see
        # per layer: QKV (3h), out (h), FFN1 (ff), FFN2 (ff) -> 4 GEMM launches averaging this many flops each
        w["encode_gemm"] = dict(kernel="hipBLASLt Cijk_* (4 Linears / layer, vendor)", bound="mfma_f32", hand=False,
                                work=2.0 * Tp * h * (3 * h + h + 2 * ff) / 4)
        w["encode_attn"] = dict(kernel="attn_varlen_kernel", bound="hbm", work=T * 4 * h * 4)   # fused-QKV rows in + context rows out
        w["encode_ln"] = dict(kernel="add_layernorm_kernel", bound="hbm", work=3 * Tp * h * 4)  # y + residual in, x out
        w["encode_gelu"] = dict(kernel="gelu_kernel", bound="hbm", work=2 * Tp * ff * 4)        # in place: read + write
    w.update({
        "dpr_score": dict(kernel="dot_scores_kernel (+ normalize_rows)", bound="mfma_f32", work=2.0 * Q * N * d),
        "dpr_rank": dict(kernel="sort_rows_kernel (f32 keys)", bound="hbm", work=e * (4 + 4 + 4)),
        # BM25: every touched posting read once (doc id + tf, 8 B) + the fp64 score plane written once
        # (round 6: doc id + the posting's tabulated float64 term, 12 B; rounds 1-5: doc id + tf, 8 B)
        "bm25_score": dict(kernel="bm25_kernel", bound="hbm", work=st.get("bm25_postings", 0) * (8 if st.get("bm25_expr") else 12) + e * 8),
        "bm25_rank": dict(kernel="sort_rows_kernel (f64 keys)", bound="hbm", work=e * (8 + 4 + 4)),
    })
    if st.get("two_kernel_fuse"):
        w["fuse_rrf"] = dict(kernel="fuse_rank_kernel", bound="hbm", work=e * (2 * 4 + 8))
        w["final_order"] = dict(kernel="sort_rows_kernel (f64 keys, placed)", bound="hbm", work=e * (8 + 4 + 4 + 8))
    else:   # two rank planes in, order + fused float64 scores out: the float64 plane between fusion and sort does not exist
        w["final_order"] = dict(kernel="sort_rows_kernel<FUSE> (rrf formed on load, f64 keys, placed)", bound="hbm", work=e * (2 * 4 + 4 + 8))
    return w


def profiled_traffic(stage, st):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (profiles/r0X_hbm_traffic.json;
    FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md).  NOT measured in this run: valid only for the
    shape it was collected on, otherwise None.  Returns (bytes, source file)."""
    if (st["Q"], st["N"], st["d"]) != (1024, 27942, 768):
        return None, None
    pats = {"dpr_score": "dot_scores_kernel", "dpr_rank": "sort_rows_kernel<1024, 28, 1,", "bm25_rank": "sort_rows_kernel<1024, 28, 2, false, 1>",
            "final_order": "sort_rows_kernel<1024, 28, 2, false, 2>", "fuse_rrf": "fuse_rank_kernel", "bm25_score": "bm25_kernel", "encode_attn": "attn_varlen_kernel",
            "encode_gelu": "gelu_kernel", "encode_ln": "add_layernorm_kernel"}
    for name in TRAFFIC_PROFILES:
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", name)))
        except OSError:
            continue
        pat = pats.get(stage, "\0")
        if stage == "bm25_rank" and st.get("bm25_lexical"):
            pat = "sort_rows_kernel<1024, 28, 2, false, 3>"      # (the zero-compacting instantiation)
        for k, v in t.items():
            if k != "_note" and pat in k:
                return v.get("hbm_bytes_corrected"), f"profiles/{name}"
    return None, None


def cpu_baseline_lleqa(st, q_emb_dev, budget_s=20.0):
    """The CPU oracle ("port") on a bounded sample of the SAME batch: score + rank + fuse + order for the first q
    queries, all host cores via OpenMP, fed with the embeddings the DEVICE scored (so the parity figures below compare
    like with like); plus the encode leg on the host (the HF module on torch's CPU threads)."""
    import ctypes as C
    from oracle import oracle
    oracle.build()
    N, d = st["N"], st["d"]
    h = st["host"]
    qs = 8
    Dn = st["Dn"].cpu().numpy()[:, :d]
    q_emb = q_emb_dev[:64].cpu().numpy()

    def run(q):
        t0 = time.perf_counter()
        Qn = oracle.normalize_rows(q_emb[:q])
        S = oracle.dot_scores(Qn, Dn, fma_chain=True)
        _, _, r_d = oracle.sort_rows_desc(S, want_rank=True)
        B = np.empty((q, N), dtype=np.float64)
        lib = oracle.lib()
        qoff = np.ascontiguousarray(h["qoff"][: q + 1]); qterms = np.ascontiguousarray(h["qterms"][: int(qoff[-1]) + 1])
        oracle._chk(lib.fzo_bm25_scores_f64(oracle._p(h["toff"]), oracle._p(h["pd"].astype(np.int32)), oracle._p(h["tf"].astype(np.int32)),
                                            oracle._p(np.ascontiguousarray(h["idf"])), oracle._p(h["lens"].astype(np.int32)),
                                            C.c_double(float(h["lens"].mean())), C.c_double(2.5), C.c_double(0.2), oracle._p(qoff),
                                            oracle._p(qterms), q, N, oracle._p(B), N), "bm25")
        o_b, _, r_b = oracle.sort_rows_desc(B, want_rank=True)
        f = oracle.fuse_rank([r_b, r_d], np.full((2, q), N, dtype=np.int32), "rrf")
        o_f, k_f = oracle.sort_rows_desc(f, init_order=o_b)
        return time.perf_counter() - t0, (S, B, o_f, k_f)
    t, _ = run(qs)
    q = int(min(64, max(qs, qs * min(8.0, (budget_s / 2) / max(t, 1e-3)))))
    t, out = run(q)
    cores = oracle.num_threads()
    oracle.set_threads(1)                      # single-thread figure next to the all-core one (4 queries: a few seconds)
    try:
        t1, _ = run(4)
    finally:
        oracle.set_threads(cores)
    res = dict(value=q / t, unit="queries/s", cores=cores, kind="port",
               sample=f"first {q} queries of the batch, score+rank+fuse+order only (no encoder forward), N={N}, d={d}, OpenMP",
               score_fuse_only_value=q / t, score_fuse_only_single_thread_value=4 / t1, score_fuse_s=t)
    if "enc" in st:
        # the encode leg on the host cores too: the same HF module the reference runs (SentenceTransformer.encode on CPU,
        # hybrid.py:97-102), fp32, torch's CPU threads; sorted by length into sub-batches of 16 as encode() does
        import copy
        cpu_model = copy.deepcopy(st["enc"].backbone).to("cpu").eval()
        ids_c, mask_c = st["ids"][:q].cpu(), st["mask"][:q].cpu()
        order = torch.argsort(mask_c.sum(1), descending=True)
        t0 = time.perf_counter()
        with torch.no_grad():
            for s0 in range(0, q, 16):
                sel = order[s0: s0 + 16]
                Lb = int(mask_c[sel].sum(1).max())
                hcpu = cpu_model(input_ids=ids_c[sel][:, :Lb], attention_mask=mask_c[sel][:, :Lb]).last_hidden_state
                _ = (hcpu * mask_c[sel][:, :Lb].unsqueeze(-1)).sum(1)
        te = time.perf_counter() - t0
        tt = 0.0
        if "texts" in st:    # the device step starts from strings: so does the host's (the same tokenizer; a millisecond or two for 64 queries)
            t0 = time.perf_counter(); st["tok"].encode_np(st["texts"][:q], 64, pad_to_max=True); tt = time.perf_counter() - t0
        res.update(value=q / (t + te + tt), encode_s=te, tokenize_s=tt, cores=max(cores, torch.get_num_threads()),
                   sample=f"first {q} queries of the batch END TO END on the host: HF fp32 forward on torch's CPU threads ({te:.2f} s) + oracle "
                          f"score+rank+fuse+order with OpenMP ({t:.2f} s), N={N}, d={d}")
        del cpu_model
    return res, out, q


# ---------------------------------------------------------------------------------------------------------------------
# configs_measured: BASELINE.json configs 2-5 at their quoted sizes, timed live with HIP events, same roofline arithmetic
# ---------------------------------------------------------------------------------------------------------------------
def rand_plane(ops, Q, N, g, scale=1.0, shift=0.0):
    p = ops.alloc_plane(Q, N, torch.float32, "cuda")
    p.copy_(torch.randn((Q, N), generator=g, device="cuda") * scale + shift)
    return p


def measure_configs(dev, N=27942):
    from fusion_amd import ops
    from fusion_amd.planes import RankedSystem
    from fusion_amd.retrievers.hybrid import Aggregator, _rank_scores, weight_grid
    out = []
    g = torch.Generator(device=dev).manual_seed(11)

    # -- config 2: DPR bi-encoder dot-product scoring, full LLeQA corpus, Q in {195, 1024} --------------------------
    d = 768
    Dn = ops.normalize_rows(torch.randn((N, d), generator=g, device=dev))
    Qn = ops.normalize_rows(torch.randn((1024, d), generator=g, device=dev))
    S = ops.alloc_plane(1024, N, torch.float32, dev)
    for Q in (1024, 195):
        Qs, So = Qn[:Q].contiguous(), S[:Q]
        ms = timeit_ms(lambda: ops.dot_scores(Qs, Dn, out=So), n=20)
        out.append(dict(config="2: DPR cos-sim scoring", shape=dict(Q=Q, N=N, d=d), **roof("dot_scores_kernel", ms, 2.0 * Q * N * d, "mfma_f32")))
        ms = timeit_ms(lambda: ops.sort_rows_desc(So, want_keys=False, want_rank=True), n=10)
        ops.sort_bucket_rank_rows(reset=True)
        ops.sort_rows_desc(So, want_keys=False, want_rank=True)
        br = ops.sort_bucket_rank_rows(reset=True)
        prev = os.environ.get("FZ_SORT_BUCKET_RANK")
        os.environ["FZ_SORT_BUCKET_RANK"] = "0"     # the same rows by the four digit passes (round 3's path; same permutation, tests/test_gpu_sort_bucket.py)
        try:
            ms_digits = timeit_ms(lambda: ops.sort_rows_desc(So, want_keys=False, want_rank=True), n=10)
        finally:
            if prev is None: os.environ.pop("FZ_SORT_BUCKET_RANK", None)
            else: os.environ["FZ_SORT_BUCKET_RANK"] = prev
        out.append(dict(config="2: DPR full ranking", shape=dict(Q=Q, N=N), **roof("sort_rows_kernel (f32 keys)", ms, Q * N * 12, "hbm"),
                        ms_digit_passes=ms_digits, rows_bucket_ranked=br[0], rows_pair_swapped_back=br[1], rows_handed_to_digit_passes=br[2]))
    del Dn, Qn

    # -- config 1's lexical side: the ranking sort of BM25-like float64 rows as a function of their share of EXACT zeros (documents that share
    #    no term with the query; the bench step's own synthetic index keeps its stop-word-like terms, so its rows hold ~2 % zeros) -- the
    #    zero-compacting instantiation (fz_sort_rows_desc_lexical, what BM25.search_device calls) against the plain one, same rows, same outputs
    Bz = ops.alloc_plane(1024, N, torch.float64, dev)
    for zf in (0.0, 0.4, 0.6, 0.8):
        x = torch.distributions.Gamma(0.8, 0.25).sample((1024, N)).to(dev).double() + 0.01
        x[torch.rand((1024, N), generator=g, device=dev) < zf] = 0.0
        Bz.copy_(x); del x
        # (the better of two runs of 10: one 40 ms stall inside a run -- seen once in a dozen bench runs, on no kernel in particular -- is 4 ms on its average)
        ms_lex = min(timeit_ms(lambda: ops.sort_rows_desc(Bz, want_keys=False, want_rank=True, lexical=True), n=10) for _ in range(2))
        ms_plain = min(timeit_ms(lambda: ops.sort_rows_desc(Bz, want_keys=False, want_rank=True), n=10) for _ in range(2))
        ops.sort_zero_compact_rows(reset=True)
        a_ = ops.sort_rows_desc(Bz, want_keys=False, want_rank=True, lexical=True)
        zc = ops.sort_zero_compact_rows(reset=True)
        b_ = ops.sort_rows_desc(Bz, want_keys=False, want_rank=True)
        out.append(dict(config="1: BM25 ranking sort vs the rows' share of exact zeros", shape=dict(Q=1024, N=N, zero_share=zf),
                        **roof("sort_rows_kernel<SORT_ROWS_ZC> (f64 keys, zero compaction)", ms_lex, 1024 * N * 16, "hbm"), ms_plain_instantiation=ms_plain,
                        rows_compacted=zc[0], rows_kept_whole=zc[1], outputs_equal=bool(torch.equal(a_[0], b_[0]) and torch.equal(a_[2], b_[2]))))
    del Bz

    log("configs: DPR done")
    # -- config 3: ColBERT MaxSim, Q = 195, L_q = 64, L_d ~ clip(N(300,120),16,512), dim 128, fp16 unit-norm tokens ---
    rng = np.random.default_rng(0)
    lens = np.clip(rng.normal(300, 120, N), 16, 512).astype(np.int64)
    off = np.zeros(N + 1, dtype=np.int64); off[1:] = np.cumsum(lens)
    sumL = int(off[-1])
    Dtok = torch.nn.functional.normalize(torch.randn((sumL, 128), generator=g, device=dev), dim=-1).half()
    Doff = torch.from_numpy(off).to(dev)
    for Q, n in ((195, 5), (1024, 2)):
        Qtok = torch.nn.functional.normalize(torch.randn((Q, 64, 128), generator=g, device=dev), dim=-1).half()
        So = S[:Q]
        ms = timeit_ms(lambda: ops.maxsim(Qtok, Dtok, Doff, out=So, max_doc_len=512), n=n, warm=1)
        out.append(dict(config="3: ColBERT MaxSim", shape=dict(Q=Q, N=N, Lq=64, sumL=sumL, dim=128),
                        **roof("maxsim_kernel", ms, 2.0 * Q * 64 * sumL * 128, "mfma_f16")))
    del Dtok, Qtok

    log("configs: MaxSim done")
    # -- config 3b: SPLADE-shaped dense scoring, V = 32,005 (the reference scores SPLADE vectors densely, hybrid.py:101-103)
    V, Vp, Q = 32005, 32008, 1024
    Ds = torch.zeros((N, Vp), device=dev)
    for c0 in range(0, N, 4096):
        c1 = min(N, c0 + 4096)
        Ds[c0:c1, :V] = torch.log1p(torch.relu(torch.randn((c1 - c0, V), generator=g, device=dev) - 1.0))
    Ds = ops.normalize_rows(Ds)
    Qs = torch.zeros((Q, Vp), device=dev); Qs[:, :V] = torch.log1p(torch.relu(torch.randn((Q, V), generator=g, device=dev) - 1.5))
    Qs = ops.normalize_rows(Qs)
    ms = timeit_ms(lambda: ops.dot_scores(Qs, Ds, out=S), n=3, warm=1)
    out.append(dict(config="3b: SPLADE dense cos-sim scoring", shape=dict(Q=Q, N=N, V=V), **roof("dot_scores_kernel", ms, 2.0 * Q * N * V, "mfma_f32")))
    del Ds, Qs

    log("configs: SPLADE scoring done")
    # -- config 4: 4-way nsf fusion, colbert plane 40 % invalid; min-max / z-score / percentile-rank; Q in {1024, 195} --
    names = ["bm25", "dpr", "splade", "colbert"]
    for Q in (1024, 195):
        planes = [rand_plane(ops, Q, N, g, s + 1.0, float(s)) for s in range(4)]
        # ranked the way Ranker hands systems on (hybrid._rank_scores): rank / order planes, and mean | std | min | max of every list as
        # by-products of the ranking sort; ColBERT as PLAID-style short lists (the last 40 % of every ranking absent)
        systems = {n: _rank_scores(p, np.arange(N), int(0.6 * N) if n == "colbert" else None) for n, p in zip(names, planes)}
        ranks = [None if s.full else s.rank for s in systems.values()]   # what Aggregator.fuse_device passes: validity of the partial list only
        w = [0.25] * 4
        fused = ops.alloc_plane(Q, N, torch.float32, dev)
        distr = [torch.quantile(p[:8].flatten()[:1000000].double(), torch.linspace(0, 1, 1001, device=dev, dtype=torch.float64)).float().contiguous()
                 for p in planes]
        # algorithmic bytes: S score planes in, fused plane out (+ the partial system's validity, 1 bit per document)
        work = (4 + 1) * Q * N * 4 + Q * N // 8
        orders = [s.order for s in systems.values()]
        vbits = [s.valid_bits() for s in systems.values()]     # the partial list's validity as a bitmap, built once per system
        lens4 = torch.stack([s.lens for s in systems.values()]).contiguous()
        for norm in ("min-max", "z-score", "percentile-rank"):
            # called the way Aggregator.fuse_device calls it: every system brings the statistics its ranking sort produced (the short ColBERT
            # lists: over their listed prefix), the partial list's validity is a bitmap -- the fusion is ONE flat pass, nothing else is launched
            def call():
                st = [s.stats(norm) for s in systems.values()] if norm != "percentile-rank" else None
                return ops.fuse_nsf(planes, ranks, w, norm, distr if norm == "percentile-rank" else None, out=fused, stats=st, valid_bits=vbits)
            ms = timeit_ms(call, n=10)
            out.append(dict(config=f"4: nsf {norm} fusion, S=4, colbert 40% absent" + (", P=1001" if norm == "percentile-rank" else ""), shape=dict(Q=Q, N=N, S=4),
                            **roof("fuse_nsf kernels", ms, work, "hbm")))
        # percentile-rank / NCE with the tables the reference READS (hybrid.py:412,451: the `_28k` table = len(corpus) + 1 quantiles per
        # system): one system's table LDS-resident at a time (csrc/tables.hip); tables prepared inside the call, as fuse_device does
        P28 = N + 1
        distr28 = []
        for p_ in planes:
            pool = torch.sort(p_[:64, :N].flatten().double()).values
            pos = torch.linspace(0, pool.numel() - 1, P28, device=dev, dtype=torch.float64)
            lo_ = pos.floor().long(); hi_ = torch.clamp(lo_ + 1, max=pool.numel() - 1)
            distr28.append((pool[lo_] + (pool[hi_] - pool[lo_]) * (pos - lo_)).float().contiguous())
        for norm in ("percentile-rank", "normal-curve-equivalent"):
            ms = timeit_ms(lambda: ops.fuse_nsf(planes, ranks, w, norm, distr28, out=fused, valid_bits=vbits), n=10)
            assert ops.last_tables_path == "lds-swap"
            row = dict(config=f"4: nsf {norm}, S=4, P={P28} (the table size hybrid.py:412,451 read), colbert 40% absent", shape=dict(Q=Q, N=N, S=4, P=P28),
                       **roof("fuse_nsf_bigtab_kernel", ms, work, "hbm"))
            if Q == 1024 and norm == "percentile-rank":   # what these sizes cost before round 4: fz_fuse_nsf_f32's global-memory search
                row["ms_round3_path"] = timeit_ms(lambda: ops.fuse_nsf(planes, ranks, w, norm, distr28, out=fused, valid_bits=vbits, tables=False), n=3)
            out.append(row)
        del distr28
        ms = timeit_ms(lambda: Aggregator.fuse_device(systems, "nsf", "min-max", dict(zip(names, w)), {}), n=5)
        out.append(dict(config="4: Aggregator.fuse_device nsf min-max END TO END (stats + fuse + insertion order + final sort)", shape=dict(Q=Q, N=N, S=4),
                        ms=ms, queries_per_s=Q / (ms * 1e-3)))
        # float64 fused rows over full lists (the BM25 + DPR RRF hybrid of the headline step): the full lists Aggregator.fuse returns vs the
        # top-1000 form main() reads (hybrid.py:537; rows selected, not sorted)
        two = {n: systems[n] for n in ("bm25", "dpr")}
        lens_two = torch.stack([s.lens for s in two.values()]).contiguous()
        ms = timeit_ms(lambda: ops.fuse_rank([s.rank for s in two.values()], lens_two, "rrf"), n=10)
        out.append(dict(config="1: rrf fusion alone (fuse_rank_kernel, S=2; the full-list path forms it inside the final sort since round 5, long rows and the top-k selection still launch it)",
                        shape=dict(Q=Q, N=N, S=2), **roof("fuse_rank_kernel", ms, Q * N * (2 * 4 + 8), "hbm")))
        ms_full = timeit_ms(lambda: Aggregator.fuse_device(two, "rrf", None, {}, {}), n=5)
        ms_top = timeit_ms(lambda: Aggregator.fuse_device(two, "rrf", None, {}, {}, topk=1000), n=5)
        out.append(dict(config="1: Aggregator.fuse_device rrf BM25+DPR, topk=1000 (round 5: the fused full sort, cut -- cheaper than fuse + fz_select_topk_f + two 2k-key sorts) vs the full lists",
                        shape=dict(Q=Q, N=N, S=2, k=1000), ms=ms_top, ms_full_lists=ms_full, queries_per_s=Q / (ms_top * 1e-3)))
        if Q == 195:   # the 1771-vector sweep of hybrid.py:404-426 on the LLeQA test split size
            grid = weight_grid(names)
            labels = [rng.choice(N, size=int(rng.integers(1, 6)), replace=False).tolist() for _ in range(Q)]
            for norm in ("min-max", "z-score"):
                Aggregator.tune(systems, norm, grid[:3], labels, {})
                torch.cuda.synchronize(); t0 = time.perf_counter()
                Aggregator.tune(systems, norm, grid, labels, {})
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
                out.append(dict(config=f"4: weight sweep nsf {norm}, {len(grid)} vectors (np.float64 lattice, float64 sweep)", shape=dict(Q=Q, N=N, S=4, W=len(grid)),
                                ms=dt * 1e3, ms_per_weight_vector=dt * 1e3 / len(grid), note="wall clock incl. host-side metrics"))
        if Q == 195:
            # the drop-in boundary with the REFERENCE'S OWN TYPES in and out (hybrid.py:66-75,170-220: list[Q] of list[<= N] of
            # {'corpus_id', 'score'} per system -> the fused lists in the same form): host packing (csrc/pyhost.c + numpy), upload, the
            # device fusion, download, rebuilding the dicts.  The first 32 queries of the batch (3.4 M dicts in, 0.9 M out per call); the
            # cost is per query.  Aggregator.fuse(..., as_device=True) callers -- main() -- pay the device part only.
            from fusion_amd.retrievers.hybrid import pack_ranked_lists
            Qb = 32
            sub = {n: RankedSystem(scores=s.scores[:Qb], order=s.order[:Qb], rank=s.rank[:Qb], lens=s.lens[:Qb], ids=np.arange(N) + 1, full=s.full)
                   for n, s in systems.items()}
            as_lists = {n: s.to_lists() for n, s in sub.items()}
            t0 = time.perf_counter(); pack_ranked_lists(as_lists); pack_ms = (time.perf_counter() - t0) * 1e3
            for method, norm, lw in (("rrf", None, None), ("nsf", "min-max", dict(zip(names, w)))):
                Aggregator.fuse(as_lists, method, norm, lw, {})                      # warm
                torch.cuda.synchronize(); t0 = time.perf_counter()
                res_lists = Aggregator.fuse(as_lists, method, norm, lw, {})
                total_ms = (time.perf_counter() - t0) * 1e3
                torch.cuda.synchronize(); t0 = time.perf_counter()
                fd = Aggregator.fuse(as_lists, method, norm, lw, {}, as_device=True)
                torch.cuda.synchronize(); to_dev_ms = (time.perf_counter() - t0) * 1e3   # pack + upload + device fusion
                t0 = time.perf_counter(); fd.to_lists(); unpack_ms = (time.perf_counter() - t0) * 1e3
                assert len(res_lists) == Qb and len(res_lists[0]) == N
                out.append(dict(config=f"boundary: Aggregator.fuse {method}{' ' + norm if norm else ''}, reference types in and out", shape=dict(Q=Qb, N=N, S=4),
                                ms_per_query=total_ms / Qb, queries_per_s=Qb / (total_ms * 1e-3), pack_ms_per_query=pack_ms / Qb,
                                pack_upload_device_ms_per_query=to_dev_ms / Qb, download_unpack_ms_per_query=unpack_ms / Qb,
                                note="host-bound by construction: 4 x 27,942 dicts in and 27,942 out per query; round 3: ~102 ms per query"))
            del as_lists, sub, res_lists, fd
        del planes, systems, ranks, fused

    log("configs: fusion + sweep done")
    # -- config 5 at one GPU: one 1/8 shard of mMARCO (what each GPU does at G = 8), no collective ---------------------
    from fusion_amd.distributed import ShardedDenseIndex
    Nl, Q, k = 8841823 // 8, 1024, 1000
    Dm = torch.empty((Nl, 768), dtype=torch.float32, device=dev)
    for c0 in range(0, Nl, 1 << 19):
        c1 = min(Nl, c0 + (1 << 19))
        Dm[c0:c1] = ops.normalize_rows(torch.randn((c1 - c0, 768), generator=g, device=dev))
    Qm = ops.normalize_rows(torch.randn((Q, 768), generator=g, device=dev))
    idx = ShardedDenseIndex(Dm, 0)
    ms = timeit_ms(lambda: idx.local_topk(Qm, k), n=3, warm=1)
    out.append(dict(config="5: mMARCO 1/8 shard, chunked GEMM with the top-1000 threshold filter as its epilogue + folds (no collective)", shape=dict(Q=Q, N=Nl, d=768, k=k),
                    **roof("dot_scores_kernel<filter epilogue> + topk kernels", ms, 2.0 * Q * Nl * 768, "mfma_f32")))
    return out



def mmarco_encode_rate(enc, dev, n_passages=16384, seed=17, shard_rows=8841823 // 8):
    """config 5's ENCODE leg (sentence_transformers.py:339-345: every 50,000-passage chunk is encoded before it is scored) at mMARCO's
    shape -- short passages, ~ clip(N(70, 30), 8, 256) word pieces: a different GEMM-row regime from the 296-token LLeQA articles -- through
    DenseEncoder.encode_ids_corpus (padding-free forward, sub-batches of 65,536 token rows).  Data-parallel: every GPU encodes its own
    shard, no collective.  Returns passages/s, tokens/s, the fraction of the fp32-MFMA peak and what a 1/8 shard of the corpus costs."""
    rng = np.random.default_rng(seed)
    cfg = enc.backbone.config
    lens = np.clip(rng.normal(70, 30, n_passages), 8, 256).astype(np.int64)
    ids = torch.from_numpy(rng.integers(7, cfg.vocab_size - 1, (n_passages, 256))).to(dev)
    T = int(lens.sum())
    h, ff, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    flops = 2.0 * T * L * (4 * h * h + 2 * h * ff) + 4.0 * L * h * float((lens.astype(np.float64) ** 2).sum())
    tuning_was_on = torch.cuda.tunable.is_enabled() and torch.cuda.tunable.tuning_is_enabled()
    if tuning_was_on:
        torch.cuda.tunable.tuning_enable(False)      # recorded solutions stay in use; new sub-batch shapes are not tuned on first use
    try:
        enc.encode_ids_corpus(ids, lens)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        enc.encode_ids_corpus(ids, lens)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    finally:
        if tuning_was_on:
            torch.cuda.tunable.tuning_enable(True)
    return dict(passages=n_passages, tokens=T, mean_len=float(lens.mean()), sub_batch_token_rows=int(enc.packed_tokens), ms=dt * 1e3,
                passages_per_s=n_passages / dt, tokens_per_s=T / dt, flops=flops, bound="mfma", achieved=flops / dt / 1e12, peak=MFMA_F32_PEAK_TF,
                unit="TFLOP/s", frac=flops / dt / (MFMA_F32_PEAK_TF * 1e12), shard_rows=shard_rows, full_shard_encode_s=shard_rows / (n_passages / dt),
                full_corpus_one_gpu_s=8841823 / (n_passages / dt))


# ---------------------------------------------------------------------------------------------------------------------
# config 4 as a PIPELINE (hybrid.py:344-358,431-455): four systems end to end, per stage, and the corpus-side encode
# ---------------------------------------------------------------------------------------------------------------------
def measure_pipeline4(dev, N=27942, queries=(1024, 195)):
    """BASELINE.json configs[3] end to end on one GPU: BM25 + DPR + SPLADE + ColBERT (each: query encode where it has an encoder, score,
    full ranking), the 4-way nsf min-max fusion with equal weights (hybrid.py:448) and the final order -- HIP events per stage, queries/s
    for the whole chain.  The corpus side (document embeddings / SPLADE vectors / ColBERT token matrix / BM25 index) is static and built
    once, untimed; what encoding it costs is reported separately (corpus_encode: tokens/s per encoder on a 1/8 LLeQA-shaped sample)."""
    from fusion_amd import encoders, ops
    from fusion_amd.planes import RankedSystem
    from fusion_amd.retrievers.hybrid import Aggregator, _rank_scores
    out = []
    # recorded hipBLASLt solutions stay in use (fusion_amd/tuned/gemm_gfx950.csv), but shapes that are not in the file are NOT tuned on
    # first use here: the SPLADE vocabulary head and the corpus encode's sub-batches bring dozens of new GEMM shapes, minutes of tuning
    tuning_was_on = torch.cuda.tunable.is_enabled() and torch.cuda.tunable.tuning_is_enabled()
    if tuning_was_on:
        torch.cuda.tunable.tuning_enable(False)
    g = torch.Generator(device=dev).manual_seed(31)
    rng = np.random.default_rng(31)
    ids_np = np.arange(N)
    enc = {k: encoders.random_init(k, device=dev, size="base", seed=i) for i, k in enumerate(("dpr", "splade", "colbert"))}
    cfg = enc["dpr"].backbone.config
    V, Vp = cfg.vocab_size, -(-cfg.vocab_size // 4) * 4
    # ---- corpus side, static --------------------------------------------------------------------------------------
    Dn = ops.normalize_rows(torch.randn((N, 768), generator=g, device=dev))
    # SPLADE corpus side: SPLADE-shaped vectors (~200 active terms of 32,005 per document), kept as the product keeps them (Ranker.single_vector_search):
    # an inverted index of the normalised rows.  The dense matrix stays around only to time the dense GEMM on the same data for the record.
    Ds = ops.normalize_rows(splade_like(rng, N, V, 200, dev))
    Ds_index = ops.sparse_index(Ds, V)
    enc["splade"].calibrate_sparsity()                # random-init head -> a trained SPLADE's sparsity (a few dozen active terms per query)
    dl = np.clip(rng.normal(300, 120, N), 16, 512).astype(np.int64)
    off = np.zeros(N + 1, dtype=np.int64); off[1:] = np.cumsum(dl)
    Dtok = torch.nn.functional.normalize(torch.randn((int(off[-1]), 128), generator=g, device=dev), dim=-1).half()
    Doff = torch.from_numpy(off).to(dev)
    Vb, blens, tok, doc, pz = synth_bm25_index(N, np.random.default_rng(99))
    uniq, tf = np.unique(tok.astype(np.int64) * N + doc, return_counts=True)
    pt, pd = uniq // N, uniq % N
    df = np.bincount(pt, minlength=Vb)
    toff = np.zeros(Vb + 1, dtype=np.int64); np.cumsum(df, out=toff[1:])
    bm = dict(toff=torch.from_numpy(toff).to(dev), pdoc=torch.from_numpy(pd.astype(np.int32)).to(dev), ptf=torch.from_numpy(tf.astype(np.int32)).to(dev),
              idf=torch.from_numpy(np.log10((N - df + 0.5) / (df + 0.5))).to(dev), doc_len=torch.from_numpy(blens.astype(np.int32)).to(dev), avgdl=float(blens.mean()))
    bm["doc_norm"] = ops.bm25_doc_norms(bm["doc_len"], bm["avgdl"], 2.5, 0.2)
    bm["slice_off"] = ops.bm25_slice_offsets(bm["toff"], bm["pdoc"], N)

    log("pipeline4: corpus side built")
    for Q in queries:
        qids, _, qlen = synth_query_tokens(rng, Q, V, cfg.pad_token_id)
        qids_d = torch.from_numpy(qids).to(dev)
        cq = np.where(np.arange(64)[None, :] < qlen[:, None], qids, encoders.MASK_TOKEN_ID)     # ColBERT: padded with the mask token, all attended
        cq_d = torch.from_numpy(cq).to(dev)
        qn = rng.integers(4, 16, Q)
        qterms = torch.from_numpy(rng.choice(Vb, size=int(qn.sum()), p=pz).astype(np.int32)).to(dev)
        qoff_h = np.zeros(Q + 1, dtype=np.int64); np.cumsum(qn, out=qoff_h[1:])
        qoff = torch.from_numpy(qoff_h).to(dev)
        names = ["bm25", "dpr", "splade", "colbert"]
        w = {n: 1 / 4 for n in names}                                                           # hybrid.py:448

        def step(mark):
            mark("start")
            B, B32 = ops.bm25_scores(bm["toff"], bm["pdoc"], bm["ptf"], bm["idf"], bm["doc_len"], bm["avgdl"], 2.5, 0.2, qoff, qterms, Q, N, doc_norm=bm["doc_norm"],
                                     slice_off=bm["slice_off"], want_f32=True)      # as BM25.search_device: the float32 plane from the same launch
            mark("bm25_score")
            st4 = torch.empty((4, Q), dtype=torch.float32, device=dev)
            o_b, _, r_b = ops.sort_rows_desc(B, want_keys=False, want_rank=True, stats_out=st4)   # as BM25.search_device
            sys_b = RankedSystem(scores=B32, order=o_b, rank=r_b, lens=torch.full((Q,), N, dtype=torch.int32, device=dev), ids=ids_np,
                                 full=True, scores64=B, score_sorted=True, stats4=st4)
            mark("bm25_rank")
            e = enc["dpr"].encode_ids_packed(qids_d, qlen); mark("dpr_encode")
            S_d = ops.dot_scores(ops.normalize_rows(e), Dn); mark("dpr_score")
            sys_d = _rank_scores(S_d, ids_np, None); mark("dpr_rank")
            v = enc["splade"].encode_ids_packed(qids_d, qlen); mark("splade_encode")
            S_s = ops.sparse_cos_scores(v, Ds_index); mark("splade_score")      # normalise + non-zero lists + fz_sparse_dot_f32
            sys_s = _rank_scores(S_s, ids_np, None); mark("splade_rank")
            Qtok = enc["colbert"].encode_query_ids(cq_d); mark("colbert_encode")
            S_c = ops.maxsim(Qtok, Dtok, Doff, max_doc_len=512); mark("colbert_maxsim")
            sys_c = _rank_scores(S_c, ids_np, None); mark("colbert_rank")
            fused = Aggregator.fuse_device(dict(bm25=sys_b, dpr=sys_d, splade=sys_s, colbert=sys_c), "nsf", "min-max", w, {})
            mark("fuse_and_order")
            return fused
        for _ in range(2):
            step(lambda n: None)
            torch.cuda.synchronize()
            log(f"pipeline4: warm-up step done (Q={Q})")
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            step(lambda n: None)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
        ev = Events()
        step(ev.mark)
        torch.cuda.synchronize()
        stages, _ = ev.durations_ms()
        sumL = int(off[-1])
        v_last = enc["splade"].encode_ids_packed(qids_d, qlen)
        splade_dense_ms = timeit_ms(lambda: ops.dot_scores(ops.normalize_rows(v_last), Ds), n=2, warm=1)    # the reference's dense form on the same vectors
        splade_query_nnz = float(torch.count_nonzero(v_last).item()) / Q
        splade_diff = float((ops.dot_scores(ops.normalize_rows(v_last), Ds) - ops.sparse_cos_scores(v_last, Ds_index)).abs().max())
        del v_last
        enc["colbert"].amp = False                 # the same pipeline with a float32 ColBERT query encoder (both ways)
        colbert_fp32_ms = timeit_ms(lambda: enc["colbert"].encode_query_ids(cq_d), n=2, warm=1)
        step(lambda n: None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            step(lambda n: None)
        torch.cuda.synchronize()
        wall_fp32 = (time.perf_counter() - t0) / reps
        enc["colbert"].amp = True
        out.append(dict(config="4: BM25+DPR+SPLADE+ColBERT pipeline END TO END (query encode x3, score x4, full ranking x4, nsf min-max fusion, final order)",
                        shape=dict(Q=Q, N=N, S=4), ms=wall * 1e3, queries_per_s=Q / wall, ms_colbert_fp32=wall_fp32 * 1e3,
                        queries_per_s_colbert_fp32=Q / wall_fp32, stages_ms=stages,
                        dtypes=dict(dpr="f32", splade="f32", bm25="f64",
                                    colbert="encoder Linears float16 (colbert-ai runs query() / doc() under autocast; multi_dense_biencoder.py:55 'amp': True), everything "
                                            "else of the forward float32; token vectors float16; MaxSim f16 MFMA, f32 accumulate"),
                        colbert_encode_fp32_ms=colbert_fp32_ms,
                        splade=dict(scoring="inverted index of the normalised corpus vectors (fz_sparse_dot_f32): the dense cos_sim's products minus its exact zeros",
                                    doc_nnz_mean=Ds_index.nnz / N, query_nnz_mean=splade_query_nnz, index_MB=(Ds_index.nnz * 8 + (V + 1) * 8) / 1e6,
                                    dense_MB=N * Vp * 4 / 1e6, dense_gemm_ms_same_vectors=splade_dense_ms, max_abs_diff_vs_dense=splade_diff),
                        dominant_stage=max(stages, key=stages.get),
                        stage_rooflines={"dpr_score": roof("dot_scores_kernel", stages["dpr_score"], 2.0 * Q * N * 768, "mfma_f32")["frac"],
                                         "colbert_maxsim": roof("maxsim_kernel", stages["colbert_maxsim"], 2.0 * Q * 64 * sumL * 128, "mfma_f16")["frac"]}))
    del Ds, Ds_index, Dtok, Dn

    # ---- corpus-side encode (hybrid.py:101; the reference's dominant cost, re-paid by each of run_hybrid.sh's processes) -----------
    n = N // 8
    lens = np.clip(rng.normal(300, 120, n), 16, 512).astype(np.int64)
    ids = torch.from_numpy(rng.integers(7, V - 1, (n, 512))).to(dev)
    T = int(lens.sum())
    h, ff, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    body = 2.0 * T * L * (4 * h * h + 2 * h * ff)                                # the four Linears of every layer
    attn = 4.0 * L * h * float((lens.astype(np.float64) ** 2).sum())             # q k^T and p v over each sequence
    heads = {"dpr": 0.0, "splade": 2.0 * T * (h * h + h * V), "colbert": 2.0 * T * h * 128}
    runs = {"dpr": lambda: enc["dpr"].encode_ids_corpus(ids, lens), "splade": lambda: enc["splade"].encode_ids_packed(ids, lens),
            "colbert": lambda: enc["colbert"].encode_doc_ids(ids, lens)}
    for k in ("dpr", "splade", "colbert", "colbert_fp32"):
        log(f"corpus encode: {k}, {T} tokens")
        amp = k == "colbert"                                                     # float16 Linears (the default, as colbert-ai encodes)
        if k.startswith("colbert"):
            enc["colbert"].amp = amp
        run = runs[k.split("_")[0]]
        run()                                                                    # first pass: TunableOp settles the sub-batch shapes
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        fl = body + attn + heads[k.split("_")[0]]
        peak = MFMA_F16_PEAK_TF if amp else MFMA_F32_PEAK_TF                     # the Linears (97 % of the FLOPs) run on that pipe
        what = "float16 Linears (colbert-ai's autocast), float32 elsewhere" if amp else "fp32"
        out.append(dict(config=f"corpus encode, {k} (CamemBERT-base-shaped, {what}, padding-free forward), 1/8 of the LLeQA-shaped corpus",
                        shape=dict(docs=n, tokens=T, mean_len=float(lens.mean())), ms=dt * 1e3, tokens_per_s=T / dt, docs_per_s=n / dt,
                        full_corpus_s_estimate=dt * 8, flops=fl, bound="mfma", achieved=fl / dt / 1e12, peak=peak, unit="TFLOP/s",
                        frac=fl / dt / (peak * 1e12)))
    enc["colbert"].amp = True
    log("corpus encode: mMARCO-shaped passages (config 5's encode leg)")
    out.append(dict(config="5: mMARCO-shaped corpus ENCODE on one GPU (DPR, fp32, padding-free, 65,536-row sub-batches; each GPU encodes its own 1/8 shard, no collective)",
                    **{"shape": dict(passages=16384, lens="clip(N(70,30),8,256)")}, **mmarco_encode_rate(enc["dpr"], dev)))
    if tuning_was_on:
        torch.cuda.tunable.tuning_enable(True)
    return out

# ---------------------------------------------------------------------------------------------------------------------
# N > 1: the corpus-sharded mMARCO config (north_star's multi-GPU configuration)
# ---------------------------------------------------------------------------------------------------------------------
def bench_sharded(args, dev, rank, world, dist):
    from fusion_amd import encoders, ops
    from fusion_amd.distributed import ShardedDenseIndex, allgather_rows, allgather_topk, shard_bounds
    N, d, Q, k = args.mmarco_docs, args.dim, args.queries, args.topk
    lo, hi = shard_bounds(N, world, rank)
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    Dn = torch.empty((hi - lo, d), dtype=torch.float32, device=dev)
    for c0 in range(0, hi - lo, 1 << 20):   # generated on the device, shard by shard: 27 GB never cross PCIe
        c1 = min(hi - lo, c0 + (1 << 20))
        Dn[c0:c1] = ops.normalize_rows(torch.randn((c1 - c0, d), generator=g, device=dev))
    index = ShardedDenseIndex(Dn, lo, None)
    rng = np.random.default_rng(5)
    enc = None
    if not args.no_encode:
        enc = encoders.random_init("dpr", device=dev, size=args.encoder_size, seed=0)
        if not args.no_gemm_tuning:
            encoders.enable_gemm_tuning()
        ids, mask, qlen_all = synth_query_tokens(rng, Q, enc.backbone.config.vocab_size, 1)
        qlo, qhi = shard_bounds(Q, world, rank)                         # this rank encodes its 1/world of the queries
        ids_t = torch.from_numpy(ids[qlo:qhi]).to(dev)
        qlen = qlen_all[qlo:qhi]
    else:
        q_emb = torch.from_numpy(rng.normal(0, 1, (Q, d)).astype(np.float32)).to(dev)

    def step(mark=None):
        if mark: mark("start")
        if enc is not None:
            e_loc = enc.encode_ids_packed(ids_t, qlen)
            if mark: mark("encode_local")
            e = allgather_rows(e_loc, Q)
            if mark: mark("allgather_embeddings")
        else:
            e = q_emb
        Qn = ops.normalize_rows(e)
        s, i = index.local_topk(Qn, k, mark=mark)
        gs, gi = allgather_topk(s, i, None)
        if mark: mark("allgather_topk_merge")
        return gs, gi

    def barrier():
        torch.cuda.synchronize()
        if dist: dist.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    el = time.perf_counter() - t0
    if dist:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    # per-kernel durations of this rank, HIP events, second instrumented region
    tot, cnt = {}, {}
    for _ in range(args.steps):
        ev = Events()
        step(ev.mark)
        torch.cuda.synchronize()
        dt, dc = ev.durations_ms()
        for kx, v in dt.items():
            tot[kx] = tot.get(kx, 0.0) + v; cnt[kx] = cnt.get(kx, 0) + dc[kx]
    stages = {kx: v / args.steps for kx, v in tot.items()}
    per_launch = {kx: tot[kx] / cnt[kx] for kx in tot}
    # the collective on its own (payload = the two [Q, k] lists this rank contributes)
    s0 = torch.zeros((Q, k), dtype=torch.float32, device=dev); i0 = torch.zeros((Q, k), dtype=torch.int64, device=dev)
    ag_ms = None
    if dist:
        gs_ = torch.empty((world * Q, k), dtype=torch.float32, device=dev); gi_ = torch.empty((world * Q, k), dtype=torch.int64, device=dev)
        def ag():
            dist.all_gather_into_tensor(gs_, s0); dist.all_gather_into_tensor(gi_, i0)
        ag_ms = timeit_ms(ag, n=10)

    # correctness of the collective path on a reduced corpus: sharded search == single-GPU search, bit for bit
    Ns = 200_000
    gsm = torch.Generator(device=dev).manual_seed(4242)                  # the same small corpus on every rank
    Dsm = ops.normalize_rows(torch.randn((Ns, d), generator=gsm, device=dev))
    Qsm = ops.normalize_rows(torch.randn((64, d), generator=gsm, device=dev))
    slo, shi = shard_bounds(Ns, world, rank)
    sh = ShardedDenseIndex(Dsm[slo:shi].contiguous(), slo, None); sh.CHUNK = 65536
    ss, si = sh.search(Qsm, k)
    whole = ShardedDenseIndex(Dsm, 0, None); whole.CHUNK = 65536
    ws, wi = whole.local_topk(Qsm, k)
    same = bool(torch.equal(ss, ws) and torch.equal(si, wi))
    if dist:
        flag = torch.tensor([int(same)], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        same = bool(flag.item())
        devs = [None] * world
        dist.all_gather_object(devs, f"rank {rank}: cuda:{dev.index} {torch.cuda.get_device_name(dev)}")
    else:
        devs = [f"rank 0: cuda:{dev.index} {torch.cuda.get_device_name(dev)}"]
    del Dsm, sh, whole

    # config 5's encode leg: every rank encodes mMARCO-shaped passages of ITS shard (data-parallel, no collective); slowest rank reported
    enc_leg = None
    if enc is not None and not args.no_corpus_encode:
        mine = mmarco_encode_rate(enc, dev, n_passages=args.encode_sample, seed=17 + rank, shard_rows=hi - lo)
        rates = [mine]
        if dist:
            rates = [None] * world
            dist.all_gather_object(rates, mine)
        slow = min(rates, key=lambda r: r["passages_per_s"])
        enc_leg = dict(slow, ranks=world, passages_per_s_all_ranks=sum(r["passages_per_s"] for r in rates),
                       what="DenseEncoder.encode_ids_corpus over passages of clip(N(70,30),8,256) pieces on every rank at once; the slowest rank's figures, "
                            "full_shard_encode_s = this rank's shard rows / its rate")

    chunk = min(ShardedDenseIndex.CHUNK, hi - lo)
    rl_all = {}
    if "shard_gemm_filter" in per_launch:   # the shard's GEMM with the threshold filter as its epilogue (everything behind the exact head)
        head_docs = min(hi - lo, max(8192, -(-8 * k // 4096) * 4096))
        rl_all["shard_gemm_filter"] = roof("dot_scores_kernel<filter epilogue>", per_launch["shard_gemm_filter"],
                                           2.0 * Q * (hi - lo - head_docs) * d / cnt["shard_gemm_filter"] * args.steps, "mfma_f32",
                                           launches_per_step=cnt["shard_gemm_filter"] // args.steps)
    elif "shard_gemm" in per_launch:
        rl_all["shard_gemm"] = roof("dot_scores_kernel", per_launch["shard_gemm"], 2.0 * Q * (hi - lo) * d / cnt["shard_gemm"] * args.steps, "mfma_f32",
                                    launches_per_step=cnt["shard_gemm"] // args.steps)
    if "shard_topk_stream" in stages and "shard_gemm_filter" not in per_launch:   # unfused: one streaming pass over all the shard's scores
        rl_all["shard_topk_stream"] = roof("topk_filter (threshold filter, append) + topk_rows head + sort_rows folds", stages["shard_topk_stream"],
                                           Q * (hi - lo) * 4, "hbm", launches_per_step=1)
    dom = max((kx for kx in rl_all), key=lambda kx: stages.get(kx, 0.0))
    rl = dict(rl_all[dom], stage=dom, traffic=None, traffic_source=None)
    res = {"metric": METRIC, "value": Q * args.steps / el, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic",
           "config": {"workload": f"mMARCO-fr-shaped sharded DPR encode+score (BASELINE.json configs[4]): N={N} passages x d={d} fp32 row-sharded x{world}, "
                                  f"Q={Q} queries per step, top-{k}; data-parallel CamemBERT-base-shaped fp32 query encoder + all-gather of embeddings, "
                                  "chunked fp32-MFMA cos-sim GEMM with the streaming top-k's threshold filter as its epilogue (no score plane), ONE RCCL all-gather of per-shard top-k + local merge",
                      "corpus": N, "dim": d, "queries_per_step": Q, "topk": k, "encode_in_step": enc is not None,
                      "parallelism": f"corpus row-sharded x{world} (RCCL all-gather of [Q,k] lists), encoder data-parallel x{world}",
                      **({"corpus_encode_passages_per_s_per_gpu": round(enc_leg["passages_per_s"], 1), "corpus_encode_mfma_f32_frac": round(enc_leg["frac"], 4),
                          "full_shard_encode_s_per_gpu": round(enc_leg["full_shard_encode_s"], 1)} if enc_leg else {})},
           "stages_ms": stages, "roofline": rl, "roofline_all": rl_all,
           "collective": {"op": "all_gather_into_tensor x2 (fp32 scores, int64 ids)", "payload_bytes_per_rank": Q * k * 12,
                          "gathered_bytes_per_rank": world * Q * k * 12, "ms": ag_ms,
                          "embeddings_allgather_bytes_per_rank": (-(-Q // world)) * d * 4 if enc is not None else 0},
           "ranks": devs, "shard_rows": hi - lo, "corpus_encode": enc_leg,
           "sharded_equals_single_gpu": {"equal": same, "corpus": Ns, "queries": 64, "k": k,
                                         "what": "ShardedDenseIndex.search over this world's shards vs one-GPU local_topk of the whole corpus, scores and ids bit for bit"}}
    if rank == 0 and world > 1 and not args.no_one_gpu_reference:
        # the SAME workload on this GPU alone (all queries encoded here, the whole corpus searched here, no collective): what the
        # N-GPU value is to be held against -- the default one-GPU bench line is the LLeQA workload, a different job
        try:
            whole = torch.empty((N, d), dtype=torch.float32, device=dev)
            whole[lo:hi] = Dn
            for r in range(world):
                if r == rank:
                    continue
                rlo, rhi = shard_bounds(N, world, r)
                gr = torch.Generator(device=dev).manual_seed(1000 + r)
                for c0 in range(rlo, rhi, 1 << 20):
                    c1 = min(rhi, c0 + (1 << 20))
                    whole[c0:c1] = ops.normalize_rows(torch.randn((c1 - c0, d), generator=gr, device=dev))
            widx = ShardedDenseIndex(whole, 0, None)
            ids_all = torch.from_numpy(ids).to(dev) if enc is not None else None

            def step1():
                e = enc.encode_ids_packed(ids_all, qlen_all) if enc is not None else q_emb
                return widx.local_topk(ops.normalize_rows(e), k)
            n1 = max(1, min(args.steps, 3))
            step1(); torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n1):
                step1()
            torch.cuda.synchronize()
            e1 = time.perf_counter() - t1
            res["same_workload_one_gpu"] = {"n_gpus": 1, "value": Q * n1 / e1, "unit": "queries/s", "ms_per_step": 1e3 * e1 / n1, "steps": n1,
                                            "what": "rank 0 alone: all queries encoded and the whole corpus searched on one GPU, no collective; "
                                                    "timed after the N-GPU region while the other ranks idle"}
            del whole, widx
        except RuntimeError as ex:   # e.g. out of memory on a smaller part
            res["same_workload_one_gpu"] = {"error": str(ex)[:200]}
    if rank == 0 and not args.no_cpu_baseline:
        # the oracle's search (cos -> top-k) on a bounded sample: 16 queries x 1/8 of this rank's shard, OpenMP
        from oracle import oracle
        oracle.build()
        nq, nd = 16, min(hi - lo, 1 << 20) // 8
        Qh, Dh = ops.normalize_rows(e_sample(dev, nq, d)).cpu().numpy(), Dn[:nd].cpu().numpy()
        t0 = time.perf_counter()
        oracle.topk_rows(oracle.dot_scores(Qh, Dh, fma_chain=True), k)
        dt = time.perf_counter() - t0
        res["cpu_baseline"] = dict(value=nq / (dt * ((hi - lo) * world / nd)), unit="queries/s", cores=oracle.num_threads(), kind="port",
                                   sample=f"oracle cos-sim + top-{k} of {nq} queries x {nd} passages in {dt:.2f} s, scaled linearly to the {N}-passage corpus "
                                          "(score + top-k only, no encoder)")
    return res


def e_sample(dev, n, d):
    g = torch.Generator(device=dev).manual_seed(77)
    return torch.randn((n, d), generator=g, device=dev)


# ---------------------------------------------------------------------------------------------------------------------
# The ONE stdout line: compact by construction.  Round 4's line grew to 20 kB and the driver's record came back unparsed;
# everything bulky (`roofline_all`, `configs_measured`, per-rank device names ...) now goes to bench_detail.json next to this
# script (and gpurun_out/ when present) and to stderr, and tests/test_bench_line_cpu.py pins the line's size.
# ---------------------------------------------------------------------------------------------------------------------
LINE_BUDGET = 6000                                    # bytes; the test asserts it on recorded detail files
HEAD_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")


def _r(x, nd=4):
    """Round floats for the compact line (value / ms_per_step keep every digit); non-finite floats become None: strict JSON."""
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return round(x, nd) if abs(x) < 1e6 else round(x, 1)
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


def _short(s, n=110):
    return s if not isinstance(s, str) or len(s) <= n else s[: n - 3] + "..."


def compact_line(res):
    """The headline dict the driver parses, from the full result `res` (what bench_detail.json holds)."""
    line = {k: res[k] for k in HEAD_KEYS if k in res}
    cfg = dict(res.get("config", {}))
    cfg["workload"] = _short(cfg.get("workload", ""), 240)
    line["config"] = _r(cfg)
    if "stages_ms" in res:
        line["stages_ms"] = _r(res["stages_ms"])
    if "score_fuse_qps_per_gpu" in res:
        line["score_fuse_qps_per_gpu"] = _r(res["score_fuse_qps_per_gpu"], 1)
    rl = res.get("roofline")
    if rl:
        rl = dict(rl)
        if rl.get("traffic_source"):
            rl["traffic_source"] = rl["traffic_source"].split(" ")[0]            # the file name only
        rl["kernel"] = _short(rl.get("kernel", ""), 80)
        line["roofline"] = _r(rl, 5)
    cb = res.get("cpu_baseline")
    if cb:
        cb = dict(cb)
        cb["sample"] = _short(cb.get("sample", ""), 118)
        line["cpu_baseline"] = _r(cb)
    if "parity_sample" in res:
        line["parity_sample"] = {k: (float(f"{v:.3g}") if isinstance(v, float) else v) for k, v in res["parity_sample"].items()}
    ns = res.get("north_star_targets")
    if ns:                                                                       # bare numbers: value, target, met
        line["north_star_targets"] = {k: ({"value": _r(v["value"]), "target": v["target"], "met": v["met"]} if v else None) for k, v in ns.items()}
    for k in ("collective", "same_workload_one_gpu", "rehearsal"):               # the N > 1 line's own pieces
        if k in res:
            v = res[k]
            line[k] = _r({kk: _short(vv, 60) for kk, vv in v.items() if kk != "what"}) if isinstance(v, dict) else _short(v, 100)
    if "sharded_equals_single_gpu" in res:
        line["sharded_equals_single_gpu"] = {k: v for k, v in res["sharded_equals_single_gpu"].items() if k != "what"}
    if "ranks" in res:
        line["ranks"] = len(res["ranks"])
    if "shard_rows" in res:
        line["shard_rows"] = res["shard_rows"]
    line["detail"] = "bench_detail.json"
    return line


def emit(res):
    """Full record -> bench_detail.json (+ gpurun_out/) and stderr; the compact line -> stdout, alone, last."""
    full = json.dumps(res)
    for path in (os.path.join(ROOT, "bench_detail.json"), os.path.join(ROOT, "gpurun_out", "bench_detail.json")):
        try:
            if os.path.isdir(os.path.dirname(path)):
                with open(path, "w") as f:
                    f.write(full + "\n")
        except OSError as ex:
            log(f"could not write {path}: {ex}")
    print("[bench detail] " + full, file=sys.stderr, flush=True)
    line = json.dumps(compact_line(res), allow_nan=False)
    if len(line) > LINE_BUDGET:
        log(f"compact line is {len(line)} B > {LINE_BUDGET}: dropping stages_ms")
        slim = compact_line(res); slim.pop("stages_ms", None)
        line = json.dumps(slim, allow_nan=False)
    sys.stderr.flush()
    print(line, flush=True)


def launcher_argv(argv, n_gpus, port):
    """`python bench.py --gpus N ...` outside torch.distributed.run: the command line of the N-rank job this process starts as a CHILD
    (one rank per GPU, RCCL over xGMI; 127.0.0.1 rendezvous: the container hostname may not resolve) -- the driver's own form."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__), *argv]


def launch_ranks(args):
    """Start the N ranks and relay rank 0's JSON line.  This parent makes NO GPU call (a process that has touched the GPU must not be
    replaced or forked into workers on this pool): it only starts the child, streams its output through, and exits with its code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = launcher_argv(sys.argv[1:], args.gpus, port)
    log(f"--gpus {args.gpus} without WORLD_SIZE: starting {' '.join(cmd[1:7])} ... as a child process")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for out in proc.stdout:
        if out.startswith('{"metric"'):
            line = out.rstrip("\n")          # rank 0's one line: printed last, alone
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        rc = 1
    sys.exit(rc)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)                     # never returns
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    backend = os.environ.get("FUSION_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if local >= max(ndev, 1):
        # one process per GPU: RCCL hangs or fails with two ranks on one device, and a shared device is not an N-GPU measurement
        if not (args.rehearsal and backend != "nccl"):
            raise SystemExit(f"LOCAL_RANK {local} but only {ndev} visible GPU(s): one rank per GPU (pass --rehearsal with "
                             "FUSION_BENCH_BACKEND=gloo to rehearse the control flow on a smaller box)")
        local %= max(ndev, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    workload = args.workload if args.workload != "auto" else ("mmarco" if world > 1 else "lleqa")

    if workload == "mmarco":
        res = bench_sharded(args, dev, rank, world, dist)
        if args.rehearsal:
            res["rehearsal"] = "ranks share devices: control-flow rehearsal only, the numbers mean nothing"
        if rank == 0:
            emit(res)
        if dist:
            torch.cuda.synchronize()
            dist.barrier()                      # rank 0's extras (one-GPU reference, CPU baseline) end before any rank tears the group down
            dist.destroy_process_group()
        return

    st = build_lleqa(args, dev, rank)
    if "texts" in st:   # the tokenising thread's Python part (ids -> arrays, ~4 ms) must not hold the launching thread off for a whole default 5 ms slice
        sys.setswitchinterval(float(os.environ.get("FZ_SWITCH_INTERVAL", "0.0005")))
    open_query_stream(st, args.warmup + args.steps + 1, serial=args.serial_tokenize)
    run_steps(st, args.warmup)

    def barrier():
        torch.cuda.synchronize()
        if dist: dist.barrier()
        torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    out = run_steps(st, args.steps)      # text -> ids is INSIDE the timed region: one batch tokenised (for the next step) per step
    barrier()
    elapsed = time.perf_counter() - t0
    close_query_stream(st)
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel durations, live, with HIP events over a second instrumented region of the same K steps
    tot, cnt = {}, {}
    for _ in range(args.steps):
        ev = Events()
        step_lleqa(st, ev)
        torch.cuda.synchronize()
        dt, dc = ev.durations_ms()
        for k, v in dt.items():
            tot[k] = tot.get(k, 0.0) + v; cnt[k] = cnt.get(k, 0) + dc[k]
    stages = {k: v / args.steps for k, v in tot.items()}                               # ms per step
    per_launch = {k: tot[k] / cnt[k] for k in tot}                                     # ms per launch
    stages["encode"] = sum(v for k, v in stages.items() if k.startswith("encode_"))    # the whole forward

    host = None
    if rank == 0 and "texts" in st:
        # what text -> ids costs on this host, on its own (device idle), the upload, and the same steps WITHOUT the overlap
        st["tok"].encode_np(st["texts"], 64, pad_to_max=True)
        t0 = time.perf_counter()
        for _ in range(5):
            st["tok"].encode_np(st["texts"], 64, pad_to_max=True)
        tok_ms = (time.perf_counter() - t0) / 5 * 1e3
        h2d_ms = timeit_ms(lambda: st["ids"].copy_(st["pinned"][0], non_blocking=True), n=20)
        ns = max(2, min(5, args.steps))
        open_query_stream(st, 2 + ns + 1, serial=not args.serial_tokenize)
        run_steps(st, 2)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run_steps(st, ns)
        torch.cuda.synchronize()
        other_ms = (time.perf_counter() - t0) / ns * 1e3
        close_query_stream(st)
        host = dict(tokenize_ms=tok_ms, h2d_ids_ms=h2d_ms, host_threads=os.cpu_count(),
                    pieces_per_query=float(np.mean(st["qlen"])), tokenizer="32,005-piece BPE, camembert layout, trained on synthetic French-like text (tools/train_synth_tokenizer.py)",
                    overlapped=not args.serial_tokenize,
                    **{("ms_per_step_overlapped" if args.serial_tokenize else "ms_per_step_serial_tokenize"): other_ms})

    if rank == 0:
        Q, N, d = st["Q"], st["N"], st["d"]
        work = algorithmic_work(st)
        all_roof = {k: roof(work[k]["kernel"], per_launch[k], work[k]["work"], work[k]["bound"], ms_per_step=stages[k],
                            launches_per_step=cnt[k] // args.steps, hand_written=work[k].get("hand", True)) for k in work if k in per_launch}
        hand = {k: v for k, v in all_roof.items() if k in PATH_STAGES}                  # the path's own kernels
        dom = max(hand, key=lambda k: hand[k]["ms_per_step"])                          # dominant = most time per step among them
        traffic, src = profiled_traffic(dom, st)
        for k in all_roof:                                                             # every stage carries its profiled traffic too
            t_k, s_k = profiled_traffic(k, st)
            all_roof[k]["traffic"], all_roof[k]["traffic_source"] = t_k, s_k
        rl = dict(hand[dom], stage=dom, traffic=traffic,
                  traffic_source=(src + " (rocprofv3 PMC pass of this command at this shape; profile-derived, not measured in this run)") if src else None)
        res = {
            "metric": METRIC,
            "value": world * Q * args.steps / elapsed, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"LLeQA-shaped BM25+DPR RRF hybrid: Q={Q} queries/GPU x N={N} articles, d={d}; "
                                   f"{'CamemBERT-base-shaped fp32 query encoder (random init; ' + st.get('encode_mode', '') + ' forward' + ('' if args.no_gemm_tuning or st.get('encode_mode') != 'packed' else ', hipBLASLt solutions recorded with TunableOp') + ') + ' if not args.no_encode else 'NO encoder + '}"
                                   "fp32-MFMA cos-sim + BM25(f64) + full stable ranking + RRF(f64) + final order",
                       "queries_per_gpu": Q, "corpus": N, "dim": d, "fusion": "rrf", "systems": ["bm25", "dpr"],
                       "bm25_rows_zero_share_estimate": round(st["bm25_zero_share_estimate"], 4), "bm25_rank_sort": "lexical (zero-compacting)" if st["bm25_lexical"] else "plain",
                       "encode_in_step": not args.no_encode, "parallelism": f"query-sharded x{world}, corpus replicated",
                       "input": "query strings (tokenised on the host inside the timed step)" if host else "token ids resident on the device",
                       "tokenize_in_step": host is not None, "tokenize_overlapped_with_device": bool(host and host["overlapped"]),
                       **({"tokenizer_fallback": True, "tokenizer_fallback_reason": st["tokenizer_fallback"]} if st.get("tokenizer_fallback") else {}),
                       **({"tokenizer_threads": st["tokenizer_threads"]} if host and "tokenizer_threads" in st else {}),
                       **({"tokenize_ms": round(host["tokenize_ms"], 3), "h2d_ids_ms": round(host["h2d_ids_ms"], 4),
                           "ms_per_step_serial_tokenize": round(host.get("ms_per_step_serial_tokenize", float("nan")), 3)} if host else {})},
            "stages_ms": stages, "host": host,
            "score_fuse_qps_per_gpu": Q / (sum(v for k, v in stages.items() if not k.startswith("encode")) * 1e-3),
            "roofline": rl, "roofline_all": all_roof,
        }
        if not args.no_cpu_baseline:
            _, _, (S, B, q_emb) = out
            cb, (S_o, B_o, o_f, k_f), q = cpu_baseline_lleqa(st, q_emb)
            res["cpu_baseline"] = cb
            # parity on the sample while we are here (same embeddings on both sides): device scores vs oracle scores,
            # device BM25 vs oracle BM25, and the final ranked lists against the oracle fed with the DEVICE's own scores
            order, scores, _ = out
            _, _, r_d = oracle_sort(S[:q]); o_b, _, r_b = oracle_sort(B[:q])
            from oracle import oracle
            f = oracle.fuse_rank([r_b, r_d], np.full((2, q), N, dtype=np.int32), "rrf")
            o_ref, k_ref = oracle.sort_rows_desc(f, init_order=o_b)
            res["parity_sample"] = {"queries": q, "cos_max_abs_err": float(np.max(np.abs(S[:q].cpu().numpy() - S_o))),
                                    "bm25_bit_exact": bool(np.array_equal(B[:q].cpu().numpy(), B_o)),
                                    "ranked_lists_identical": bool(np.array_equal(order[:q].cpu().numpy(), o_ref)),
                                    "fused_scores_bit_exact": bool(np.array_equal(scores[:q].cpu().numpy(), k_ref))}
        if world == 1 and not args.no_configs and (Q, N) == (1024, 27942):
            del st, out
            torch.cuda.empty_cache()
            log("headline + cpu baseline done; configs_measured ...")
            res["configs_measured"] = measure_configs(dev, N)
            torch.cuda.empty_cache()
            log("configs 2-5 done; config-4 pipeline ...")
            res["configs_measured"] += measure_pipeline4(dev, N)
            # the figures the round's work moves, repeated inside `config` (the driver's parsed record keeps `config`, not the stdout tail)
            for c in res["configs_measured"]:
                name, Qc = c.get("config", ""), c.get("shape", {}).get("Q")
                if name.startswith("4: BM25+DPR+SPLADE+ColBERT pipeline") and Qc in (1024, 195):
                    res["config"][f"config4_ms_q{Qc}"] = round(c["ms"], 2)
                    res["config"][f"config4_queries_per_s_q{Qc}"] = round(c["queries_per_s"], 1)
                    res["config"][f"config4_ms_colbert_fp32_q{Qc}"] = round(c["ms_colbert_fp32"], 2)
                elif name.startswith("4: nsf percentile-rank, S=4, P=") and Qc == 1024:
                    res["config"]["config4_percentile_P27943_ms"] = round(c["ms"], 4)
                    res["config"]["config4_percentile_P27943_hbm_frac"] = round(c["frac"], 4)
                elif name.startswith("boundary: Aggregator.fuse rrf"):
                    res["config"]["boundary_rrf_ms_per_query"] = round(c["ms_per_query"], 2)
                elif name.startswith("5: mMARCO-shaped corpus ENCODE"):
                    res["config"]["config5_encode_passages_per_s"] = round(c["passages_per_s"], 1)
                    res["config"]["config5_encode_tokens_per_s"] = round(c["tokens_per_s"], 1)
                    res["config"]["config5_encode_mfma_f32_frac"] = round(c["frac"], 4)
                    res["config"]["config5_full_shard_encode_s_per_gpu_at_8"] = round(c["full_shard_encode_s"], 1)
                elif name.startswith("5: mMARCO 1/8 shard"):
                    res["config"]["config5_shard_search_ms_q1024"] = round(c["ms"], 2)
        res["north_star_targets"] = north_star_targets(res)
        emit(res)
    if dist:
        dist.destroy_process_group()


def north_star_targets(res):
    """The two numeric targets BASELINE.json's north_star states, read off THIS run: >= 90 % of the HBM peak on the normalisation +
    fusion pass and >= 70 % MFMA utilisation on DPR scoring at query batch 1024 (fp32 MFMA: the 1e-4 score contract rules out f16)."""
    d = res["roofline_all"].get("dpr_score")
    out = {"dpr_mfma_frac": None if d is None else {"value": d["frac"], "target": 0.70, "met": d["frac"] >= 0.70, "kernel": d["kernel"],
                                                      "where": "dpr_score stage of the timed step (Q = queries_per_gpu)", "peak_TFLOPs": MFMA_F32_PEAK_TF}}
    fuse = {}
    r = res["roofline_all"].get("fuse_rrf")
    if r is not None:
        fuse["rrf S=2, in the timed step"] = r["frac"]
    for c in res.get("configs_measured", []):
        if c.get("config", "").startswith("4: nsf") and c.get("shape", {}).get("Q") == 1024:
            fuse[c["config"][3:]] = c["frac"]
        elif c.get("config", "").startswith("1: rrf fusion alone") and c.get("shape", {}).get("Q") == 1024:
            fuse["rrf S=2, fuse_rank_kernel alone"] = c["frac"]
    best = max(fuse.values()) if fuse else None
    out["fuse_hbm_frac"] = {"value": best, "target": 0.90, "met": bool(best is not None and best >= 0.90), "all": fuse, "peak_GBs": HBM_PEAK_GBS,
                            "note": "fractions of the 8.0 TB/s spec; MI355X_MICROARCH.md measures 6.29 TB/s for a float4 copy on this part (0.79 of spec), "
                                    "so 0.90 of spec is above what the memory system delivers -- the fractions are reported against spec all the same"}
    return out


def oracle_sort(plane):
    from oracle import oracle
    return oracle.sort_rows_desc(plane.cpu().numpy(), want_rank=True)


if __name__ == "__main__":
    main()
