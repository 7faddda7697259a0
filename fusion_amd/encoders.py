"""PyTorch-ROCm transformer forwards for the encode stage (plumbing, not the product).

Reference call sites:
  DPR     SentenceTransformerCustom(...).encode(batch_size=64, convert_to_tensor=True)  hybrid.py:98-102,
          built as Transformer + mean Pooling (src/utils/common.py:13-20, scripts/run_dpr.sh:27), fp32, max_seq_length 512
  SPLADE  SPLADE.forward: amax_L log1p(relu(logits * mask))  (splade/splade.py:88-99), encode() splade/base.py:253-291,
          max_query_length 64 / max_doc_length 512 (hybrid.py:96)
  ColBERT query_maxlen 64 (queries padded with [MASK] and attended, run_colbert.sh:29), doc_maxlen 512, dim 128,
          cosine similarity = L2-normalised token vectors (run_colbert.sh:26-27), punctuation masked in documents (:28)

No checkpoints or tokenizer files exist offline, so `random_init()` builds CamemBERT-base-SHAPED modules with random
weights (same FLOPs / bytes as the real ones) and `HashTokenizer` maps whitespace tokens to ids deterministically.
With real checkpoints on disk, `from_pretrained()` loads them through transformers.
"""
from __future__ import annotations

import hashlib
import os

import numpy as np
import torch
import torch.nn as nn

from .tokenization import prefetch

CAMEMBERT_BASE = dict(vocab_size=32005, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                      max_position_embeddings=514, type_vocab_size=1, pad_token_id=1, bos_token_id=5, eos_token_id=6)
TINY = dict(vocab_size=512, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
            max_position_embeddings=130, type_vocab_size=1, pad_token_id=1, bos_token_id=5, eos_token_id=6)
MASK_TOKEN_ID = 32004  # camembert <mask>


class HashTokenizer:
    """Deterministic whitespace tokenizer for synthetic text (no sentencepiece model offline)."""

    def __init__(self, vocab_size: int, pad_id: int = 1, bos_id: int = 5, eos_id: int = 6, mask_id: int | None = None):
        self.vocab_size, self.pad_token_id, self.bos_id, self.eos_id = vocab_size, pad_id, bos_id, eos_id
        self.mask_token_id = mask_id if mask_id is not None else vocab_size - 1
        self._first = 7

    def _tok(self, w: str) -> int:
        h = int.from_bytes(hashlib.blake2b(w.encode("utf-8"), digest_size=4).digest(), "little")
        return self._first + h % (self.vocab_size - self._first - 1)

    def __call__(self, texts: list[str], max_length: int, pad_to_max: bool = False):
        rows = [[self.bos_id] + [self._tok(w) for w in t.split()][: max_length - 2] + [self.eos_id] for t in texts]
        L = max_length if pad_to_max else max(len(r) for r in rows)
        ids = torch.full((len(rows), L), self.pad_token_id, dtype=torch.long)
        mask = torch.zeros((len(rows), L), dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, : len(r)] = torch.tensor(r)
            mask[i, : len(r)] = 1
        return ids, mask


def _backbone(cfg: dict, mlm: bool = False):
    from transformers import CamembertConfig, CamembertForMaskedLM, CamembertModel
    c = CamembertConfig(**cfg)
    return CamembertForMaskedLM(c) if mlm else CamembertModel(c, add_pooling_layer=False)


class _Base(nn.Module):
    max_query_length = 64
    max_doc_length = 512
    packed_tokens = 65536   # token rows per padding-free forward (FFN activations: 65536 x 3072 fp32 = 0.8 GB)

    def _packed_forward(self, backbone):
        """PackedBertForward over `backbone` when this encoder sits on a GPU and the model has 64-wide heads, else None."""
        if getattr(self, "_packed", None) is None and self._device.type == "cuda" and PackedBertForward.supports(backbone.config):
            self._packed = PackedBertForward(backbone)
        return getattr(self, "_packed", None)

    def __init__(self, tokenizer, device):
        super().__init__()
        self.tokenizer = tokenizer
        self._device = torch.device(device)

    def _clamp_lengths(self, config):
        """Never tokenise past the model's position table (RoBERTa-style positions start at padding_idx + 1)."""
        limit = int(getattr(config, "max_position_embeddings", 514)) - 2
        self.max_query_length = min(self.max_query_length, limit)
        self.max_doc_length = min(self.max_doc_length, limit)

    def _batches(self, sentences, batch_size, max_len, pad_to_max=False):
        # sort by length as SentenceTransformer.encode does, restore order afterwards; the NEXT sub-batch is tokenised on a host thread
        # while this one's forward is launched (tokenization.prefetch: same batches, same order)
        order = sorted(range(len(sentences)), key=lambda i: -len(sentences[i]))

        def host():
            for s in range(0, len(order), batch_size):
                idx = order[s: s + batch_size]
                yield (idx, *self.tokenizer([sentences[i] for i in idx], max_len, pad_to_max))
        for idx, ids, mask in prefetch(host()):
            yield idx, ids.to(self._device, non_blocking=True), mask.to(self._device, non_blocking=True)


def _token_batches(base, sentences, max_len, batch_size, tokenize=None):
    """Sub-batches for the padding-free forward, longest sentences first (as SentenceTransformer.encode sorts), cut by an
    estimate of the TOKEN count (activations stay under ~1 GB) rather than by sentence count.
    Yields (indices into `sentences`, ids [b, L] on the HOST, lengths [b] numpy); sub-batch i + 1 is tokenised on a host thread while
    the caller runs sub-batch i on the device (tokenization.prefetch)."""
    order = sorted(range(len(sentences)), key=lambda i: -len(sentences[i]))

    def host():
        s = 0
        while s < len(order):
            step = max(batch_size, base.packed_tokens // min(max_len, 8 + 2 * len(sentences[order[s]].split())))
            idx = order[s: s + step]
            ids, mask = (tokenize or base.tokenizer)([sentences[i] for i in idx], max_len)
            yield idx, ids, mask.sum(1).numpy()
            s += step
    return prefetch(host())


def _id_batches(lengths, max_tokens: int):
    """Sub-batches of already tokenised sequences for the padding-free forward: longest first (as SentenceTransformer.encode sorts),
    cut where the running TOKEN count would pass `max_tokens`.  Yields index arrays into `lengths`."""
    import numpy as np
    lengths = np.asarray(lengths, dtype=np.int64)
    order = np.argsort(-lengths, kind="stable")
    csum = np.cumsum(lengths[order])
    s = 0
    while s < len(order):
        e = int(np.searchsorted(csum, (csum[s - 1] if s else 0) + max_tokens, side="right"))
        e = max(e, s + 1)
        yield order[s:e]
        s = e


class FusedBertForward:
    """Lean fp32 forward of a BERT/RoBERTa-style encoder for a whole query batch (same weights, same maths as the HF
    module; checked against it in tests).  What it does differently from calling the HF module per length bucket:

      * every Linear runs ONCE per layer on the token rows of ALL buckets concatenated ([T, 768] with T ~ 40 k instead
        of eight [5 k, 768] slices): hipBLASLt's fp32-MFMA GEMMs run at 111-148 TFLOP/s at M = 40 k vs 117-132 at
        M = 5 k, and the launch count drops 8x;
      * Q, K, V projections are one [T,768] x [768,2304] GEMM (fused weight);
      * attention (the only per-sequence part) runs per bucket on views of the concatenated activations.
    Padding inside a bucket is at most (bucket's longest - shortest) tokens per sequence.
    """

    def __init__(self, backbone):
        emb = backbone.embeddings
        self.word, self.pos, self.type0 = emb.word_embeddings.weight, emb.position_embeddings.weight, emb.token_type_embeddings.weight[0]
        self.emb_ln = (emb.LayerNorm.weight, emb.LayerNorm.bias, emb.LayerNorm.eps)
        self.pad_idx = emb.padding_idx if getattr(emb, "padding_idx", None) is not None else backbone.config.pad_token_id
        self.heads = backbone.config.num_attention_heads
        if getattr(backbone.config, "hidden_act", "gelu") != "gelu":
            raise ValueError(f"{type(self).__name__}: hidden_act {backbone.config.hidden_act!r} is not the erf GELU this forward applies")
        self.layers = []
        for lyr in backbone.encoder.layer:
            a, o = lyr.attention.self, lyr.attention.output
            self.layers.append(dict(
                wqkv=torch.cat([a.query.weight, a.key.weight, a.value.weight], 0).contiguous(),
                bqkv=torch.cat([a.query.bias, a.key.bias, a.value.bias], 0).contiguous(),
                wo=o.dense.weight, bo=o.dense.bias, ln1=(o.LayerNorm.weight, o.LayerNorm.bias, o.LayerNorm.eps),
                w1=lyr.intermediate.dense.weight, b1=lyr.intermediate.dense.bias,
                w2=lyr.output.dense.weight, b2=lyr.output.dense.bias,
                ln2=(lyr.output.LayerNorm.weight, lyr.output.LayerNorm.bias, lyr.output.LayerNorm.eps)))

    @torch.no_grad()
    def __call__(self, input_ids: torch.Tensor, lengths, n_buckets: int = 8) -> torch.Tensor:
        """input_ids [n, Lmax] (padded with pad_idx), lengths: HOST token counts -> mean-pooled [n, hidden] fp32."""
        import numpy as np
        F = torch.nn.functional
        dev = input_ids.device
        lengths = np.asarray(lengths)
        n = len(lengths)
        order = np.argsort(-lengths, kind="stable")
        per = -(-n // max(1, n_buckets))
        buckets, ids_rows, pos_rows, off = [], [], [], 0
        for s in range(0, n, per):
            idx = order[s: s + per]
            L = int(lengths[idx[0]])
            sel = torch.from_numpy(idx).to(dev, non_blocking=True)
            ids = input_ids.index_select(0, sel)[:, :L]
            lens = torch.from_numpy(lengths[idx].astype(np.int64)).to(dev, non_blocking=True)
            keep = torch.arange(L, device=dev).unsqueeze(0) < lens.unsqueeze(1)             # [b, L] real tokens
            pos = torch.where(keep, torch.arange(L, device=dev).unsqueeze(0) + self.pad_idx + 1,
                              torch.full((1, 1), self.pad_idx, device=dev, dtype=torch.long))  # RoBERTa position ids
            # additive key mask built ONCE per bucket (SDPA would otherwise re-materialise it from the bool mask in every layer)
            amask = torch.zeros((len(idx), 1, 1, L), dtype=torch.float32, device=dev).masked_fill_(~keep[:, None, None, :], float("-inf"))
            buckets.append((sel, len(idx), L, off, keep, lens, amask))
            ids_rows.append(ids.reshape(-1)); pos_rows.append(pos.reshape(-1))
            off += len(idx) * L
        ids_cat, pos_cat = torch.cat(ids_rows), torch.cat(pos_rows)
        x = self.word[ids_cat] + self.pos[pos_cat] + self.type0
        x = F.layer_norm(x, (x.shape[1],), *self.emb_ln)
        H = self.heads
        D = x.shape[1] // H
        for ly in self.layers:
            qkv = F.linear(x, ly["wqkv"], ly["bqkv"])                                        # [T, 3*hidden], one GEMM
            ctx = torch.empty_like(x)
            for sel, b, L, o, keep, lens, amask in buckets:
                blk = qkv[o: o + b * L].view(b, L, 3, H, D)
                q, k, v = blk[:, :, 0].transpose(1, 2), blk[:, :, 1].transpose(1, 2), blk[:, :, 2].transpose(1, 2)
                a = F.scaled_dot_product_attention(q, k, v, attn_mask=amask)
                ctx[o: o + b * L] = a.transpose(1, 2).reshape(b * L, H * D)
            x = F.layer_norm(F.linear(ctx, ly["wo"], ly["bo"]) + x, (x.shape[1],), *ly["ln1"])
            h = F.gelu(F.linear(x, ly["w1"], ly["b1"]))
            x = F.layer_norm(F.linear(h, ly["w2"], ly["b2"]) + x, (x.shape[1],), *ly["ln2"])
        out = torch.empty((n, x.shape[1]), dtype=torch.float32, device=dev)
        for sel, b, L, o, keep, lens, amask in buckets:
            hb = x[o: o + b * L].view(b, L, -1)
            m = keep.unsqueeze(-1).to(hb.dtype)
            out.index_copy_(0, sel, ((hb * m).sum(1) / lens.clamp_min(1).unsqueeze(1).to(hb.dtype)).float())   # Pooling(mean)
        return out


TUNED_GEMMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "gemm_gfx950.csv")


def enable_gemm_tuning(results_file: str | None = TUNED_GEMMS, max_duration_ms: int = 30) -> None:
    """Let PyTorch's TunableOp pick the hipBLASLt / rocBLAS solution of every fp32 Linear shape (process-wide switch).
    The library heuristics leave 10 % on the table at the encoder's shapes (M = 37 k packed rows: QKV 1.03 -> 0.91 ms, out
    0.37 -> 0.31, FFN 1.27 / 1.30 -> 1.20 / 1.15; same fp32-MFMA arithmetic).  Results recorded in `results_file` (this image's
    library versions; TunableOp ignores the file if its validators differ) are reused; a shape that is not in it is tuned on
    first use (a fraction of a second per shape) -- PackedBertForward pads its row count to a multiple of 512 while tuning is
    on, so batches of similar size share their entries.  New results go to a scratch file, never into the package."""
    import tempfile
    import torch.cuda.tunable as tn
    torch.cuda.init()            # the recorded results are validated against the live device / library versions
    tn.enable(True)
    tn.tuning_enable(True)
    tn.set_max_tuning_duration(int(max_duration_ms))
    tn.set_filename(os.path.join(tempfile.gettempdir(), f"fusion_amd_tunableop_{os.getpid()}.csv"))
    if results_file and os.path.exists(results_file):
        tn.read_file(results_file)


class PackedBertForward(FusedBertForward):
    """The same forward with NO padding anywhere: token rows of all sequences are packed back to back ([T, hidden], T = the
    real token count -- 9 % fewer rows than eight length buckets at the LLeQA query-length mix), the Linears run on the
    packed rows, and everything between them is the library's HIP kernels (include/fusion_hip.h): fz_embed_layernorm_f32
    (embedding gather + sum + LayerNorm), fz_attn_varlen_f32 (attention straight from the fused-QKV rows, no gather/scatter,
    no mask), fz_add_layernorm_f32 (residual + LayerNorm in one pass), fz_gelu_f32 (the FFN's erf GELU, in place) and fz_segment_mean_f32 (mean Pooling).
    head_dim must be 64 (BERT-base family)."""

    ROW_GRANULE = 512
    AMP_ATTENTION = True    # the mixed-precision forward's attention on float16 MFMAs (autocast's matmuls); False: float32 arithmetic on the float16 rows
    FULL_ROWS = 65536       # = _Base.packed_tokens: the corpus encode cuts its sub-batches at this many token rows
    # Mixed precision as torch.autocast applies it to a BERT forward (what colbert-ai's Checkpoint wraps every query() / doc() in):
    # the Linears take float16 operands (float32 accumulate on the matrix pipe); everything between them -- attention, GELU,
    # residual + LayerNorm -- stays in float32 here (autocast keeps LayerNorm and softmax in float32 and lets GELU follow its float16
    # input: this forward is never less precise than that).  None = float32 Linears.
    amp_dtype = None

    def _low(self, t: torch.Tensor) -> torch.Tensor:
        """The weight / bias in amp_dtype, converted once."""
        cache = self.__dict__.setdefault("_low_cache", {})
        key = (id(t), self.amp_dtype)
        if key not in cache:
            cache[key] = (t, t.detach().to(self.amp_dtype))      # keeps `t` alive: ids are not reused
        return cache[key][1]

    def _linear(self, x, w, b):
        if self.amp_dtype is None:
            return torch.nn.functional.linear(x, w, b)
        return torch.nn.functional.linear(x.to(self.amp_dtype), self._low(w), self._low(b)).float()

    def _layers_f16(self, x, ctx, strips_d, H, mark, T):
        """The encoder layers with float16 Linears end to end: every Linear takes a float16 operand written by the kernel in front of it
        (attention, residual + LayerNorm and GELU emit it next to / instead of their float32 result) and returns float16, as under
        autocast; the residual stream, the attention arithmetic and the normalisations are float32.  No conversion pass anywhere."""
        from . import ops
        F = torch.nn.functional
        lo = self._low
        x16 = x.to(torch.float16)
        ctx16 = torch.empty(ctx.shape, dtype=torch.float16, device=x.device)
        if T < ctx16.shape[0]:
            ctx16[T:].zero_()                      # the pad rows (attention writes the real ones in every layer)
        for ly in self.layers:
            qkv16 = F.linear(x16, lo(ly["wqkv"]), lo(ly["bqkv"])); mark("encode_gemm")
            ops.attn_varlen_f16(qkv16, strips_d, H, ctx16, amp=self.AMP_ATTENTION); mark("encode_attn")
            y16 = F.linear(ctx16, lo(ly["wo"]), lo(ly["bo"])); mark("encode_gemm")
            x = ops.add_layernorm_x16(y16, x, *ly["ln1"], out16=x16); mark("encode_ln")
            h16 = F.linear(x16, lo(ly["w1"]), lo(ly["b1"])); mark("encode_gemm")
            ops.gelu_f16_(h16); mark("encode_gelu")
            y16 = F.linear(h16, lo(ly["w2"]), lo(ly["b2"])); mark("encode_gemm")
            x = ops.add_layernorm_x16(y16, x, *ly["ln2"], out16=x16); mark("encode_ln")
        return x

    @staticmethod
    def supports(config) -> bool:
        # 64-wide heads (the attention kernel) and the exact erf GELU (fz_gelu_f32): the BERT-base / CamemBERT family
        return config.hidden_size == 64 * config.num_attention_heads and getattr(config, "hidden_act", "gelu") == "gelu"

    @torch.no_grad()
    def hidden(self, input_ids: torch.Tensor, lengths, mark=None):
        """input_ids [n, Lmax] (anything beyond a row's length is ignored), lengths: HOST token counts ->
        (last hidden states of the real tokens, packed [T, hidden] fp32 in row order; cu_rows [n+1] int32 on the device).
        `mark(name)`: optional instrumentation hook (bench.py records a HIP event per call); every call closes the interval
        since the previous one: "encode_embed" the embedding kernel (+ index uploads), "encode_gemm" one hipBLASLt Linear,
        "encode_attn" / "encode_ln" / "encode_gelu" one launch of fz_attn_varlen_f32 / fz_add_layernorm_f32 / fz_gelu_f32."""
        import numpy as np
        from . import ops
        F = torch.nn.functional
        dev = input_ids.device
        n, Lmax = input_ids.shape
        H = self.heads
        if self.word.shape[1] != H * 64:
            raise ValueError(f"PackedBertForward: head_dim {self.word.shape[1] // H} is not 64 (use FusedBertForward)")
        lengths = np.minimum(np.asarray(lengths, dtype=np.int64), Lmax)
        strips, cu = ops.attn_strips(lengths)
        T = int(cu[-1])
        tables = torch.from_numpy(np.concatenate([strips.reshape(-1), cu])).to(dev, non_blocking=True)
        strips_d, cu_d = tables[: strips.size].view(-1, 4), tables[strips.size:]
        if T == 0:
            return torch.zeros((0, self.word.shape[1]), dtype=torch.float32, device=dev), cu_d
        cols = np.arange(T, dtype=np.int64) - np.repeat(cu[:-1].astype(np.int64), lengths)
        host = np.concatenate([np.repeat(np.arange(n, dtype=np.int64) * Lmax, lengths) + cols, cols + (self.pad_idx + 1)])
        meta = torch.from_numpy(host).to(dev, non_blocking=True)                       # one upload: gather indices + position ids
        ids = input_ids.reshape(-1)[meta[:T]]
        # while GEMM tuning is on, the row count is padded to a multiple of ROW_GRANULE (zero rows: every op between the GEMMs is
        # row-wise, so they never touch a real row) -- tuned solutions are keyed by the exact shape
        Tp = -(-T // self.ROW_GRANULE) * self.ROW_GRANULE if torch.cuda.tunable.is_enabled() else T
        if torch.cuda.tunable.is_enabled() and self.FULL_ROWS * 7 // 8 < T <= self.FULL_ROWS:
            Tp = self.FULL_ROWS                  # a (nearly) full sub-batch of the corpus encode: ONE row count, the one the recorded solutions are for
        x = None
        if Tp != T:                               # only the pad rows need zeros: the real ones are written by the kernel below
            x = torch.empty((Tp, self.word.shape[1]), dtype=torch.float32, device=dev)
            x[T:].zero_()
        # embedding gather + position + type + LayerNorm in one pass over the packed rows
        x = ops.embed_layernorm(self.word, self.pos, self.type0, ids, meta[T:], *self.emb_ln, out=x)
        ctx = torch.empty_like(x)                 # attention writes the real rows of this buffer in every layer; the pad rows stay zero
        if Tp != T:
            ctx[T:].zero_()
        mark = mark or (lambda name: None)
        mark("encode_embed")
        if self.amp_dtype == torch.float16 and self.word.shape[1] % 8 == 0 and self.layers[0]["w1"].shape[0] % 8 == 0:
            return self._layers_f16(x, ctx, strips_d, H, mark, T)[:T], cu_d
        for ly in self.layers:
            qkv = self._linear(x, ly["wqkv"], ly["bqkv"]); mark("encode_gemm")
            ops.attn_varlen(qkv, strips_d, H, out=ctx); mark("encode_attn")
            y = self._linear(ctx, ly["wo"], ly["bo"]); mark("encode_gemm")
            x = ops.add_layernorm(y, x, *ly["ln1"]); mark("encode_ln")
            h = self._linear(x, ly["w1"], ly["b1"]); mark("encode_gemm")
            h = ops.gelu_(h); mark("encode_gelu")                   # in place, non-temporal loads
            y = self._linear(h, ly["w2"], ly["b2"]); mark("encode_gemm")
            x = ops.add_layernorm(y, x, *ly["ln2"]); mark("encode_ln")
        return x[:T], cu_d

    @torch.no_grad()
    def __call__(self, input_ids: torch.Tensor, lengths, n_buckets: int = 0, mark=None) -> torch.Tensor:
        """-> mean-pooled [n, hidden] fp32 (zeros for an empty sequence)."""
        from . import ops
        x, cu_d = self.hidden(input_ids, lengths, mark)
        return ops.segment_mean(x, cu_d)


class DenseEncoder(_Base):
    """DPR bi-encoder: CamemBERT + mean pooling over the attention mask, fp32."""

    def __init__(self, backbone, tokenizer, device):
        super().__init__(tokenizer, device)
        self.backbone = backbone.to(self._device).eval()
        self.dim = backbone.config.hidden_size
        self._clamp_lengths(backbone.config)

    @torch.no_grad()
    def encode_ids(self, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        h = self.backbone(input_ids=input_ids, attention_mask=attention_mask).last_hidden_state
        m = attention_mask.unsqueeze(-1).to(h.dtype)
        return (h * m).sum(1) / m.sum(1).clamp_min(1e-9)   # sentence-transformers Pooling(mean)

    @torch.no_grad()
    def encode_ids_fused(self, input_ids: torch.Tensor, lengths, n_buckets: int = 8) -> torch.Tensor:
        """Same embeddings as encode_ids (to fp32 rounding), through FusedBertForward."""
        if getattr(self, "_fused", None) is None:
            self._fused = FusedBertForward(self.backbone)
        return self._fused(input_ids, lengths, n_buckets)

    @torch.no_grad()
    def encode_ids_packed(self, input_ids: torch.Tensor, lengths, mark=None) -> torch.Tensor:
        """Same embeddings again, padding-free (PackedBertForward: HIP attention / LayerNorm / pooling kernels)."""
        if getattr(self, "_packed", None) is None:
            self._packed = PackedBertForward(self.backbone)
        return self._packed(input_ids, lengths, mark=mark)   # raises for head_dim != 64

    @torch.no_grad()
    def encode_ids_corpus(self, input_ids: torch.Tensor, lengths) -> torch.Tensor:
        """encode_ids_packed for more sequences than one forward should hold: sub-batches of at most `packed_tokens` token rows."""
        lengths = np.asarray(lengths)     # (a Python list of token counts is as good as an array)
        out = torch.empty((input_ids.shape[0], self.dim), dtype=torch.float32, device=self._device)
        for idx in _id_batches(lengths, self.packed_tokens):
            sel = torch.from_numpy(idx).to(self._device)
            out[sel] = self.encode_ids_packed(input_ids.index_select(0, sel), lengths[idx])
        return out

    @torch.no_grad()
    def encode_ids_bucketed(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, lengths, n_buckets: int = 8) -> torch.Tensor:
        """Same result as encode_ids, without paying for padding: sequences are sorted by length (as
        SentenceTransformer.encode does, hybrid.py:101-102) and run in `n_buckets` sub-batches, each trimmed to its own
        longest sequence.  `lengths` is the HOST array of token counts (known from tokenisation: no device sync)."""
        import numpy as np
        lengths = np.asarray(lengths)
        n = len(lengths)
        order = np.argsort(-lengths, kind="stable")
        out = torch.empty((n, self.dim), dtype=torch.float32, device=self._device)
        per = -(-n // max(1, n_buckets))
        for s in range(0, n, per):
            idx = order[s: s + per]
            L = int(lengths[idx[0]])
            sel = torch.from_numpy(idx).to(self._device, non_blocking=True)
            ids = input_ids.index_select(0, sel)[:, :L]
            mask = attention_mask.index_select(0, sel)[:, :L]
            out.index_copy_(0, sel, self.encode_ids(ids, mask).float())
        return out

    @torch.no_grad()
    def encode(self, sentences: list[str], batch_size: int = 64, query_mode: bool = True, **_) -> torch.Tensor:
        out = torch.empty((len(sentences), self.dim), dtype=torch.float32, device=self._device)
        fwd = self._packed_forward(self.backbone)
        if fwd is not None:
            for idx, ids, lens in _token_batches(self, sentences, self.max_doc_length, batch_size):
                out[torch.tensor(idx, device=self._device)] = fwd(ids.to(self._device, non_blocking=True), lens)
            return out
        for idx, ids, mask in self._batches(sentences, batch_size, self.max_doc_length):   # max_seq_length = 512 (hybrid.py:99)
            out[torch.tensor(idx, device=self._device)] = self.encode_ids(ids, mask).float()
        return out


class SpladeEncoder(_Base):
    """SPLADE-max: amax over tokens of log1p(relu(MLM logits * mask))  (splade/splade.py:88-99)."""

    def __init__(self, mlm, tokenizer, device):
        super().__init__(tokenizer, device)
        self.mlm = mlm.to(self._device).eval()
        self.dim = mlm.config.vocab_size
        self._clamp_lengths(mlm.config)

    @torch.no_grad()
    def encode_ids(self, input_ids, attention_mask) -> torch.Tensor:
        logits = self.mlm(input_ids=input_ids, attention_mask=attention_mask).logits
        return torch.amax(torch.log1p(torch.relu(logits * attention_mask.unsqueeze(-1))), dim=1)

    @torch.no_grad()
    def calibrate_sparsity(self, per_token: float = 3.5e-5, seed: int = 0):
        """Synthetic runs only: shift the decoder bias of a RANDOM-INIT head so that a token's logit is positive with probability `per_token`
        -- a trained SPLADE's FLOPS regulariser leaves a few dozen active terms per query and a few hundred per document (3.5e-5 x 36 / 300
        tokens of 32,005 terms ~ 40 / 330); a random head activates half the vocabulary.  Measured on one random batch."""
        g = torch.Generator().manual_seed(seed)
        ids = torch.randint(7, self.dim - 1, (8, 64), generator=g).to(self._device)
        logits = self.mlm(input_ids=ids, attention_mask=torch.ones_like(ids)).logits.float()
        z = float(torch.distributions.Normal(0.0, 1.0).icdf(torch.tensor(1.0 - per_token)))
        bias = self.mlm.lm_head.decoder.bias
        bias -= logits.mean() + z * logits.std()
        return self

    HEAD_TOKENS = 65536     # token rows per forward + head pass (the fused head writes no [T, vocab] logits: 16384 rows of them were 2.1 GB)
    FUSED_HEAD = True       # False: decoder GEMM -> [T, vocab] logits -> fz_segment_splade_max_f32 (the reference's shape, splade.py:94)

    @torch.no_grad()
    def _pool_packed(self, x: torch.Tensor, cu_d: torch.Tensor, mark=None) -> torch.Tensor:
        """MLM head + SPLADE-max pooling over packed hidden rows x [T, hidden] of the sequences cu_d [n+1] -> [n, vocab].
        With a RoBERTa-style head (dense -> GELU -> LayerNorm -> decoder) the vocabulary projection runs as fz_splade_head_max_f32: the
        pooling is the GEMM's epilogue and the [T, vocab] logits -- 4.2 GB per batch of 64 x 512 tokens in the reference, splade.py:94,
        ~1 TB over the LLeQA corpus -- are never written."""
        from . import ops
        head = self.mlm.lm_head
        fused = self.FUSED_HEAD and all(hasattr(head, n) for n in ("dense", "layer_norm", "decoder")) and x.shape[0] > 0
        if not fused:
            out = None
            for r0 in range(0, max(x.shape[0], 1), 16384):       # the logits plane in slices of 16384 rows (2.1 GB)
                cu = (cu_d.clamp(r0, min(r0 + 16384, x.shape[0])) - r0).int()
                part = ops.segment_splade_max(head(x[r0: r0 + 16384]), cu)
                out = part if out is None else torch.maximum(out, part)
            if mark: mark("splade_head_pool")
            return out
        h = torch.nn.functional.linear(x, head.dense.weight, head.dense.bias)
        h = ops.gelu_(h) if h.numel() % 4 == 0 and h.is_contiguous() else torch.nn.functional.gelu(h)
        h = ops.add_layernorm(h, None, head.layer_norm.weight, head.layer_norm.bias, head.layer_norm.eps)
        bias = head.decoder.bias if head.decoder.bias is not None else torch.zeros(head.decoder.weight.shape[0], device=x.device)
        out = ops.splade_head_max(h, head.decoder.weight, bias.contiguous(), cu_d)
        if mark: mark("splade_head_pool")
        return out

    @torch.no_grad()
    def encode_ids_packed(self, input_ids: torch.Tensor, lengths, mark=None) -> torch.Tensor:
        """Already tokenised sequences [n, Lmax] (+ HOST token counts) -> SPLADE vectors [n, vocab] fp32, padding-free; sub-batches of
        at most HEAD_TOKENS token rows."""
        fwd = self._packed_forward(getattr(self.mlm, self.mlm.base_model_prefix))
        if fwd is None:
            raise RuntimeError("SpladeEncoder.encode_ids_packed needs the padding-free forward (GPU, 64-wide heads)")
        lengths = np.asarray(lengths)
        out = torch.empty((input_ids.shape[0], self.dim), dtype=torch.float32, device=self._device)
        for idx in _id_batches(lengths, self.HEAD_TOKENS):
            sel = torch.from_numpy(idx).to(self._device)
            x, cu_d = fwd.hidden(input_ids.index_select(0, sel), lengths[idx], mark)
            out[sel] = self._pool_packed(x, cu_d, mark)
        return out

    @torch.no_grad()
    def encode(self, sentences, batch_size: int = 64, query_mode: bool = True, **_) -> torch.Tensor:
        out = torch.empty((len(sentences), self.dim), dtype=torch.float32, device=self._device)
        max_len = self.max_query_length if query_mode else self.max_doc_length
        fwd = self._packed_forward(getattr(self.mlm, self.mlm.base_model_prefix))
        if fwd is not None:
            self.packed_tokens = min(self.packed_tokens, self.HEAD_TOKENS)
            for idx, ids, lens in _token_batches(self, sentences, max_len, batch_size):
                x, cu_d = fwd.hidden(ids.to(self._device, non_blocking=True), lens)
                out[torch.tensor(idx, device=self._device)] = self._pool_packed(x, cu_d)
            return out
        for idx, ids, mask in self._batches(sentences, batch_size, max_len):
            out[torch.tensor(idx, device=self._device)] = self.encode_ids(ids, mask).float()
        return out


class ColbertEncoder(_Base):
    """ColBERT: CamemBERT -> Linear(768,128, bias=False) -> L2-normalised token vectors (fp16).

    Tokenisation follows colbert-ai (requirements.txt:15, absent from the tree; its published QueryTokenizer / DocTokenizer):
    the text is prefixed with ". " and the token after <s> is then overwritten with the [Q] / [D] marker id
    (`tokenizer.convert_tokens_to_ids("[unused0]" / "[unused1]")`: the unknown-token id in a CamemBERT vocabulary); queries are
    padded to query_maxlen with the mask token and -- `--attend_to_mask_tokens`, run_colbert.sh:29 -- attended; document tokens
    whose id is in the punctuation skiplist are dropped (`--mask_punctuation`, run_colbert.sh:28).  `q_marker_id` /
    `d_marker_id` None (synthetic HashTokenizer runs): no marker, no prefix."""
    dim = 128

    def __init__(self, backbone, tokenizer, device, punct_ids: tuple[int, ...] = (), linear_weight: torch.Tensor | None = None,
                 q_marker_id: int | None = None, d_marker_id: int | None = None, attend_to_mask_tokens: bool = True, amp: bool = True):
        super().__init__(tokenizer, device)
        self.backbone = backbone.to(self._device).eval()
        # colbert-ai encodes under mixed precision: Checkpoint wraps query() / doc() in its MixedPrecisionManager's autocast context, and the
        # repository's ColBERT runs set 'amp': True (multi_dense_biencoder.py:55; colbert_ir.py:110,124 for the training forward).  On a GPU
        # the Linears of the backbone therefore take float16 operands (fp32 accumulate); amp=False keeps the whole forward in float32.
        self.amp = bool(amp) and self._device.type == "cuda"
        if linear_weight is not None:
            self.dim = int(linear_weight.shape[0])
        self.linear = nn.Linear(backbone.config.hidden_size, self.dim, bias=False).to(self._device)
        if linear_weight is not None:     # the checkpoint's projection (colbert-ai saves it as `linear.weight`)
            if tuple(linear_weight.shape) != (self.dim, backbone.config.hidden_size):
                raise ValueError(f"ColBERT projection {tuple(linear_weight.shape)} does not fit hidden size {backbone.config.hidden_size}")
            with torch.no_grad():
                self.linear.weight.copy_(linear_weight.to(self._device, torch.float32))
        self._clamp_lengths(backbone.config)
        self.punct_ids = torch.tensor(list(punct_ids), dtype=torch.long, device=self._device)
        self.q_marker_id, self.d_marker_id, self.attend_to_mask_tokens = q_marker_id, d_marker_id, attend_to_mask_tokens

    def _packed_forward(self, backbone):
        fwd = super()._packed_forward(backbone)
        if fwd is not None:
            fwd.amp_dtype = torch.float16 if self.amp else None
        return fwd

    def _marked(self, texts: list[str], max_len: int, marker_id: int | None, pad_to_max: bool):
        """colbert-ai tensorize(): '. ' + text, tokenise, ids[:, 1] = marker."""
        if marker_id is None:
            return self.tokenizer(texts, max_len, pad_to_max)
        ids, mask = self.tokenizer([". " + t for t in texts], max_len, pad_to_max)
        ids[:, 1] = marker_id
        return ids, mask

    def _project(self, x: torch.Tensor) -> torch.Tensor:
        """The 128-d projection of packed hidden rows, in the precision the HF route (_tokens) runs it: under colbert-ai's autocast
        the Linear takes and returns float16 (float32 accumulate); the normalisation that follows is float32 either way."""
        if self.amp:
            return torch.nn.functional.linear(x.half(), self.linear.weight.half()).float()
        return self.linear(x)

    @torch.no_grad()
    def _tokens(self, ids, mask):
        with torch.autocast(self._device.type, dtype=torch.float16, enabled=self.amp):      # the HF module as colbert-ai runs it
            h = self.backbone(input_ids=ids, attention_mask=mask).last_hidden_state
            v = self.linear(h)
        return torch.nn.functional.normalize(v.float(), p=2, dim=-1)

    @torch.no_grad()
    def encode_queries(self, queries: list[str], batch_size: int = 64) -> torch.Tensor:
        """-> [Q, query_maxlen, dim] fp16; pads with the mask token and attends to it (run_colbert.sh:29)."""
        import numpy as np
        from . import ops
        Lq = self.max_query_length
        out = torch.empty((len(queries), Lq, self.dim), dtype=torch.float16, device=self._device)
        fwd = self._packed_forward(self.backbone) if self.attend_to_mask_tokens else None   # every query is Lq attended tokens: already "packed"
        step = batch_size if fwd is None else max(batch_size, self.packed_tokens // Lq)
        for s0 in range(0, len(queries), step):
            part = queries[s0: s0 + step]
            ids, mask = self._marked(part, Lq, self.q_marker_id, True)
            ids = torch.where(mask.bool(), ids, torch.full_like(ids, self.tokenizer.mask_token_id)).to(self._device, non_blocking=True)
            if fwd is not None:
                x, _ = fwd.hidden(ids, np.full(len(part), Lq))
                out[s0: s0 + len(part)] = ops.normalize_rows(self._project(x)).view(len(part), Lq, self.dim).half()
            else:
                # colbert-ai multiplies Q by (ids != pad), all ones once the padding is the mask token: every row is kept
                att = torch.ones_like(ids) if self.attend_to_mask_tokens else mask.to(self._device)
                out[s0: s0 + len(part)] = self._tokens(ids, att).half()
        return out

    @torch.no_grad()
    def encode_query_ids(self, ids: torch.Tensor, mark=None) -> torch.Tensor:
        """Already tokenised queries [Q, query_maxlen] (mask-token padded, all attended) -> [Q, query_maxlen, dim] fp16."""
        import numpy as np
        from . import ops
        fwd = self._packed_forward(self.backbone)
        if fwd is None:
            raise RuntimeError("ColbertEncoder.encode_query_ids needs the padding-free forward (GPU, 64-wide heads)")
        Q, Lq = ids.shape
        x, _ = fwd.hidden(ids, np.full(Q, Lq), mark)
        out = ops.normalize_rows(self._project(x)).view(Q, Lq, self.dim).half()
        if mark: mark("colbert_project")
        return out

    @torch.no_grad()
    def encode_doc_ids(self, input_ids: torch.Tensor, lengths):
        """Already tokenised documents [n, Lmax] (+ HOST token counts) -> (Dtok [sumL, dim] fp16 packed in input order, Doff [n+1] int64);
        no punctuation skiplist on this path (ids only: synthetic corpora)."""
        import numpy as np
        from . import ops
        fwd = self._packed_forward(self.backbone)
        if fwd is None:
            raise RuntimeError("ColbertEncoder.encode_doc_ids needs the padding-free forward (GPU, 64-wide heads)")
        lengths = np.minimum(np.asarray(lengths, dtype=np.int64), input_ids.shape[1])
        off = np.zeros(len(lengths) + 1, dtype=np.int64)
        np.cumsum(lengths, out=off[1:])
        tok = torch.empty((int(off[-1]), self.dim), dtype=torch.float16, device=self._device)
        for idx in _id_batches(lengths, self.packed_tokens):
            sel = torch.from_numpy(idx).to(self._device)
            x, _ = fwd.hidden(input_ids.index_select(0, sel), lengths[idx])
            v = ops.normalize_rows(self._project(x)).half()
            rows = np.concatenate([np.arange(off[i], off[i + 1]) for i in idx]) if len(idx) else np.zeros(0, dtype=np.int64)
            tok[torch.from_numpy(rows).to(self._device)] = v
        return tok, torch.from_numpy(off).to(self._device)

    @torch.no_grad()
    def encode_docs(self, docs: list[str], batch_size: int = 64):
        """-> (Dtok [sumL,128] fp16 packed in corpus order, Doff [N+1] int64): padding and punctuation tokens dropped."""
        fwd = self._packed_forward(self.backbone)
        if fwd is not None:
            return self._encode_docs_packed(fwd, docs, batch_size)
        per_doc: list[torch.Tensor | None] = [None] * len(docs)
        order = sorted(range(len(docs)), key=lambda i: -len(docs[i]))
        for s0 in range(0, len(order), batch_size):
            idx = order[s0: s0 + batch_size]
            ids, mask = self._marked([docs[i] for i in idx], self.max_doc_length, self.d_marker_id, False)
            ids, mask = ids.to(self._device), mask.to(self._device)
            v = self._tokens(ids, mask).half()
            keep = mask.bool()
            if self.punct_ids.numel():
                keep &= ~torch.isin(ids, self.punct_ids)
            for r, i in enumerate(idx):
                per_doc[i] = v[r][keep[r]]
        lens = torch.tensor([t.shape[0] for t in per_doc], dtype=torch.int64)
        off = torch.zeros(len(docs) + 1, dtype=torch.int64)
        off[1:] = torch.cumsum(lens, 0)
        tok = torch.cat(per_doc, 0) if per_doc else torch.empty((0, self.dim), dtype=torch.float16, device=self._device)
        return tok.contiguous(), off.to(self._device)

    @torch.no_grad()
    def _encode_docs_packed(self, fwd, docs, batch_size):
        """Padding-free document side: the packed token rows of the forward ARE the ragged layout fz_maxsim_f16 reads; punctuation
        rows are dropped with a gather whose indices come from the HOST copy of the ids (no device sync), and the sub-batches
        (longest documents first) are stitched back into corpus order by one final gather."""
        import numpy as np
        from . import ops
        punct = self.punct_ids.cpu().numpy()
        chunks, counts, place = [], np.zeros(len(docs), dtype=np.int64), []
        base = 0
        for idx, ids, lens in _token_batches(self, docs, self.max_doc_length, batch_size,
                                             tokenize=lambda texts, L: self._marked(texts, L, self.d_marker_id, False)):
            x, _ = fwd.hidden(ids.to(self._device, non_blocking=True), lens)
            v = ops.normalize_rows(self._project(x)).half()                               # [T, 128], rows in sub-batch order
            ids_np = ids.numpy()
            keep_rows, starts = [], np.zeros(len(idx) + 1, dtype=np.int64)
            np.cumsum(lens, out=starts[1:])
            for r, i in enumerate(idx):
                k = np.flatnonzero(~np.isin(ids_np[r, : lens[r]], punct)) if punct.size else np.arange(lens[r])
                counts[i] = len(k)
                keep_rows.append(k + starts[r])
                place.append((i, base, len(k)))
                base += len(k)
            keep = np.concatenate(keep_rows) if keep_rows else np.zeros(0, dtype=np.int64)
            chunks.append(v if len(keep) == v.shape[0] else v[torch.from_numpy(keep).to(self._device, non_blocking=True)])
        off = np.zeros(len(docs) + 1, dtype=np.int64)
        np.cumsum(counts, out=off[1:])
        gather = np.empty(int(off[-1]), dtype=np.int64)
        for i, b, c in place:
            gather[off[i]: off[i] + c] = np.arange(b, b + c)
        tok = torch.cat(chunks, 0) if chunks else torch.empty((0, self.dim), dtype=torch.float16, device=self._device)
        tok = tok[torch.from_numpy(gather).to(self._device, non_blocking=True)] if len(gather) else tok
        return tok.contiguous(), torch.from_numpy(off).to(self._device)


def random_init(kind: str, device="cuda", size: str = "base", seed: int = 0, tokenizer: str = "hash"):
    """CamemBERT-base-shaped (or tiny) encoder with random weights: `kind` in {'dpr','splade','colbert'}.
    tokenizer: 'hash' (whitespace words hashed into the vocabulary) or 'synth-fr' (base size only: the 32,005-piece BPE of
    tokenization.SynthFrenchTokenizer -- real sub-word work on the host)."""
    cfg = dict(CAMEMBERT_BASE if size == "base" else TINY)
    g = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        if tokenizer == "synth-fr":
            from .tokenization import SynthFrenchTokenizer
            tok = SynthFrenchTokenizer()
            if tok.vocab_size != cfg["vocab_size"]:
                raise ValueError(f"the synthetic tokenizer has {tok.vocab_size} pieces, the {size} model {cfg['vocab_size']}")
        else:
            tok = HashTokenizer(cfg["vocab_size"], cfg["pad_token_id"], cfg["bos_token_id"], cfg["eos_token_id"])
        if kind == "dpr":
            return DenseEncoder(_backbone(cfg), tok, device)
        if kind == "splade":
            return SpladeEncoder(_backbone(cfg, mlm=True), tok, device)
        if kind == "colbert":
            return ColbertEncoder(_backbone(cfg), tok, device)
        raise ValueError(kind)
    finally:
        torch.random.set_rng_state(g)


def _checkpoint_tensor(model_name_or_path: str, name: str) -> torch.Tensor | None:
    """One tensor of a local checkpoint directory (model.safetensors or pytorch_model.bin), None when it is not there."""
    st = os.path.join(model_name_or_path, "model.safetensors")
    if os.path.exists(st):
        from safetensors import safe_open
        with safe_open(st, framework="pt", device="cpu") as f:
            return f.get_tensor(name) if name in f.keys() else None
    pt = os.path.join(model_name_or_path, "pytorch_model.bin")
    if os.path.exists(pt):
        return torch.load(pt, map_location="cpu", weights_only=True).get(name)
    return None


def from_pretrained(model_name_or_path: str, kind: str, device="cuda"):
    """Load a real checkpoint from a local directory / HF cache (no network on the build boxes).
    kind: 'dpr' (AutoModel + mean pooling), 'splade' (AutoModelForMaskedLM), 'colbert' (colbert-ai layout: base model +
    `linear.weight` + artifact.metadata), 'monobert' (AutoModelForSequenceClassification, hybrid.py:139-163)."""
    import json
    import string
    from transformers import AutoModel, AutoModelForMaskedLM, AutoModelForSequenceClassification, AutoTokenizer
    hf_tok = AutoTokenizer.from_pretrained(model_name_or_path, local_files_only=True)

    class _Tok:
        pad_token_id, mask_token_id = hf_tok.pad_token_id, hf_tok.mask_token_id

        def __call__(self, texts, max_length, pad_to_max=False):
            e = hf_tok(texts, padding="max_length" if pad_to_max else True, truncation=True, max_length=max_length, return_tensors="pt")
            return e["input_ids"], e["attention_mask"]

        def encode_pairs(self, first, second, max_length):   # "<s> query </s></s> document </s>", truncation as CrossEncoder.smart_batching_collate
            e = hf_tok(first, second, padding=True, truncation="longest_first", max_length=max_length, return_tensors="pt")
            return e["input_ids"], e["attention_mask"]
    if kind == "dpr":
        return DenseEncoder(AutoModel.from_pretrained(model_name_or_path, local_files_only=True), _Tok(), device)
    if kind == "splade":
        return SpladeEncoder(AutoModelForMaskedLM.from_pretrained(model_name_or_path, local_files_only=True), _Tok(), device)
    if kind == "colbert":
        meta = {}
        mp = os.path.join(model_name_or_path, "artifact.metadata")       # colbert-ai's ColBERTConfig dump
        if os.path.exists(mp):
            with open(mp) as f:
                meta = json.load(f)
        lin = _checkpoint_tensor(model_name_or_path, "linear.weight")
        if lin is None:
            raise FileNotFoundError(f"{model_name_or_path}: no `linear.weight` -- not a ColBERT checkpoint (colbert-ai saves the 768 -> dim "
                                    "projection under that name next to the base model)")
        punct = ()
        if meta.get("mask_punctuation", True):                            # run_colbert.sh:28; colbert-ai's skiplist: first sub-token of each symbol
            punct = tuple(sorted({hf_tok.encode(sym, add_special_tokens=False)[0] for sym in string.punctuation
                                  if hf_tok.encode(sym, add_special_tokens=False)}))
        enc = ColbertEncoder(AutoModel.from_pretrained(model_name_or_path, local_files_only=True), _Tok(), device, punct_ids=punct,
                             linear_weight=lin,
                             q_marker_id=hf_tok.convert_tokens_to_ids(meta.get("query_token_id", "[unused0]")),
                             d_marker_id=hf_tok.convert_tokens_to_ids(meta.get("doc_token_id", "[unused1]")),
                             attend_to_mask_tokens=bool(meta.get("attend_to_mask_tokens", True)), amp=bool(meta.get("amp", True)))
        enc.max_query_length = min(enc.max_query_length, int(meta.get("query_maxlen", enc.max_query_length)))   # hybrid.py:129 passes 64 / 512
        enc.max_doc_length = min(enc.max_doc_length, int(meta.get("doc_maxlen", enc.max_doc_length)))
        return enc
    if kind == "monobert":
        model = AutoModelForSequenceClassification.from_pretrained(model_name_or_path, local_files_only=True)
        act = getattr(model.config, "sbert_ce_default_activation_function", None)   # CrossEncoderCustom.__init__ (sentence_transformers.py:546-551)
        if act is None:
            act = "sigmoid" if model.config.num_labels == 1 else "identity"
        elif "Sigmoid" in act:
            act = "sigmoid"
        elif "Identity" in act:
            act = "identity"
        else:
            raise ValueError(f"cross-encoder activation {act!r} is not supported")
        return CrossEncoder(model, _Tok(), device, activation=act)
    raise ValueError(kind)


class CrossEncoder(_Base):
    """monoBERT reranker (hybrid.py:139-163, CrossEncoderCustom): CamemBERT sequence classifier over "<s> query </s></s> doc </s>",
    one logit per pair, fp32.  `predict(pairs)` is what Ranker.cross_encoder_search calls."""

    def __init__(self, classifier, tokenizer, device, max_length: int = 512, activation: str = "identity"):
        super().__init__(tokenizer, device)
        self.model = classifier.to(self._device).eval()
        self._clamp_lengths(classifier.config)
        self.max_length = min(max_length, self.max_doc_length)
        self.activation = activation     # CrossEncoder.predict applies the default activation: Sigmoid for one label (ST 2.2.2)

    def _pair_ids(self, pairs, max_len):
        """(query, document) pairs -> ids, mask.  A real tokenizer encodes the PAIR ("<s> q </s></s> d </s>", longest-first
        truncation); the synthetic HashTokenizer has no pair API: the separator is an ordinary token there."""
        if hasattr(self.tokenizer, "encode_pairs"):
            return self.tokenizer.encode_pairs([q for q, _ in pairs], [d for _, d in pairs], max_len)
        return self.tokenizer([q + " </s> " + d for q, d in pairs], max_len)

    @torch.no_grad()
    def predict(self, pairs: list[tuple[str, str]], batch_size: int = 64) -> torch.Tensor:
        out = torch.empty(len(pairs), dtype=torch.float32, device=self._device)
        texts = [q + " " + d for q, d in pairs]          # only for the length-sorted batching
        base = getattr(self.model, self.model.base_model_prefix)
        fwd = self._packed_forward(base) if hasattr(self.model, "classifier") else None
        order = sorted(range(len(pairs)), key=lambda i: -len(texts[i]))
        s0 = 0
        while s0 < len(order):
            step = batch_size if fwd is None else max(batch_size, self.packed_tokens // min(self.max_length, 8 + 2 * len(texts[order[s0]].split())))
            idx = order[s0: s0 + step]
            s0 += step
            ids, mask = self._pair_ids([pairs[i] for i in idx], self.max_length)
            if fwd is not None:
                # padding-free forward; the classification head reads the first token (<s>) of every pair: features[:, 0, :]
                x, cu_d = fwd.hidden(ids.to(self._device, non_blocking=True), mask.sum(1).numpy())
                first = x.index_select(0, cu_d[:-1].long())
                logits = self.model.classifier(first.unsqueeze(1)).reshape(len(idx), -1)
            else:
                logits = self.model(input_ids=ids.to(self._device), attention_mask=mask.to(self._device)).logits
            if logits.shape[1] != 1:
                raise ValueError(f"cross-encoder with {logits.shape[1]} labels: the rerank stage expects one relevance logit (hybrid.py:159)")
            score = logits[:, 0].float()                     # one label: score[0] (CrossEncoder.predict)
            out[torch.tensor(idx, device=self._device)] = torch.sigmoid(score) if self.activation == "sigmoid" else score
        return out


def random_cross_encoder(device="cuda", size: str = "tiny", seed: int = 0):
    from transformers import CamembertConfig, CamembertForSequenceClassification
    cfg = dict(CAMEMBERT_BASE if size == "base" else TINY)
    g = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        model = CamembertForSequenceClassification(CamembertConfig(num_labels=1, **cfg))
        tok = HashTokenizer(cfg["vocab_size"], cfg["pad_token_id"], cfg["bos_token_id"], cfg["eos_token_id"])
        return CrossEncoder(model, tok, device)
    finally:
        torch.random.set_rng_state(g)


# ---------------------------------------------------------------------------------------------------------------
# N2: corpus-side cache.  run_hybrid.sh starts one process per (combo, fusion, normalisation): 99 processes that each
# re-encode the same 27.9 k articles with the same checkpoint (hybrid.py:101, run_hybrid.sh:43).  The encoded corpus is
# a pure function of (checkpoint, corpus text): keep it on disk, keyed by both, and memory-map it back.
# ---------------------------------------------------------------------------------------------------------------
def corpus_cache_key(model_name_or_path: str, documents: list[str], extra: str = "") -> str:
    h = hashlib.blake2b(digest_size=16)
    h.update(model_name_or_path.encode()); h.update(extra.encode()); h.update(str(len(documents)).encode())
    for d in documents:
        h.update(hashlib.blake2b(d.encode("utf-8"), digest_size=8).digest())
    return h.hexdigest()


def cached_tensors(cache_dir: str | None, key: str, names: list[str], compute):
    """Return `compute()`'s tensors (tuple, CPU or GPU), loading them from / saving them to `cache_dir/key.*.npy`."""
    import os
    import numpy as np
    if cache_dir is None:
        return compute()
    paths = [os.path.join(cache_dir, f"{key}.{n}.npy") for n in names]
    if all(os.path.exists(p) for p in paths):
        return tuple(torch.from_numpy(np.array(np.load(p, mmap_mode="r"))) for p in paths)   # memory-mapped read
    out = compute()
    os.makedirs(cache_dir, exist_ok=True)
    for p, t in zip(paths, out):
        tmp = p + f".tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            np.save(f, t.detach().cpu().numpy())
        os.replace(tmp, p)       # atomic: concurrent sweep processes never see a half-written file
    return out
