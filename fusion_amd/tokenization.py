"""Text -> token ids on the host, and the overlap of that work with the device.

The reference's encode starts from strings: `model.encode(queries)` (hybrid.py:101-102) tokenises inside sentence-transformers,
`BaseModel.encode` tokenises per batch (splade/base.py:142-171, 274-291).  Offline there is no CamemBERT SentencePiece file, so
`SynthFrenchTokenizer` loads a 32,005-piece BPE of camembert's layout trained on synthetic French-like text (tools/
train_synth_tokenizer.py -> tuned/synth_fr_tokenizer.json.gz; the Rust `tokenizers` library -- an OPTIONAL dependency, listed in
requirements.txt, needed only by this class -- which releases the GIL and uses every host core in encode_batch) -- real sub-word work of the right size, so that what text -> ids costs can be MEASURED and hidden:
`prefetch()` runs a batch generator one step ahead on a host thread, i.e. batch i + 1 is tokenised while the GPU runs batch i."""
from __future__ import annotations

import gzip
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SYNTH_FR = os.path.join(_HERE, "tuned", "synth_fr_tokenizer.json.gz")


def cap_host_threads(n: int | None = None, override: bool = False) -> int:
    """Size tokenizers' rayon pool: sets RAYON_NUM_THREADS for this process and returns the value in force.  PROCESS-GLOBAL, and read by
    rayon only when its pool is first used -- so this is for the program's entry point (bench.py, a serving loop) to call once before the
    first encode_batch, not for a library constructor: it caps every user of `tokenizers` in the process.  Why cap at all: encode_batch
    fans out over a pool sized to EVERY core the machine shows (128-256 on the GPU hosts, of which a job owns a share); left alone it
    crowds out the thread that launches the GPU work it is meant to hide behind (63.7 instead of 49.5 ms per step).  A handful
    of workers tokenise a 1024-query batch in a fraction of a device step.  n=None: min(8, half the visible cores); an existing setting wins
    unless override."""
    if n is None:
        n = max(1, min(8, (os.cpu_count() or 2) // 2))
    if override or "RAYON_NUM_THREADS" not in os.environ:
        os.environ["RAYON_NUM_THREADS"] = str(int(n))
    return int(os.environ["RAYON_NUM_THREADS"])


class SynthFrenchTokenizer:
    """Callable like encoders.HashTokenizer / the HF wrapper: (texts, max_length, pad_to_max) -> (ids [B, L] int64, mask [B, L] int64)
    host tensors, "<s> pieces </s>" truncated to max_length, padded with <pad> = 1."""

    def __init__(self, path: str = SYNTH_FR, threads: int | None = None):
        """threads: cap tokenizers' rayon pool for THIS PROCESS (see cap_host_threads) -- an explicit request; the constructor by itself
        changes no process-global state."""
        if threads is not None:
            cap_host_threads(threads, override=True)
        from tokenizers import Tokenizer
        with gzip.open(path, "rb") as f:
            self._tok = Tokenizer.from_str(f.read().decode("utf-8"))
        self.vocab_size = self._tok.get_vocab_size()
        self.pad_token_id = self._tok.token_to_id("<pad>")
        self.mask_token_id = self._tok.token_to_id("<mask>")
        self.bos_id, self.eos_id = self._tok.token_to_id("<s>"), self._tok.token_to_id("</s>")
        self._trunc = None

    def encode_np(self, texts: list[str], max_length: int, pad_to_max: bool = False):
        """-> (ids [B, L] int64, lengths [B] int64) numpy; L = max_length (pad_to_max) or the batch's longest row."""
        if self._trunc != max_length:
            self._tok.enable_truncation(max_length=max_length)
            self._trunc = max_length
        batch = getattr(self._tok, "encode_batch_fast", self._tok.encode_batch)    # (the _fast form skips the offsets nobody reads here)
        enc = batch(list(texts))                                  # Rust, GIL released, RAYON_NUM_THREADS workers
        lens = np.fromiter((len(e.ids) for e in enc), dtype=np.int64, count=len(enc))
        L = max_length if pad_to_max else int(lens.max(initial=1))
        ids = np.full((len(enc), L), self.pad_token_id, dtype=np.int64)
        for i, e in enumerate(enc):
            ids[i, : lens[i]] = e.ids
        return ids, lens

    def __call__(self, texts: list[str], max_length: int, pad_to_max: bool = False):
        ids, lens = self.encode_np(texts, max_length, pad_to_max)
        mask = (np.arange(ids.shape[1])[None, :] < lens[:, None]).astype(np.int64)
        return torch.from_numpy(ids), torch.from_numpy(mask)


def prefetch(batches, depth: int = 1):
    """Iterate `batches` (any iterable whose items are EXPENSIVE TO PRODUCE ON THE HOST: tokenised sub-batches) with the production of
    the next `depth` items running on a worker thread while the caller consumes the current one (launches its GPU work).  Same items,
    same order as plain iteration; an exception raised while producing an item is re-raised at the point where that item is due.
    Only the worker thread advances the iterator."""
    it = iter(batches)
    done = object()
    ex = ThreadPoolExecutor(max_workers=1, thread_name_prefix="fusion-amd-prefetch")
    try:
        futs = [ex.submit(next, it, done) for _ in range(max(1, depth))]
        while futs:
            item = futs.pop(0).result()
            if item is done:
                break
            futs.append(ex.submit(next, it, done))
            yield item
    finally:
        ex.shutdown(wait=True, cancel_futures=True)
