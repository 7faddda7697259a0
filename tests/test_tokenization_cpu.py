"""Host side of 'encode starts from strings' (hybrid.py:101-102; splade/base.py:142-171): the synthetic-French sub-word tokenizer and the
prefetch that tokenises sub-batch i + 1 on a host thread while sub-batch i runs -- same items, same order, same embeddings as the serial
loop.  No GPU."""
import threading
import time

import numpy as np
import pytest
import torch

pytest.importorskip("tokenizers", reason="the synthetic-French BPE needs the optional `tokenizers` wheel (requirements.txt)")

from fusion_amd import encoders
from fusion_amd.synth_text import FrenchLike
from fusion_amd.tokenization import SynthFrenchTokenizer, prefetch


@pytest.fixture(scope="module")
def tok():
    return SynthFrenchTokenizer()


def test_tokenizer_has_camemberts_layout(tok):
    assert tok.vocab_size == encoders.CAMEMBERT_BASE["vocab_size"] == 32005
    assert (tok.pad_token_id, tok.bos_id, tok.eos_id, tok.mask_token_id) == (1, 5, 6, encoders.MASK_TOKEN_ID)


def test_tokenizer_output_shape_truncation_padding(tok):
    rng = np.random.default_rng(0)
    texts = FrenchLike().sentences(rng, 64, 1, 120, question=True) + [""]
    ids, mask = tok(texts, 64)
    assert ids.dtype == torch.int64 and ids.shape == mask.shape and ids.shape[1] == 64          # some sentence fills the budget
    lens = mask.sum(1)
    assert int(lens.max()) == 64 and int(lens.min()) == 2                                     # "<s> </s>" for the empty string
    for i in range(len(texts)):
        L = int(lens[i])
        assert ids[i, 0] == tok.bos_id and ids[i, L - 1] == tok.eos_id and bool((ids[i, L:] == tok.pad_token_id).all())
        assert bool((mask[i, :L] == 1).all()) and int(ids[i].max()) < tok.vocab_size
    ids2, _ = tok(texts, 64, pad_to_max=True)
    assert ids2.shape == (len(texts), 64)
    short, _ = tok(["le la"], 64)
    assert short.shape[1] <= 6                                                                 # padded to the batch's longest row only
    a, la = tok.encode_np(texts, 32)
    assert a.shape[1] == 32 and la.max() == 32 and np.array_equal(a[:, 0], np.full(len(texts), tok.bos_id))


def test_subword_statistics_are_realistic(tok):
    """Frequent words are single pieces, rare ones split: about 1.2-1.4 pieces per word (CamemBERT on French text: ~1.3)."""
    rng = np.random.default_rng(3)
    texts = FrenchLike().sentences(rng, 512, 5, 45, question=True)
    _, lens = tok.encode_np(texts, 512)
    ratio = (lens - 2).sum() / sum(len(t.split()) for t in texts)
    assert 1.1 < ratio < 1.6, ratio


def test_prefetch_yields_the_same_items_in_order_and_overlaps():
    produced, main = [], threading.get_ident()

    def gen():
        for i in range(6):
            time.sleep(0.02)
            produced.append((i, threading.get_ident()))
            yield i * i
    t0 = time.perf_counter()
    got = []
    for x in prefetch(gen()):
        time.sleep(0.02)                                   # the "device work" of this item
        got.append(x)
    dt = time.perf_counter() - t0
    assert got == [i * i for i in range(6)] == list(gen())
    assert all(t != main for _, t in produced[:6])         # produced on the worker thread ...
    assert dt < 0.02 * 12 * 0.85                           # ... while the consumer worked: faster than producing and consuming in turn
    assert list(prefetch([])) == [] and list(prefetch(iter(range(3)), depth=4)) == [0, 1, 2]


def test_prefetch_reraises_where_the_item_was_due():
    def bad():
        yield 1
        yield 2
        raise KeyError("tokenizer failed")
    seen = []
    with pytest.raises(KeyError):
        for x in prefetch(bad()):
            seen.append(x)
    assert seen == [1, 2]


def test_encode_through_the_prefetch_equals_the_serial_loop(tok):
    """DenseEncoder.encode (tiny model, CPU): sub-batches tokenised one step ahead on a host thread == every sentence encoded from ids
    tokenised up front, in the caller's thread."""
    enc = encoders.random_init("dpr", device="cpu", size="tiny", seed=1)
    rng = np.random.default_rng(5)
    texts = [" ".join(f"w{rng.integers(0, 300)}" for _ in range(int(rng.integers(1, 40)))) for _ in range(150)]
    got = enc.encode(texts, batch_size=16)
    ids, mask = enc.tokenizer(texts, enc.max_doc_length)
    with torch.no_grad():
        exp = torch.cat([enc.encode_ids(ids[i: i + 1, : int(mask[i].sum())], mask[i: i + 1, : int(mask[i].sum())]) for i in range(len(texts))])
    assert got.shape == exp.shape
    assert float((got - exp).abs().max()) < 2e-5           # (padding inside a sub-batch moves fp32 sums by rounding only)
    order = sorted(range(len(texts)), key=lambda i: -len(texts[i]))
    serial = [(idx, i.clone(), m.clone()) for idx, i, m in ((order[s: s + 16], *enc.tokenizer([texts[j] for j in order[s: s + 16]], enc.max_doc_length))
                                                            for s in range(0, len(order), 16))]
    piped = list(enc._batches(texts, 16, enc.max_doc_length))
    assert len(serial) == len(piped)
    for (i0, a0, m0), (i1, a1, m1) in zip(serial, piped):
        assert i0 == i1 and torch.equal(a0, a1) and torch.equal(m0, m1)
