#!/usr/bin/env python3
"""Train the sub-word tokenizer the benches tokenise with (the reference's encode starts from STRINGS --
hybrid.py:101-102 -> model.encode(queries); splade/base.py:142-171 tokenises inside encode -- and no tokenizer file exists offline).
CamemBERT's is a 32,005-piece SentencePiece model; this one is a byte-fallback-free BPE of the same size and layout (Metaspace pieces,
<s> ... </s> template, camembert's special-token ids) trained with the `tokenizers` library on fusion_amd/synth_text.py's French-like
text.  Output: fusion_amd/tuned/synth_fr_tokenizer.json.gz (a few hundred kB).  Deterministic; run in the build container (no network):
    python tools/train_synth_tokenizer.py"""
import gzip
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fusion_amd.synth_text import FrenchLike  # noqa: E402

OUT = os.path.join(ROOT, "fusion_amd", "tuned", "synth_fr_tokenizer.json.gz")
VOCAB = 32005
SPECIALS = ["<s>NOTUSED", "<pad>", "</s>NOTUSED", "<unk>", "<unk>NOTUSED", "<s>", "</s>"]      # camembert-base ids 0..6; <mask> = 32004


def main():
    from tokenizers import Tokenizer, decoders, models, normalizers, pre_tokenizers, processors, trainers
    gen = FrenchLike()
    rng = np.random.default_rng(2024)
    corpus = gen.sentences(rng, 60000, 5, 45, question=True) + gen.sentences(rng, 12000, 100, 500)
    tok = Tokenizer(models.BPE(unk_token="<unk>"))
    tok.normalizer = normalizers.Sequence([normalizers.NFKC(), normalizers.Replace("  ", " ")])
    tok.pre_tokenizer = pre_tokenizers.Metaspace(replacement="▁", prepend_scheme="always")
    tok.decoder = decoders.Metaspace(replacement="▁", prepend_scheme="always")
    trainer = trainers.BpeTrainer(vocab_size=VOCAB - 1, special_tokens=SPECIALS, show_progress=False, min_frequency=2)
    tok.train_from_iterator(corpus, trainer=trainer, length=len(corpus))
    tok.add_special_tokens(["<mask>"])
    tok.post_processor = processors.TemplateProcessing(single="<s> $A </s>", pair="<s> $A </s> </s> $B </s>",
                                                       special_tokens=[("<s>", tok.token_to_id("<s>")), ("</s>", tok.token_to_id("</s>"))])
    assert tok.token_to_id("<pad>") == 1 and tok.token_to_id("<s>") == 5 and tok.token_to_id("</s>") == 6, "camembert's special-token ids"
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with gzip.GzipFile(OUT, "wb", mtime=0) as f:
        f.write(tok.to_str().encode("utf-8"))
    print(f"{OUT}: vocab {tok.get_vocab_size()} ({os.path.getsize(OUT)} bytes); <mask> = {tok.token_to_id('<mask>')}")
    e = tok.encode(corpus[0])
    print(corpus[0], "->", len(e.ids), "pieces:", e.tokens[:24])


if __name__ == "__main__":
    main()
