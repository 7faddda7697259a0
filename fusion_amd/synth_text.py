"""Synthetic French-LIKE text for the benches and tests (there is no dataset and no tokenizer file offline): pseudo-words built from
French syllables under a Zipf law, joined by real function words, with elisions and accents -- text a sub-word tokenizer has real work
on (frequent words = one piece, rare ones = several), of LLeQA's shape (questions of 5-45 words, articles of 100-500).  Deterministic in
the seed.  Used by tools/train_synth_tokenizer.py (which wrote fusion_amd/tuned/synth_fr_tokenizer.json.gz), bench.py and the tests."""
from __future__ import annotations

import numpy as np

_ONSETS = ["", "b", "c", "ch", "d", "f", "g", "gr", "j", "l", "m", "n", "p", "pr", "qu", "r", "s", "t", "tr", "v", "bl", "cr", "pl", "st", "dr", "fl"]
_NUCLEI = ["a", "e", "i", "o", "u", "é", "è", "ai", "au", "eau", "ou", "oi", "an", "en", "on", "in", "eu", "ie", "ui", "â", "ê", "io"]
_CODAS = ["", "", "", "r", "s", "t", "l", "n", "x", "nt", "re", "le", "te", "se", "que", "ment", "tion", "eur", "age", "ité", "aire", "able"]
_FUNCTION = ["le", "la", "les", "de", "des", "du", "un", "une", "et", "à", "au", "aux", "en", "dans", "pour", "par", "sur", "avec", "que", "qui",
             "est", "sont", "a", "ont", "peut", "doit", "ne", "pas", "ce", "cette", "ces", "son", "sa", "ses", "leur", "il", "elle", "on", "nous",
             "vous", "si", "ou", "où", "comment", "quand", "quel", "quelle", "quels", "combien", "pourquoi", "mon", "ma", "mes", "se", "y",
             "l'", "d'", "qu'", "n'", "s'", "j'", "c'"]


def lexicon(n_words: int = 120000, seed: int = 0) -> list[str]:
    """`n_words` distinct pseudo-words of 1-4 syllables (short ones first: they get the high Zipf ranks, as in a real language)."""
    rng = np.random.default_rng(seed)
    seen, out = set(_FUNCTION), []
    nsyl = 1
    while len(out) < n_words:
        for _ in range(4 * n_words):
            k = int(rng.integers(max(1, nsyl - 1), nsyl + 1))
            w = "".join(_ONSETS[rng.integers(len(_ONSETS))] + _NUCLEI[rng.integers(len(_NUCLEI))] + (_CODAS[rng.integers(len(_CODAS))] if s == k - 1 else "")
                        for s in range(k))
            if len(w) >= 2 and w not in seen:
                seen.add(w); out.append(w)
                if len(out) >= min(n_words, 600 * nsyl ** 3):
                    break
        nsyl += 1
    return out[:n_words]


class FrenchLike:
    def __init__(self, n_words: int = 120000, seed: int = 0, zipf: float = 1.07):
        self.words = lexicon(n_words, seed)
        p = 1.0 / np.arange(1, n_words + 1) ** zipf
        self.p = p / p.sum()

    def sentences(self, rng: np.random.Generator, n: int, min_words: int, max_words: int, question: bool = False) -> list[str]:
        lens = rng.integers(min_words, max_words + 1, n)
        total = int(lens.sum())
        content = rng.choice(len(self.words), size=total, p=self.p)
        func = rng.integers(0, len(_FUNCTION), total)
        is_func = rng.random(total) < 0.42                      # about four running words in ten are function words
        caps = rng.random(total) < 0.03
        out, k = [], 0
        for L in lens.tolist():
            toks = []
            for j in range(k, k + L):
                w = _FUNCTION[func[j]] if is_func[j] else self.words[content[j]]
                if caps[j] and not is_func[j]:
                    w = w.capitalize()
                if toks and toks[-1].endswith("'"):            # elision: l' + word
                    toks[-1] += w
                else:
                    toks.append(w)
            if toks and toks[-1].endswith("'"):
                toks[-1] = toks[-1][:-1] + "e"
            s = " ".join(toks)
            s = s[:1].upper() + s[1:]
            out.append(s + (" ?" if question else "."))
            k += L
        return out
