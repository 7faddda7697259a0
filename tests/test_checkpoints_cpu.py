"""encoders.from_pretrained on real (tiny, locally written) checkpoints: the ColBERT projection, [Q] / [D] markers and
punctuation skiplist come from the checkpoint / tokenizer, the monoBERT loader encodes true pairs.  CPU: the encoders take
their plain HF path on device='cpu' (the padding-free HIP path is compared with this one in the -m gpu suite).
Expected values are an independent restatement of colbert-ai's tokenisation + forward (requirements.txt:15; absent from the
reference tree: parity unpinned) and of sentence-transformers' CrossEncoder.predict."""
import string

import numpy as np
import torch

from checkpoint_utils import write_colbert_checkpoint, write_monobert_checkpoint

QUERIES = ["le juge peut , un bail ?", "la loi", "article de le code civil ( droit ) chat chien chat chien chat chien chat chien chat chien"]
DOCS = ["le chat est un chien .", "article : le bail , la loi ; le code civil !", "droit"]


def colbert_reference(fast, base, lin, queries, docs, Lq, Ld):
    """colbert-ai: QueryTokenizer / DocTokenizer.tensorize + ColBERT.query / .doc, written out with plain HF calls."""
    unk = fast.unk_token_id
    assert fast.convert_tokens_to_ids("[unused0]") == unk            # CamemBERT-style vocabulary: both markers are <unk>
    q = fast([". " + t for t in queries], padding="max_length", truncation=True, max_length=Lq, return_tensors="pt")
    ids = q["input_ids"].clone(); ids[:, 1] = unk
    ids[ids == fast.pad_token_id] = fast.mask_token_id
    with torch.no_grad():
        Q = torch.nn.functional.normalize(base(input_ids=ids, attention_mask=torch.ones_like(ids)).last_hidden_state @ lin.T, p=2, dim=2)
    d = fast([". " + t for t in docs], padding="longest", truncation="longest_first", max_length=Ld, return_tensors="pt")
    dids = d["input_ids"].clone(); dids[:, 1] = unk
    skip = {fast.encode(sym, add_special_tokens=False)[0] for sym in string.punctuation}
    with torch.no_grad():
        D = torch.nn.functional.normalize(base(input_ids=dids, attention_mask=d["attention_mask"]).last_hidden_state @ lin.T, p=2, dim=2)
    keep = [[(int(x) not in skip) and (int(x) != fast.pad_token_id) for x in row] for row in dids]
    return Q, [D[i][torch.tensor(keep[i])] for i in range(len(docs))]


def test_colbert_from_pretrained_loads_projection_markers_and_skiplist(tmp_path):
    from fusion_amd import encoders
    fast, base, lin = write_colbert_checkpoint(str(tmp_path / "colbert"))
    enc = encoders.from_pretrained(str(tmp_path / "colbert"), "colbert", device="cpu")
    assert enc.dim == 16 and torch.equal(enc.linear.weight.detach(), lin)          # the checkpoint's projection, not a random one
    assert enc.q_marker_id == fast.unk_token_id == enc.d_marker_id
    assert enc.max_query_length == 16 and enc.max_doc_length == 40
    assert set(enc.punct_ids.tolist()) == {fast.encode(sym, add_special_tokens=False)[0] for sym in string.punctuation}
    Qe, De = colbert_reference(fast, base, lin, QUERIES, DOCS, 16, 40)
    Qg = enc.encode_queries(QUERIES)
    assert Qg.shape == (3, 16, 16) and Qg.dtype == torch.float16
    assert torch.max(torch.abs(Qg.float() - Qe)).item() <= 1e-3                    # fp16 storage of unit vectors
    tok, off = enc.encode_docs(DOCS)
    assert off.tolist() == np.concatenate([[0], np.cumsum([len(x) for x in De])]).tolist()
    assert torch.max(torch.abs(tok.float() - torch.cat(De))).item() <= 1e-3
    # the punctuation rows really are gone: doc 1 has four symbols (the "." placeholder became the [D] marker and stays)
    assert len(De[1]) == len(fast(". " + DOCS[1])["input_ids"]) - 4


def test_monobert_from_pretrained_encodes_true_pairs(tmp_path):
    from fusion_amd import encoders
    from fusion_amd.retrievers.hybrid import Ranker
    fast, model = write_monobert_checkpoint(str(tmp_path / "mono"))
    ce = encoders.from_pretrained(str(tmp_path / "mono"), "monobert", device="cpu")
    assert ce.activation == "sigmoid"                                              # one label: CrossEncoder's default activation
    pairs = [(q, d) for q in QUERIES[:2] for d in DOCS]
    got = ce.predict(pairs)
    e = fast([p[0] for p in pairs], [p[1] for p in pairs], padding=True, truncation="longest_first", max_length=128, return_tensors="pt")
    assert (e["input_ids"] == fast.eos_token_id).sum(1).min().item() == 3          # "<s> q </s></s> d </s>": a real pair encoding
    with torch.no_grad():
        exp = torch.sigmoid(model(**e).logits[:, 0])
    assert torch.max(torch.abs(got - exp)).item() <= 1e-6
