"""Tiny on-disk checkpoints for the loader tests (written by the tests themselves into tmp_path; no network)."""
import json
import os

import torch


def tiny_tokenizer(path):
    """A word-level tokenizer with RoBERTa-style specials and pair template, saved as a PreTrainedTokenizerFast."""
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast
    words = ["le", "la", "loi", "article", "code", "civil", "chat", "chien", "droit", "bail", "juge", "peut", "est", "un", "une", "de"]
    punct = list("!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~")
    vocab = {"<s>": 0, "<pad>": 1, "</s>": 2, "<unk>": 3}
    for w in words + punct:
        vocab[w] = len(vocab)
    vocab["<mask>"] = len(vocab)
    tok = Tokenizer(models.WordLevel(vocab=vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.WhitespaceSplit(), pre_tokenizers.Punctuation()])
    tok.post_processor = processors.TemplateProcessing(single="<s> $A </s>", pair="<s> $A </s> </s> $B </s>",
                                                       special_tokens=[("<s>", 0), ("</s>", 2)])
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, bos_token="<s>", eos_token="</s>", unk_token="<unk>", pad_token="<pad>",
                                   mask_token="<mask>", cls_token="<s>", sep_token="</s>", model_max_length=128)
    fast.save_pretrained(path)
    return fast, len(vocab)


def tiny_config(vocab_size, heads=4, hidden=64):
    return dict(vocab_size=vocab_size, hidden_size=hidden, num_hidden_layers=2, num_attention_heads=heads, intermediate_size=128,
                max_position_embeddings=130, type_vocab_size=1, pad_token_id=1, bos_token_id=0, eos_token_id=2)


def write_colbert_checkpoint(path, seed=0, dim=16, heads=4, hidden=64):
    """colbert-ai layout: the base model under its prefix (`roberta.`) + `linear.weight` + artifact.metadata."""
    from safetensors.torch import save_file
    from transformers import CamembertConfig, CamembertModel
    os.makedirs(path, exist_ok=True)
    fast, V = tiny_tokenizer(path)
    torch.manual_seed(seed)
    cfg = CamembertConfig(**tiny_config(V, heads, hidden))
    base = CamembertModel(cfg, add_pooling_layer=False).eval()
    lin = torch.randn(dim, hidden) * 0.2
    sd = {"roberta." + k: v.contiguous() for k, v in base.state_dict().items()}
    sd["linear.weight"] = lin
    save_file(sd, os.path.join(path, "model.safetensors"))
    cfg.architectures = ["HF_ColBERT"]
    cfg.save_pretrained(path)
    with open(os.path.join(path, "artifact.metadata"), "w") as f:
        json.dump({"query_maxlen": 16, "doc_maxlen": 40, "dim": dim, "mask_punctuation": True, "attend_to_mask_tokens": True,
                   "query_token_id": "[unused0]", "doc_token_id": "[unused1]", "similarity": "cosine"}, f)
    return fast, base, lin


def write_monobert_checkpoint(path, seed=1, heads=4, hidden=64):
    from transformers import CamembertConfig, CamembertForSequenceClassification
    os.makedirs(path, exist_ok=True)
    fast, V = tiny_tokenizer(path)
    torch.manual_seed(seed)
    model = CamembertForSequenceClassification(CamembertConfig(num_labels=1, **tiny_config(V, heads, hidden))).eval()
    model.save_pretrained(path)
    return fast, model
