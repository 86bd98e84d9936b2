"""The slice of onmt.io the training driver touches: special tokens, make_features, vocab (de)serialisation, and the
torchtext-free dataset / iterator classes of textdata.py."""
from collections import defaultdict

import torch

PAD_WORD = "<blank>"      # onmt/io/DatasetBase.py:7-11
UNK = 0
BOS_WORD = "<s>"
EOS_WORD = "</s>"


def make_features(batch, side, data_type="text"):
    """onmt/io/IO.py:117-142: [len, batch] ids (plus optional feature levels) -> [len, batch, nfeats]."""
    assert side in ("src", "tgt")
    data = batch.__dict__[side]
    if isinstance(data, tuple):
        data = data[0]
    keys = sorted(k for k in batch.__dict__ if (side + "_feat_") in k)
    levels = [data] + [batch.__dict__[k] for k in keys]
    if data_type == "text":
        return torch.cat([lv.unsqueeze(2) for lv in levels], 2)
    return levels[0]


def collect_feature_vocabs(fields, side):
    return []


def save_fields_to_vocab(fields):
    """onmt/io/IO.py:64-75: list of (name, vocab) for fields that have one."""
    out = []
    for k, f in fields.items():
        if f is not None and "vocab" in f.__dict__:
            f.vocab.stoi = dict(f.vocab.stoi)
            out.append((k, f.vocab))
    return out


from .textdata import (Batch, Example, Field, OrderedIterator, RandomShuffler, TextDataset, Vocab, batch, get_fields,  # noqa: E402,F401
                       load_dataset, load_vocab, pool)


def load_fields_from_vocab(vocab, data_type="text"):
    """onmt/io/IO.py:51-66: Field objects for src / tgt / indices with the saved vocabularies attached"""
    vocab = dict(vocab)
    fields = get_fields(len([k for k in vocab if k.startswith("src_feat_")]), len([k for k in vocab if k.startswith("tgt_feat_")]))
    for k, v in vocab.items():
        v.stoi = defaultdict(lambda: 0, v.stoi)
        if k in fields:
            fields[k].vocab = v
        else:
            fields[k] = Field(vocab=v)
    return fields
