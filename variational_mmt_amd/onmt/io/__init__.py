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
        if len(levels) == 1:                     # no word features: a view, not a device copy
            return levels[0].unsqueeze(2)
        return torch.cat([lv.unsqueeze(2) for lv in levels], 2)
    return levels[0]


def collect_features(fields, side="src"):
    """onmt/io/IO.py:145-156: names of the word-feature fields of one side (`src_feat_0`, `src_feat_1`, ...), in order"""
    assert side in ("src", "tgt")
    names, j = [], 0
    while "%s_feat_%d" % (side, j) in fields:
        names.append("%s_feat_%d" % (side, j))
        j += 1
    return names


def collect_feature_vocabs(fields, side):
    """onmt/io/IO.py:159-171: the vocabularies of those fields"""
    return [fields[k].vocab for k in collect_features(fields, side)]


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


def build_dataset(fields, data_type, src_path, tgt_path, src_dir=None, src_seq_length=0, tgt_seq_length=0, src_seq_length_trunc=0,
                  tgt_seq_length_trunc=0, dynamic_dict=True, sample_rate=0, window_size=0, window_stride=0, window=None,
                  normalize_audio=True, use_filter_pred=True):
    """onmt/io/IO.py:173-218 for text corpora (the translation driver builds its test set this way, translate_mm_vi.py:104-111): one
    example per line -- `src` (and `tgt`) = the line's whitespace tokens as a tuple, cut to the truncation length, `indices` = the
    line number (TextDataset.read_text_file, onmt/io/TextDataset.py:149-173); with use_filter_pred examples outside
    0 < len <= seq_length are dropped (:74-77).  Word features (tokens carrying the feature separator) and the copy-attention
    dictionaries are outside the VI_Model1 path."""
    if data_type != "text":
        raise NotImplementedError("data_type %r: only text sources are on the VI_Model1 path" % (data_type,))

    def lines(path, trunc):
        with open(path, encoding="utf-8") as f:
            for ln in f:
                toks = ln.strip().split()
                if any(u"\uffe8" in t for t in toks):
                    raise NotImplementedError("word features are outside the VI_Model1 path")
                yield tuple(toks[:trunc] if trunc else toks)
    srcs = list(lines(src_path, src_seq_length_trunc))
    tgts = list(lines(tgt_path, tgt_seq_length_trunc)) if tgt_path else None
    examples = []
    for i, s in enumerate(srcs):
        if tgts is not None and i >= len(tgts):
            break
        ex = Example()
        ex.src, ex.indices = s, i
        if tgts is not None:
            ex.tgt = tgts[i]
        if use_filter_pred and not (0 < len(s) <= src_seq_length and (tgts is None or 0 < len(tgts[i]) <= tgt_seq_length)):
            continue
        examples.append(ex)
    keys = ("src", "tgt", "indices") if tgts is not None else ("src", "indices")
    return TextDataset(examples, dict((k, fields[k]) for k in keys if k in fields))
