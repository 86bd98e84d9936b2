"""Translate a tokenised source text file with the live model (what translate_mm_vi.py:103-160 does for one sentence per
batch, minus the torchtext dataset): words -> ids by the source vocabulary (<unk> = 0), sentences bucketed by length
into decoding batches (sorted by decreasing length inside a batch, as the packed encoder needs), arg-max or beam decoding on
the GPU, ids -> words by the target vocabulary up to (not including) </s> (`_build_target_tokens`, Translation.py:30-40),
output lines in the ORIGINAL order."""
import types

import torch

from .TranslatorMultimodalVI import TranslatorMultimodalVI


def _stoi(vocab, w):
    s = vocab.stoi
    return s[w] if w in s else 0


def translate_file(model, fields, src_path, out_path, batch_size=64, beam_size=1, n_best=1, max_length=100, global_scorer=None,
                   min_length=0, max_src_len=64):
    sv, tv = fields["src"].vocab, fields["tgt"].vocab
    with open(src_path, encoding="utf-8") as f:
        sents = [ln.split() for ln in f]
    ids = [[_stoi(sv, w) for w in s[:max_src_len]] or [0] for s in sents]          # an empty line decodes from a lone <unk>
    order = sorted(range(len(ids)), key=lambda i: -len(ids[i]))
    tr = TranslatorMultimodalVI(model, fields, beam_size=beam_size, n_best=n_best, max_length=max_length,
                                global_scorer=global_scorer, min_length=min_length)
    eos = tv.stoi["</s>"] if "</s>" in tv.stoi else 3
    out = [None] * len(ids)
    was_training = getattr(model, "training", False)
    model.eval()
    try:
        for i in range(0, len(order), batch_size):
            grp = order[i:i + batch_size]
            S = len(ids[grp[0]])
            src = torch.full((S, len(grp)), 1, dtype=torch.int64, device="cpu")
            for j, k in enumerate(grp):
                src[:len(ids[k]), j] = torch.tensor(ids[k], dtype=torch.int64, device="cpu")
            lens = torch.tensor([len(ids[k]) for k in grp], dtype=torch.int64, device="cpu")
            ret = tr.translate_batch(types.SimpleNamespace(src=(src, lens), batch_size=len(grp)))
            for j, k in enumerate(grp):
                toks = ret["predictions"][j][0]
                if toks and toks[-1] == eos:
                    toks = toks[:-1]
                out[k] = " ".join(tv.itos[t] for t in toks)
    finally:
        if was_training:
            model.train()
    with open(out_path, "w", encoding="utf-8") as f:
        for ln in out:
            f.write(ln + "\n")
    return out
