"""TEST INFRASTRUCTURE -- writes tests/golden/textdata/{demo.train.1.pt, demo.valid.1.pt, demo.vocab.pt}: a tiny text dataset
pickled THROUGH the reference's own classes (onmt.io.TextDataset's __reduce_ex__/__getstate__ hack, onmt/io/DatasetBase.py:
27-35; onmt.io.save_fields_to_vocab + the Vocab __getstate__ patch, onmt/io/IO.py:17-27,64-75) in the legacy (non-zip)
torch.save container torch 0.3.1 wrote, plus demo.json with the same content in plain form.  torchtext itself is absent from
this image, so the Example / Vocab objects inside are the import stand-ins of oracle/stubs (their pickled names
`torchtext.data.Example`, `torchtext.vocab.Vocab`; the real package's `torchtext.data.example.Example` is mapped as well).
Build container only."""
import json
import os
import random
import sys
from collections import Counter

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_harness as RH          # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "textdata")


def main():
    onmt, _ = RH.import_reference()
    import torchtext
    import onmt.io
    os.makedirs(OUT, exist_ok=True)
    # s9: `super().__reduce_ex__()` (DatasetBase.py:33-35) relied on Python <= 3.6 defaulting the protocol to 0, i.e. on
    # copyreg._reduce_ex(self, 0): reconstructor + __getstate__ dict.  Python >= 3.7 requires the argument.
    import copyreg
    onmt.io.DatasetBase.ONMTDatasetBase.__reduce_ex__ = lambda self, proto=0: copyreg._reduce_ex(self, 0)
    rnd = random.Random(5)
    sw = ["ein", "mann", "hund", "läuft", "im", "park", "frau", "sitzt", "auf", "bank", "ro@@", "ter"]
    tw = ["a", "man", "dog", "runs", "in", "the", "park", "woman", "sits", "on", "bench", "red"]
    plain = {}
    for split, n in (("train", 57), ("valid", 11)):
        exs, rows = [], []
        for i in range(n):
            ex = torchtext.data.Example()
            ex.src = tuple(rnd.choice(sw) for _ in range(rnd.randint(1, 9)))
            ex.tgt = tuple(rnd.choice(tw) for _ in range(rnd.randint(1, 10)))
            ex.indices = i
            exs.append(ex)
            rows.append({"src": list(ex.src), "tgt": list(ex.tgt), "indices": i})
        ds = onmt.io.TextDataset.__new__(onmt.io.TextDataset)
        ds.examples, ds.fields = exs, []                      # preprocess.py:103-104: fields emptied before saving
        ds.data_type, ds.src_vocabs, ds.n_src_feats, ds.n_tgt_feats = "text", [], 0, 0
        torch.save(ds, os.path.join(OUT, "demo.%s.1.pt" % split), _use_new_zipfile_serialization=False)
        plain[split] = rows
    vocabs = {}
    for side, words, specials in (("src", sw, ["<unk>", "<blank>"]), ("tgt", tw, ["<unk>", "<blank>", "<s>", "</s>"])):
        v = torchtext.vocab.Vocab.__new__(torchtext.vocab.Vocab)
        v.itos = specials + sorted(words)
        v.stoi = {w: i for i, w in enumerate(v.itos)}
        v.freqs = Counter({w: 3 for w in words})
        v.vectors = None
        f = torchtext.data.Field()
        f.vocab = v
        vocabs[side] = f
    torch.save(onmt.io.save_fields_to_vocab(vocabs), os.path.join(OUT, "demo.vocab.pt"), _use_new_zipfile_serialization=False)
    plain["vocab"] = {k: f.vocab.itos for k, f in vocabs.items()}
    json.dump(plain, open(os.path.join(OUT, "demo.json"), "w", encoding="utf-8"), ensure_ascii=False)
    print("wrote", os.listdir(OUT))


if __name__ == "__main__":
    main()
