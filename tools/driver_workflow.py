"""Wall clock of the build's driver on a Multi30k-sized synthetic corpus (GPU box): where an epoch of the run scripts' recipe goes --
loading, training, validation, BLEU model selection, checkpoints.  Writes the corpus with the mirror's own dataset / vocabulary classes
(29 000 training + 1 014 validation pairs, lengths U[10,20], Zipf word ids over 10 k-word vocabularies, image rows from the committed
PyTables file, reused modulo its length), runs `variational_mmt_amd.train_mm_vi_model1.main` in this process with timers around the
trainer's phases, and prints seconds per phase.      python tools/driver_workflow.py [epochs] [extra driver flags ...]"""
import collections
import os
import random
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import variational_mmt_amd                                      # noqa: E402
onmt = variational_mmt_amd.install_as_onmt()
from variational_mmt_amd.onmt.io import textdata as td          # noqa: E402
from variational_mmt_amd import train_mm_vi_model1 as drv       # noqa: E402

H5 = os.path.join(ROOT, "tests", "golden", "h5", "pt_feats2048.h5")
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
extra = sys.argv[2:]
tmp = tempfile.mkdtemp(prefix="vmmt_wf_")
rng = random.Random(7)
V = 10000
itos = {"src": ["<unk>", "<blank>"] + ["s%d" % i for i in range(V - 2)], "tgt": ["<unk>", "<blank>", "<s>", "</s>"] + ["t%d" % i for i in range(V - 4)]}
import tables                                                   # noqa: E402  (the mirror's reader, aliased by install_as_onmt)
nrows = len(tables.open_file(H5, "r").root.global_feats)


def words(side, lo, n):
    w = itos[side]
    return tuple(w[min(len(w) - 1, lo + int((len(w) - lo) ** rng.random()) - 1)] for _ in range(n))


plain = {}
for split, n in (("train", 29000), ("valid", 1014)):
    exs = []
    for i in range(n):
        ex = td.Example()
        ex.src, ex.tgt, ex.indices = words("src", 2, rng.randint(10, 20)), words("tgt", 4, rng.randint(8, 18)), i % nrows
        exs.append(ex)
    torch.save(td.TextDataset(exs, []), os.path.join(tmp, "syn.%s.1.pt" % split))
    plain[split] = exs
vocab = []
for side in ("src", "tgt"):
    v = td.Vocab(itos[side])
    v.freqs = collections.Counter({w: 2 for w in itos[side]})
    vocab.append((side, v))
torch.save(vocab, os.path.join(tmp, "syn.vocab.pt"))
vsrc, vtgt = os.path.join(tmp, "v.src"), os.path.join(tmp, "v.tgt")
open(vsrc, "w").write("".join(" ".join(e.src) + "\n" for e in plain["valid"]))
open(vtgt, "w").write("".join(" ".join(e.tgt) + "\n" for e in plain["valid"]))

T = collections.OrderedDict()


def timed(obj, name, label=None):
    fn = getattr(obj, name)

    def wrap(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            torch.cuda.synchronize()
            T.setdefault(label or name, []).append(time.perf_counter() - t0)
    setattr(obj, name, wrap)


TM = onmt.TrainerMultimodal
from variational_mmt_amd.onmt import EarlyStop as ES            # noqa: E402
for m in ("train", "validate", "drop_checkpoint"):
    timed(TM, m)
import variational_mmt_amd.onmt.io as OIO                       # noqa: E402
timed(OIO, "load_dataset", "  load_dataset (in train / validate)")
timed(td, "_id_cache", "  id cache (in train / validate)") if hasattr(td, "_id_cache") else None
timed(ES.EarlyStop, "translate_", "  translate (in train)")
timed(ES.EarlyStop, "compute_bleus", "  BLEU (in train)")
argv = ["-data", os.path.join(tmp, "syn"), "-save_model", os.path.join(tmp, "m"), "-gpuid", "0", "-batch_size", "40", "-valid_batch_size", "40",
        "-path_to_train_img_feats", H5, "-path_to_valid_img_feats", H5, "-optim", "adam", "-learning_rate", "0.002",
        "--use_global_image_features", "--multimodal_model_type", "vi-model1", "--z_latent_dim", "500", "-dropout", "0.5", "-seed", "5",
        "-report_every", "200", "-epochs", str(epochs), "-early_stopping_criteria", "bleu", "-src", vsrc, "-tgt", vtgt,
        "-overwrite_model_file", "-patience", "50"] + extra
t0 = time.perf_counter()
drv.main(argv)
torch.cuda.synchronize()
total = time.perf_counter() - t0
print("\n== driver wall clock %.2f s for %d epochs (run scripts' model flags: 2-layer uni-directional LSTM 500, z 500, batch 40)" % (total, epochs))
for k, v in T.items():
    print("  %-22s calls %3d   total %7.2f s   mean %7.3f s   last %7.3f s" % (k, len(v), sum(v), sum(v) / len(v), v[-1]))
steps = 29000 // 40 * epochs
nested = sum(sum(T.get(k, [])) for k in ("  translate (in train)", "  BLEU (in train)"))
print("  (train() contains the in-epoch evaluations: translate + BLEU + their temporary checkpoints; ~%.2f ms per update without translate / BLEU)"
      % ((sum(T["train"]) - nested) / steps * 1e3))
