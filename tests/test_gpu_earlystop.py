"""GPU: BLEU-driven model selection / early stopping through the onmt surface: TrainerMultimodal evaluates
every N updates by translating the validation source with the live model (arg-max decoding on the GPU), scores it with the
BLEU restatement, keeps `<save_model>_BestModelBleu.pt/.pkl` and raises the stop signal after `patience` evaluations without
improvement (reference: onmt/TrainerMultimodal.py:372-396, onmt/EarlyStop.py)."""
import os
import pickle
import types

import pytest
import torch

from oracle import vi1_oracle as O
from tests.test_gpu_onmt_surface import _Batch, _Vocab, _opt

pytestmark = pytest.mark.gpu


def _files(tmp_path, c, n, seed):
    g = torch.Generator().manual_seed(seed)
    sv, tv = _Vocab(c.vs, False), _Vocab(c.vt, True)
    src_lines, tgt_lines = [], []
    for _ in range(n):
        L = int(torch.randint(1, 9, (1,), generator=g))
        ids = torch.randint(2, c.vs, (L,), generator=g).tolist()
        src_lines.append(" ".join(sv.itos[i] for i in ids))
        tgt_lines.append(" ".join(tv.itos[4 + (i % (c.vt - 4))] for i in ids))
    src_lines[3] = src_lines[3] + " neverseen"          # an out-of-vocabulary word maps to <unk>
    sp, tp = tmp_path / "val.src", tmp_path / "val.tgt"
    sp.write_text("\n".join(src_lines) + "\n", encoding="utf-8")
    tp.write_text("\n".join(tgt_lines) + "\n", encoding="utf-8")
    return str(sp), str(tp), sv, tv, src_lines


def test_translate_file_restores_order_and_matches_oracle(tmp_path):
    from variational_mmt_amd.engine import Dims
    from variational_mmt_amd.onmt.Models import NMTVIModel
    from variational_mmt_amd.onmt.translate.translate_file import translate_file
    c = O.Cfg(vs=37, vt=41, emb=12, hid=16, z=8, layers=1, brnn=True)
    p = O.init_params(c, seed=3)
    for k in p:
        if "inf_net_image" not in k:
            p[k] = p[k] * 6.0
    sp, tp, sv, tv, src_lines = _files(tmp_path, c, 23, 5)
    model = NMTVIModel(Dims(c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, c.brnn, 0.0), dtype="f32", device="cuda", param_init=0.0)
    model.engine.load_state_dict(p)
    fields = {"src": types.SimpleNamespace(vocab=sv), "tgt": types.SimpleNamespace(vocab=tv)}
    out = translate_file(model, fields, sp, str(tmp_path / "hyp"), batch_size=5, max_length=9)
    assert open(tmp_path / "hyp", encoding="utf-8").read().split("\n")[:-1] == out and len(out) == 23
    for i, line in enumerate(src_lines):
        ids = torch.tensor([sv.stoi.get(w, 0) for w in line.split()], dtype=torch.int64)
        toks, _ = O.greedy_decode(p, c, ids.view(-1, 1), torch.tensor([len(ids)]), 9)
        col = toks[:, 0].tolist()
        col = col[:col.index(3)] if 3 in col else col
        assert out[i] == " ".join(tv.itos[t] for t in col), i
    # beam search through the same entry point: one line per sentence, every token a target word
    outb = translate_file(model, fields, sp, str(tmp_path / "hypb"), batch_size=7, beam_size=3, max_length=9)
    assert len(outb) == 23 and all(all(w in tv.stoi for w in ln.split()) for ln in outb)


def test_bleu_early_stopping_through_the_trainer(tmp_path):
    import variational_mmt_amd
    onmt = variational_mmt_amd.install_as_onmt()
    c = O.Cfg(vs=29, vt=31, emb=16, hid=32, z=8, img=2048, layers=1, brnn=True)
    sp, tp, sv, tv, _ = _files(tmp_path, c, 12, 9)
    fields = {"src": types.SimpleNamespace(vocab=sv), "tgt": types.SimpleNamespace(vocab=tv)}
    opt = _opt(c, tmp_path)
    opt.early_stopping_criteria, opt.evaluate_every_n_model_updates, opt.patience, opt.start_early_stopping_at = "bleu", 2, 2, 0
    opt.src, opt.tgt, opt.path_to_valid_img_feats = sp, tp, "valid.hdf5"
    model = onmt.ModelConstructor.make_vi_model_mmt(opt, fields, True, None)
    loss = onmt.VILoss.NMTVIModel1LossCompute(model.generator, tv)
    optim = onmt.Optim("adam", 0.002, 5.0, lr_decay=0.5, start_decay_at=8)
    optim.set_parameters(model.parameters())
    bts = [O.synth_batch(c, 6, 5, 7, n_img=10, seed=300 + i, fixed_len=False) for i in range(12)]
    table = bts[0]["table"].numpy()
    trainer = onmt.TrainerMultimodal(model, loss, loss, optim, 0, 32, "text", "sents", 1, train_img_feats=table, valid_img_feats=table,
                                     multimodal_model_type="vi-model1", model_opt=opt, fields=fields)
    es = trainer.early_stop
    assert type(es).__name__ == "EarlyStop" and es.evaluate_every_nupdates == 2
    scripted = iter(["10.00", "12.50", "11.00", "9.00", "8.00", "7.00"])
    real = es.compute_bleus
    seen = []

    def compute(hyp, ref, split="valid"):
        names, scores, files = real(hyp, ref, split)        # the real path runs: translation file + BLEU restatement
        seen.append(scores[0])
        assert len(open(hyp, encoding="utf-8").read().split("\n")) == 13
        return names, [next(scripted)], files               # ... but the DECISIONS are driven by a scripted score curve
    es.compute_bleus = compute
    trainer.train([_Batch(bt, "cuda") for bt in bts], 1, None)
    # evaluations at 2, 4, 6, 8 updates: best at 4 (12.50); with patience 2 the signal comes at the 4th evaluation
    assert sorted(es.results_bleu) == [2, 4, 6, 8] and es.signal_early_stopping and trainer.n_model_updates == 8
    assert all(s != "" and 0.0 <= float(s) <= 100.0 for s in seen)
    best = pickle.load(open(opt.save_model + "_BestModelBleu.pkl", "rb"))
    assert best["n_updates"] == 4 and best["bleu"] == 12.5
    ck = torch.load(opt.save_model + "_BestModelBleu.pt", map_location="cpu", weights_only=False)
    assert sorted(ck.keys()) == ["epoch", "generator", "model", "opt", "optim", "vocab"]
    assert not [f for f in os.listdir(tmp_path) if f.startswith("tmp")]
    assert model.training
