"""GPU: the flow of train_mm_vi_model1.py (`main`, :455-560) end to end on the mirrored surface, with nothing of torchtext /
PyTables / the reference importable: vocabulary + dataset pickles -> fields -> OrderedIterator batches; HDF5 feature file ->
HBM table; make_vi_model_mmt -> TrainerMultimodal.train (one epoch) -> validate -> epoch_step -> drop_checkpoint -> reload ->
translate the validation sources with beam search.  The first update is checked against the CPU oracle on the iterator's own
first batch."""
import os
import random
import types

import numpy as np
import pytest
import torch

from oracle import vi1_oracle as O
from tests.test_gpu_onmt_surface import _opt

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_driver_flow(tmp_path):
    import variational_mmt_amd
    onmt = variational_mmt_amd.install_as_onmt()
    import onmt.io
    from variational_mmt_amd.onmt import h5tables as tables                      # `import tables` of the driver
    data = os.path.join(G, "textdata", "demo")
    # load_fields (train_mm_vi_model1.py:388-412)
    train_ds = onmt.io.load_dataset(data + ".train.1.pt")
    valid_ds = onmt.io.load_dataset(data + ".valid.1.pt")
    fields = onmt.io.load_fields_from_vocab(onmt.io.load_vocab(data + ".vocab.pt"), "text")
    fields = dict((k, f) for k, f in fields.items() if k in train_ds.examples[0].__dict__)
    train_ds.fields = valid_ds.fields = fields
    # image features (:460-481)
    h5 = os.path.join(G, "h5", "pt_feats2048.h5")
    f = tables.open_file(h5, mode="r")
    feats = f.root.global_feats[:]
    f.close()
    assert feats.shape == (60, 2048) and feats.dtype == np.float32
    c = O.Cfg(vs=len(fields["src"].vocab), vt=len(fields["tgt"].vocab), emb=16, hid=32, z=8, img=2048, layers=1, brnn=True)
    opt = _opt(c, tmp_path)
    opt.path_to_train_img_feats = h5
    model = onmt.ModelConstructor.make_vi_model_mmt(opt, fields, True, None)
    p0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    loss = onmt.VILoss.NMTVIModel1LossCompute(model.generator, fields["tgt"].vocab)
    optim = onmt.Optim("adam", 0.002, 5.0, lr_decay=0.5, start_decay_at=8)
    optim.set_parameters(model.parameters())
    # the trainer takes the numpy array (reference contract) for training and the FILE PATH for validation (streamed to HBM)
    trainer = onmt.TrainerMultimodal(model, loss, loss, optim, 0, 32, "text", "sents", 1, train_img_feats=feats, valid_img_feats=h5,
                                     multimodal_model_type="vi-model1", model_opt=opt, fields=fields)
    random.seed(3)
    train_iter = onmt.io.OrderedIterator(dataset=train_ds, batch_size=8, device="cuda", sort=False, train=True,
                                         sort_within_batch=True, repeat=False)
    # ---- first batch of the epoch against the oracle ------------------------------------------------------------------
    random.seed(3)
    first = next(iter(onmt.io.OrderedIterator(dataset=train_ds, batch_size=8, device=None, sort=False, train=True,
                                              sort_within_batch=True, repeat=False)))
    random.seed(3)
    train_iter.random_shuffler = onmt.io.RandomShuffler()
    seen = []
    report = lambda epoch, i, n, t0, lr, st, mm: (seen.append((i, st.elbo_loss, st.n_words)), onmt.VIStatistics(mm))[1]
    stats = trainer.train(train_iter, 1, report)
    assert len(seen) == 8 and trainer.n_model_updates == 8 and stats.n_words == sum(s[2] for s in seen)
    src, sl = first.src
    tgt, tl = first.tgt
    # the sample eps of that update is not kept; at the initial parameters sigma is small against the NLL, so the oracle run with
    # eps = 0 must agree in the word count exactly and in the ELBO to a few per cent
    r, Lo, _g = O.step_grads(p0, c, src, sl, tgt, torch.from_numpy(feats)[first.indices], torch.zeros(first.batch_size, c.z))
    assert seen[0][2] == Lo["n_words"]
    assert abs(seen[0][1] - float(Lo["elbo"])) / abs(float(Lo["elbo"])) < 0.05      # z differs by the sample only
    # ---- validate, lr schedule, checkpoint, reload, translate ----------------------------------------------------------
    valid_iter = onmt.io.OrderedIterator(dataset=valid_ds, batch_size=4, device="cuda", sort=False, train=False,
                                         sort_within_batch=True, repeat=False)
    vs = trainer.validate(valid_iter)
    assert np.isfinite(vs.ppl()) and vs.n_words == sum(len(e.tgt) + 1 for e in valid_ds.examples)
    trainer.epoch_step(vs.ppl(), 1)
    fname = trainer.drop_checkpoint(opt, 1, fields, vs)
    ck = torch.load(fname, map_location="cpu", weights_only=False)
    assert dict(ck["vocab"])["tgt"].itos == fields["tgt"].vocab.itos
    fields2 = onmt.io.load_fields_from_vocab(ck["vocab"], "text")
    model2 = onmt.ModelConstructor.make_vi_model_mmt(ck["opt"], fields2, True, ck)
    src_txt = tmp_path / "valid.src"
    src_txt.write_text("\n".join(" ".join(e.src) for e in valid_ds.examples) + "\n", encoding="utf-8")
    from variational_mmt_amd.onmt.translate import GNMTGlobalScorer
    from variational_mmt_amd.onmt.translate.translate_file import translate_file
    out = translate_file(model2, fields2, str(src_txt), str(tmp_path / "hyp"), batch_size=4, beam_size=5, n_best=1, max_length=12,
                         global_scorer=GNMTGlobalScorer(0.0, -0.0))
    out1 = translate_file(model, fields, str(src_txt), str(tmp_path / "hyp1"), batch_size=11, beam_size=5, n_best=1, max_length=12,
                          global_scorer=GNMTGlobalScorer(0.0, -0.0))
    assert len(out) == 11 and out == out1                                         # reloaded model, other batching: same output
