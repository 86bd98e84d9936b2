"""TEST INFRASTRUCTURE -- has the REAL reference write a training checkpoint and records what the reference itself does when it
resumes from it.  Build container only:   python oracle/make_ckpt_golden.py

  tests/golden/ckpt/ref_ckpt.pt.gz    the dict of TrainerMultimodal.drop_checkpoint (onmt/TrainerMultimodal.py:554-622: model / generator /
                                      vocab / opt / epoch / optim = the pickled onmt.Optim.Optim holding a torch.optim.Adam) after
                                      two updates, written by the reference's own drop_checkpoint + torch.save
  tests/golden/ckpt/ref_resume.npz    the batches of the two updates and of a third one, and the parameters after the third
                                      update when the REFERENCE resumes from that file the way train_mm_vi_model1.py does
                                      (:433-454, :544-556: load -> make_vi_model_mmt(checkpoint) -> build_optim -> set_parameters ->
                                      one update)

The image network's second layer (2 x 2048 x 2048 floats) would make the file 100 MB: its weights are zero and the first layer's
bias is -10 (a dead ReLU), so its gradients and Adam moments are exactly zero and the file compresses to a few hundred KB.  The
format, not the image network's arithmetic, is what this fixture pins."""
import gzip
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_harness as RH          # noqa: E402
from oracle import vi1_oracle as O            # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "ckpt")
CFG = dict(vs=37, vt=41, emb=12, hid=16, z=8, layers=1, brnn=True)
B, S, T = 5, 7, 9


def _fields(c):
    import torchtext
    f = {}
    for side, n, specials in (("src", c.vs, ["<unk>", "<blank>"]), ("tgt", c.vt, ["<unk>", "<blank>", "<s>", "</s>"])):
        v = torchtext.vocab.Vocab.__new__(torchtext.vocab.Vocab)
        v.itos = specials + ["w%d" % i for i in range(n - len(specials))]
        v.stoi = {w: i for i, w in enumerate(v.itos)}
        v.freqs, v.vectors = {}, None
        fld = torchtext.data.Field()
        fld.vocab = v
        f[side] = fld
    return f


def _update(onmt, model, tloss, optim, bt):
    """one update the way TrainerMultimodal._gradient_accumulation does it (:625-718), with the shims of the training path"""
    img = bt["table"][bt["indices"]]
    model.zero_grad()
    with RH.inject_eps(bt["eps"]), RH.training_shims():
        outputs, attns, _ = model(bt["src"].unsqueeze(2), bt["tgt"].unsqueeze(2), bt["src_len"], bt["tgt_len"], img.clone())
        batch = RH.Batch(bt["src"], bt["src_len"], bt["tgt"], bt["tgt_len"], bt["indices"])
        batch.tgt = batch.tgt[0]
        tloss.sharded_compute_loss(batch, outputs, attns, 0, T, 32, B)
    optim.step()


def main():
    onmt, _ = RH.import_reference()
    os.makedirs(OUT, exist_ok=True)
    c = O.Cfg(**CFG)
    tmp = tempfile.mkdtemp()
    opt = RH.make_opt(src_word_vec_size=c.emb, tgt_word_vec_size=c.emb, rnn_size=c.hid, z_latent_dim=c.z, enc_layers=c.layers,
                      dec_layers=c.layers, encoder_type="brnn", dropout=0.0, save_model=os.path.join(tmp, "ck"))
    fields = _fields(c)
    torch.manual_seed(0)
    import contextlib
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        model = onmt.ModelConstructor.make_vi_model_mmt(opt, fields, False, None)
    with torch.no_grad():
        for br in ("location", "scale"):
            getattr(model.inf_net_image, br).fc2.weight.zero_()
            getattr(model.inf_net_image, br).fc1.bias.fill_(-10.0)
    model.train()
    batches = [O.synth_batch(c, B, S, T, n_img=16, seed=900 + i, fixed_len=False) for i in range(3)]
    tloss = RH.make_loss(model, fields, opt, training=True)
    vloss = RH.make_loss(model, fields, opt, training=False)
    optim = onmt.Optim("adam", 0.002, 5.0, lr_decay=0.5, start_decay_at=8)
    optim.set_parameters(model.parameters())
    for bt in batches[:2]:
        _update(onmt, model, tloss, optim, bt)
    trainer = onmt.TrainerMultimodal(model, tloss, vloss, optim, 0, 32, "text", "sents", 1, train_img_feats=batches[0]["table"].numpy(),
                                     valid_img_feats=batches[0]["table"].numpy(), multimodal_model_type="vi-model1", model_opt=opt,
                                     fields=fields)
    vstats = types.SimpleNamespace(accuracy=lambda: 12.34, ppl=lambda: 56.78)
    fname = trainer.drop_checkpoint(opt, 2, fields, vstats)
    assert os.path.basename(fname) == "ck_acc_12.34_ppl_56.78_e2.pt"
    with open(fname, "rb") as fi, gzip.GzipFile(os.path.join(OUT, "ref_ckpt.pt.gz"), "wb", mtime=0) as fo:
        shutil.copyfileobj(fi, fo)
    p2 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    # ---- the reference resumes from its file (train_mm_vi_model1.py:544-556, :423, :433-454) and makes one more update
    ck = torch.load(fname, map_location=lambda storage, loc: storage, weights_only=False)
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        model2 = onmt.ModelConstructor.make_vi_model_mmt(ck["opt"], fields, False, ck)
    model2.train()
    optim2 = ck["optim"]
    optim2.optimizer.load_state_dict(ck["optim"].optimizer.state_dict())
    optim2.set_parameters(model2.parameters())
    tloss2 = RH.make_loss(model2, fields, ck["opt"], training=True)
    _update(onmt, model2, tloss2, optim2, batches[2])
    arrs = {"cfg": np.array([c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, int(c.brnn), B, S, T], dtype=np.int64),
            "optim_step": np.array(optim2._step), "optim_lr": np.array(optim2.lr)}
    for i, bt in enumerate(batches):
        arrs.update({"b%d_%s" % (i, k): v.numpy() for k, v in bt.items()})
    for k, v in model2.state_dict().items():
        if "inf_net_image" in k and v.numel() > O.BIG:
            continue                                   # zero weights / never updated (see the module docstring)
        arrs["p3_" + k] = v.detach().numpy()
        arrs["p2_" + k] = p2[k].numpy()
    np.savez_compressed(os.path.join(OUT, "ref_resume.npz"), **arrs)
    shutil.rmtree(tmp)
    print("wrote", {f: os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT)})


if __name__ == "__main__":
    main()
