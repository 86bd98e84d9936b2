"""Checkpoint interop with the REAL reference (onmt/TrainerMultimodal.py:554-622, train_mm_vi_model1.py:433-454,544-556).

tests/golden/ckpt/ref_ckpt.pt.gz was written by the reference's own drop_checkpoint after two updates (oracle/make_ckpt_golden.py);
ref_resume.npz holds the parameters after the reference resumed from that file and made a third update.

  CPU : the file unpickles through the mirror (`onmt.Optim.Optim` -> variational_mmt_amd.onmt.Optim.Optim holding the pickled
        torch.optim.Adam; `torchtext.vocab.Vocab` -> the io stand-in) with the expected content.
  GPU : load -> make_vi_model_mmt(checkpoint) -> build_optim flow -> one update on the recorded batch == the reference's parameters;
        the mirror's own checkpoint has the reference's layout (a torch.optim.Adam inside, state keyed like the reference's)."""
import gzip
import os
import shutil

import numpy as np
import pytest
import torch

from oracle import vi1_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ckpt")


@pytest.fixture(autouse=True)
def _restore_modules():
    """install_as_onmt() registers the mirror as `onmt` (and torchtext stand-ins) process-wide; other CPU tests import the REAL
    reference under the same names (oracle/ref_harness.py), so put sys.modules back afterwards"""
    import sys
    before = {k: v for k, v in sys.modules.items() if k == "onmt" or k.startswith("onmt.") or k == "torchtext" or k.startswith("torchtext.")}
    yield
    for k in [k for k in sys.modules if k == "onmt" or k.startswith("onmt.") or k == "torchtext" or k.startswith("torchtext.")]:
        if k not in before:
            del sys.modules[k]
    sys.modules.update(before)


def _load_ckpt(tmp_path):
    import variational_mmt_amd
    onmt = variational_mmt_amd.install_as_onmt()
    fname = str(tmp_path / "ref_ckpt.pt")
    with gzip.open(os.path.join(G, "ref_ckpt.pt.gz"), "rb") as fi, open(fname, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    ck = torch.load(fname, map_location=lambda storage, loc: storage, weights_only=False)      # the driver's call (:546-547)
    return onmt, ck


def test_reference_checkpoint_unpickles_through_the_mirror(tmp_path):
    onmt, ck = _load_ckpt(tmp_path)
    z = np.load(os.path.join(G, "ref_resume.npz"))
    assert sorted(ck.keys()) == ["epoch", "generator", "model", "opt", "optim", "vocab"] and ck["epoch"] == 2
    assert type(ck["optim"]) is onmt.Optim and ck["optim"].method == "adam" and ck["optim"]._step == 2
    assert isinstance(ck["optim"].optimizer, torch.optim.Adam)
    st = ck["optim"].optimizer.state_dict()
    assert all(int(float(v["step"])) == 2 for v in st["state"].values())
    vs, vt, emb, hid, zd, img, layers, brnn = [int(x) for x in z["cfg"][:8]]
    c = O.Cfg(vs=vs, vt=vt, emb=emb, hid=hid, z=zd, img=img, layers=layers, brnn=bool(brnn))
    names = dict(ck["model"])
    names.update({"generator." + k: v for k, v in ck["generator"].items()})
    assert {k: tuple(v.shape) for k, v in names.items()} == {k: tuple(s) for k, s in O.param_shapes(c).items()}
    fields = onmt.io.load_fields_from_vocab(ck["vocab"], "text")
    assert len(fields["src"].vocab) == vs and len(fields["tgt"].vocab) == vt and fields["tgt"].vocab.stoi["</s>"] == 3
    for k in z.files:
        if k.startswith("p2_"):
            assert np.array_equal(names[k[3:]].numpy(), z[k]), k
    # the reference keeps no Adam state for parameters that never get a gradient (H6: inf_net_image.scale.*)
    assert len(st["state"]) == len(names) - 4


@pytest.mark.gpu
def test_resume_from_a_reference_checkpoint_matches_the_reference(tmp_path):
    onmt, ck = _load_ckpt(tmp_path)
    z = np.load(os.path.join(G, "ref_resume.npz"))
    B, S, T = [int(x) for x in z["cfg"][8:11]]
    opt = ck["opt"]
    opt.gpuid, opt.compute_dtype = [0], "f32"
    fields = onmt.io.load_fields_from_vocab(ck["vocab"], "text")
    model = onmt.ModelConstructor.make_vi_model_mmt(opt, fields, True, ck)                       # :423
    optim = ck["optim"]                                                                          # build_optim, :433-454
    optim.optimizer.load_state_dict(ck["optim"].optimizer.state_dict())
    optim.set_parameters(model.parameters())
    assert model.engine.step_count == 0                       # as executed: a NEW Adam (Optim.py:56-70)
    bt = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("b2_")}
    loss = onmt.VILoss.NMTVIModel1LossCompute(model.generator, fields["tgt"].vocab)
    table = bt["table"]
    model.set_image_tables(train=table.numpy(), valid=table.numpy())
    model.train()
    out, attns, _ = model(bt["src"].unsqueeze(2).cuda(), bt["tgt"].unsqueeze(2).cuda(), bt["src_len"].cuda(), bt["tgt_len"].cuda(), None,
                          img_indices=bt["indices"].cuda(), img_table=model._tables["train"], eps=bt["eps"])
    loss.sharded_compute_loss(None, out, attns, 0, T, 32, B)
    optim.step()
    torch.cuda.synchronize()
    assert optim._step == int(z["optim_step"]) == 3 and abs(optim.lr - float(z["optim_lr"])) < 1e-12
    sd = model.state_dict()
    worst = 0.0
    for k in z.files:
        if not k.startswith("p3_") or "inf_net_image" in k:
            continue      # image network: semantic A (the reference under torch >= 0.4, shim s7) vs as-executed B here (H1)
        got, want, before = sd[k[3:]].cpu().double(), torch.from_numpy(z[k]).double(), torch.from_numpy(z["p2_" + k[3:]]).double()
        # first step of a fresh Adam: +-lr wherever |g| >> eps; compare where the reference moved by a clear +-lr
        moved = (want - before).abs() > 0.5 * 0.002
        assert moved.float().mean().item() > 0.05, k
        err = (got - want)[moved].abs().max().item()
        worst = max(worst, err)
        assert err <= 2e-5, (k, err)
    # ---- and the mirror's own checkpoint has the reference's layout
    trainer = onmt.TrainerMultimodal(model, loss, loss, optim, 0, 32, "text", "sents", 1, train_img_feats=table.numpy(),
                                     valid_img_feats=table.numpy(), multimodal_model_type="vi-model1", model_opt=None, fields=fields)
    opt.save_model = str(tmp_path / "mine")
    import types
    fname = trainer.drop_checkpoint(opt, 3, fields, types.SimpleNamespace(accuracy=lambda: 1.0, ppl=lambda: 2.0))
    ck2 = torch.load(fname, map_location="cpu", weights_only=False)
    assert list(ck2["model"].keys()) != [] and isinstance(ck2["optim"].optimizer, torch.optim.Adam)
    st2 = ck2["optim"].optimizer.state_dict()
    assert len(st2["state"]) == len(model.engine.grads)             # no state for the never-updated scale branch (H6)
    assert all(int(float(v["step"])) == 1 for v in st2["state"].values())
    # one tensor per parameter in the file: the optimizer's params share storage with checkpoint['model'] / ['generator']
    ptrs = {v.data_ptr() for v in ck2["model"].values()} | {v.data_ptr() for v in ck2["generator"].values()}
    assert all(p.data_ptr() in ptrs for p in ck2["optim"].params)
